// dh_output.cpp -- host-side writer of the gap-closed assembly.
//
// Restates `dentist output` for any assembly graph: the graph with the join policy source/dentist/commands/
// output.d:305-348 (common/scaffold.d:373-451 normalizeUnkownJoins, :642-723 enforceJoinPolicy), fixCropping
// :931-1003, scaffoldStarts / linearWalk (scaffold.d:1021-1295), header rule :743-759, contig slices :782-835,
// unclosed gaps as 'n' runs :837-862, upper-cased insertions :864-925, closed-gaps BED :879-891, AGP :454-573, line
// wrapping :232 (fastaLineWidth, commandline.d:1697-1699); splice coordinates are the dh_insertion fields
// (common/insertions.d:110-284).  Gap joins of any two contig ends (same orientation, anti-parallel, contig-skipping),
// extension insertions at scaffold ends, cyclic scaffolds.  No device work.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dh_internal.h"

namespace {
struct LineWriter {
    FILE *f;
    int32_t width, col = 0;
    bool ok = true;
    void put(char c)
    {
        if (fputc(c, f) == EOF) ok = false;
        if (width > 0 && ++col == width) {
            if (fputc('\n', f) == EOF) ok = false;
            col = 0;
        }
    }
    void end_record()
    {
        if (col != 0 || width <= 0) {
            if (fputc('\n', f) == EOF) ok = false;
        }
        col = 0;
    }
};
const char LOWER[5] = {'a', 'c', 'g', 't', 'n'};
const char UPPER[5] = {'A', 'C', 'G', 'T', 'N'};
}  // namespace

// ---- the assembly graph (common/scaffold.d): nodes = (contig, part) with part pre < begin < end < post, numbered
// 4 * contig + part, so that node order is number order; an undirected edge is keyed by its ordered node pair and the
// edges of the graph are kept in key order (math.d: the reference sorts them the same way).
namespace {
enum { PRE = 0, BEGIN = 1, END = 2, POST = 3 };
enum { K_CONTIG = 0, K_NGAP = 1, K_INS = 2, K_PLAIN = 3 };
typedef std::pair<int32_t, int32_t> EKey;
struct Payload {
    int32_t kind = K_PLAIN;
    int64_t len = 0;   // contig length / length of the n run
    int32_t ins = -1;  // index of the insertion record
    int32_t nov = 0;   // contig edges: splice sites handed over by the incident insertions (fixCropping)
    int32_t ov_seed[2] = {0, 0}, ov_pos[2] = {0, 0};
};
inline int32_t node_of(int32_t contig, int32_t part) { return 4 * contig + part; }
inline int32_t contig_of(int32_t node) { return node >> 2; }
inline int32_t part_of(int32_t node) { return node & 3; }
inline EKey ekey(int32_t a, int32_t b) { return a <= b ? EKey(a, b) : EKey(b, a); }
inline bool real_part(int32_t p) { return p == BEGIN || p == END; }
// join predicates, scaffold.d:159-231
inline bool e_unknown(const EKey &e)
{
    const int32_t p0 = part_of(e.first), p1 = part_of(e.second);
    return contig_of(e.first) != contig_of(e.second) && p0 != p1 && !real_part(p0) && !real_part(p1);
}
inline bool e_gap(const EKey &e) { return contig_of(e.first) != contig_of(e.second) && real_part(part_of(e.first)) && real_part(part_of(e.second)); }
inline bool e_anti(const EKey &e) { return e_gap(e) && part_of(e.first) == part_of(e.second); }
inline bool e_ext(const EKey &e)
{
    if (contig_of(e.first) != contig_of(e.second)) return false;
    return (part_of(e.first) == PRE && part_of(e.second) == BEGIN) || (part_of(e.first) == END && part_of(e.second) == POST);
}

struct AGraph {
    int32_t ncontigs = 0;
    std::map<EKey, Payload> edges;
    std::vector<std::vector<EKey>> inc;  // incident edges of every node, in edge order (rebuilt by index())
    void index()
    {
        inc.assign((size_t)4 * (size_t)ncontigs, std::vector<EKey>());
        for (const auto &kv : edges) {
            inc[(size_t)kv.first.first].push_back(kv.first);
            if (kv.first.second != kv.first.first) inc[(size_t)kv.first.second].push_back(kv.first);
        }
    }
    int32_t degree(int32_t node) const { return (int32_t)inc[(size_t)node].size(); }
    static int32_t target(const EKey &e, int32_t node) { return e.first == node ? e.second : e.first; }
};

// LinearWalk, scaffold.d:1021-1170: the edges from `start` on (optionally through `first`), until an end node or until
// the cycle closes (its closing edge is the last one handed out).  false: a fork.
bool linear_walk(const AGraph &g, int32_t start, const EKey *first, std::vector<EKey> &out, bool &cyclic, std::vector<uint8_t> &visited)
{
    out.clear();
    cyclic = false;
    std::fill(visited.begin(), visited.end(), 0);
    visited[(size_t)start] = 1;
    int32_t cur = start;
    EKey cur_join;
    auto next_of = [&](int32_t node, EKey &e) {
        for (const EKey &c : g.inc[(size_t)node])
            if (!visited[(size_t)AGraph::target(c, node)]) {
                e = c;
                return true;
            }
        return false;
    };
    if (first)
        cur_join = *first;
    else {
        if (g.degree(cur) > 2) return false;
        if (!next_of(cur, cur_join)) return true;  // an isolated node (or both neighbours visited: not on a start node)
    }
    cur = AGraph::target(cur_join, cur);
    visited[(size_t)cur] = 1;
    for (;;) {
        out.push_back(cur_join);
        if (g.degree(cur) > 2) return false;
        if (cyclic) break;
        EKey nx;
        if (!next_of(cur, nx)) {
            if (g.degree(cur) > 1) {  // lastEdgeOfCycle: the incident edge that is not the one we came by
                cyclic = true;
                for (const EKey &c : g.inc[(size_t)cur])
                    if (c != cur_join) nx = c;
                cur_join = nx;
                continue;
            }
            break;
        }
        cur_join = nx;
        cur = AGraph::target(cur_join, cur);
        visited[(size_t)cur] = 1;
    }
    return true;
}

// scaffoldStarts, scaffold.d:1209-1295: for every component the smallest of its end nodes (of a cycle: where the two walks
// from its smallest node end, i.e. that node)
bool scaffold_starts(const AGraph &g, std::vector<int32_t> &starts)
{
    const int32_t nn = 4 * g.ncontigs;
    std::vector<uint8_t> unvisited((size_t)nn, 1), vis((size_t)nn, 0);
    std::vector<EKey> walk;
    starts.clear();
    for (int32_t node = 0; node < nn; node++) {
        if (!unvisited[(size_t)node]) continue;
        unvisited[(size_t)node] = 0;
        const int32_t deg = g.degree(node);
        if (deg == 0) continue;
        int32_t best = deg == 1 ? node : INT32_MAX;
        const std::vector<EKey> firsts = g.inc[(size_t)node];
        for (const EKey &f : firsts) {
            bool cyc = false;
            if (!linear_walk(g, node, &f, walk, cyc, vis)) return false;
            int32_t last = node;
            for (const EKey &e : walk) {
                last = AGraph::target(e, last);
                unvisited[(size_t)last] = 0;
            }
            best = std::min(best, last);
        }
        starts.push_back(best);
    }
    return true;
}

// normalizeUnkownJoins, scaffold.d:373-451: an n run between two contigs moves from the symbolic nodes (c, post) -- (d, pre)
// onto the contig ends that are still free, stays between extensions, and goes where a gap join took an end
void normalize_unknown_joins(AGraph &g)
{
    g.index();
    std::vector<std::pair<EKey, Payload>> fresh;
    std::vector<EKey> gone;
    for (const auto &kv : g.edges) {
        const EKey &e = kv.first;
        if (!e_unknown(e)) continue;
        const int32_t pre_end = node_of(contig_of(e.first), END), post_begin = node_of(contig_of(e.second), BEGIN);
        const bool pre_un = g.degree(pre_end) == 1, pre_ext = g.edges.count(ekey(pre_end, e.first)) != 0, pre_gap = !pre_un && !pre_ext;
        const bool post_un = g.degree(post_begin) == 1, post_ext = g.edges.count(ekey(e.second, post_begin)) != 0, post_gap = !post_un && !post_ext;
        if (pre_un && post_un) {
            fresh.emplace_back(ekey(pre_end, post_begin), kv.second);
            gone.push_back(e);
        } else if (pre_un && post_ext) {
            fresh.emplace_back(ekey(pre_end, e.second), kv.second);
            gone.push_back(e);
        } else if (pre_ext && post_un) {
            fresh.emplace_back(ekey(e.first, post_begin), kv.second);
            gone.push_back(e);
        } else if (pre_gap || post_gap)
            gone.push_back(e);
    }
    for (const EKey &e : gone) g.edges.erase(e);
    for (const auto &f : fresh) g.edges[f.first] = f.second;
    g.index();
}

// enforceJoinPolicy, scaffold.d:642-723.  policy 0 scaffoldGaps, 1 scaffolds, 2 contigs; returns the gap joins that stay out
std::vector<std::pair<EKey, Payload>> enforce_join_policy(AGraph &g, int32_t policy)
{
    std::vector<std::pair<EKey, Payload>> forbidden;
    if (policy == 2) return forbidden;
    std::vector<EKey> allowed;
    for (const auto &kv : g.edges)
        if (e_unknown(kv.first)) {
            const int32_t c = contig_of(kv.first.first), d = contig_of(kv.first.second);
            allowed.push_back(ekey(node_of(c, END), node_of(c, POST)));
            allowed.push_back(ekey(node_of(c, END), node_of(d, BEGIN)));
            allowed.push_back(ekey(node_of(d, PRE), node_of(d, BEGIN)));
        }
    std::sort(allowed.begin(), allowed.end());
    for (const auto &kv : g.edges)
        if (e_gap(kv.first) && !std::binary_search(allowed.begin(), allowed.end(), kv.first)) forbidden.push_back(kv);
    for (const auto &f : forbidden) g.edges.erase(f.first);
    if (policy == 1) {
        normalize_unknown_joins(g);
        std::vector<std::pair<EKey, Payload>> still;
        for (const auto &f : forbidden) {  // a join between scaffolds comes back when both its ends are still free
            if (g.degree(f.first.first) == 1 && g.degree(f.first.second) == 1) {
                g.edges[f.first] = f.second;
                g.index();
            } else
                still.push_back(f);
        }
        return still;
    }
    return forbidden;
}

// the edge and the flank seeds (0 front, 1 back) of an insertion record
bool insertion_edge(const dh_insertion &r, int32_t ncontigs, EKey &e, int32_t seeds[2], int32_t &nf)
{
    const int32_t c0 = r.contig_left;
    seeds[0] = (r.join & DH_JOIN_FLANK0_FRONT) ? 0 : 1;
    seeds[1] = (r.join & DH_JOIN_FLANK1_BACK) ? 1 : 0;
    if (c0 < 0 || c0 >= ncontigs) return false;
    if (r.join & DH_JOIN_EXTENSION) {
        nf = 1;
        e = seeds[0] == 0 ? ekey(node_of(c0, PRE), node_of(c0, BEGIN)) : ekey(node_of(c0, END), node_of(c0, POST));
        return true;
    }
    nf = 2;
    const int32_t c1 = r.join == 0 && r.contig_right == 0 ? c0 + 1 : r.contig_right;  // (records made before the field existed)
    if (c1 <= c0 || c1 >= ncontigs) return false;
    e = ekey(node_of(c0, seeds[0] == 0 ? BEGIN : END), node_of(c1, seeds[1] == 0 ? BEGIN : END));
    return true;
}
}  // namespace

// test surface of the graph code above (the reference's unit vectors of normalizeUnkownJoins, linearWalk and scaffoldStarts
// run against it): default edges of `ncontigs` contigs + joins4 = (contig0, part0, contig1, part1) each; normalize != 0
// runs normalizeUnkownJoins first.  Out: the edges (edges4, *nedges), the scaffold starts (starts2 = (contig, part) each),
// and the walk from walk_start2 -- through walk_first4 when given -- as node pairs in walking direction (walk4, *walk_len,
// *cyclic).  Every output array holds `cap` entries; DH_EOVERFLOW when that is too few.
extern "C" int dh_scaffold_graph_probe(int32_t ncontigs, const int32_t *joins4, int32_t njoins, int32_t normalize,
                                       int32_t *edges4, int32_t *nedges, int32_t *starts2, int32_t *nstarts,
                                       const int32_t *walk_start2, const int32_t *walk_first4, int32_t *walk4,
                                       int32_t *walk_len, int32_t *cyclic, int32_t cap)
{
    if (ncontigs < 0 || njoins < 0 || (njoins > 0 && !joins4) || cap < 0) return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: bad argument");
    AGraph g;
    g.ncontigs = ncontigs;
    for (int32_t c = 0; c < ncontigs; c++) g.edges[ekey(node_of(c, BEGIN), node_of(c, END))].kind = K_CONTIG;
    auto ok_node = [&](int32_t c, int32_t p) { return c >= 0 && c < ncontigs && p >= 0 && p <= 3; };
    for (int32_t i = 0; i < njoins; i++) {
        const int32_t *q = joins4 + 4 * (size_t)i;
        if (!ok_node(q[0], q[1]) || !ok_node(q[2], q[3])) return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: node out of range");
        g.edges[ekey(node_of(q[0], q[1]), node_of(q[2], q[3]))];
    }
    if (normalize)
        normalize_unknown_joins(g);
    else
        g.index();
    if (nedges) {
        if ((int64_t)g.edges.size() > cap) return dh_fail(DH_EOVERFLOW, "dh_scaffold_graph_probe: cap");
        int32_t k = 0;
        for (const auto &kv : g.edges) {
            if (edges4) {
                edges4[4 * k] = contig_of(kv.first.first);
                edges4[4 * k + 1] = part_of(kv.first.first);
                edges4[4 * k + 2] = contig_of(kv.first.second);
                edges4[4 * k + 3] = part_of(kv.first.second);
            }
            k++;
        }
        *nedges = k;
    }
    if (nstarts) {
        std::vector<int32_t> st;
        if (!scaffold_starts(g, st)) return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: fork in linear walk");
        if ((int64_t)st.size() > cap) return dh_fail(DH_EOVERFLOW, "dh_scaffold_graph_probe: cap");
        for (size_t k = 0; k < st.size() && starts2; k++) {
            starts2[2 * k] = contig_of(st[k]);
            starts2[2 * k + 1] = part_of(st[k]);
        }
        *nstarts = (int32_t)st.size();
    }
    if (walk_start2 && walk_len) {
        if (!ok_node(walk_start2[0], walk_start2[1])) return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: start node out of range");
        std::vector<EKey> walk;
        std::vector<uint8_t> vis((size_t)4 * (size_t)ncontigs, 0);
        bool cyc = false;
        EKey f;
        if (walk_first4) f = ekey(node_of(walk_first4[0], walk_first4[1]), node_of(walk_first4[2], walk_first4[3]));
        if (walk_first4 && !g.edges.count(f)) return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: the first join is not in the graph");
        if (!linear_walk(g, node_of(walk_start2[0], walk_start2[1]), walk_first4 ? &f : nullptr, walk, cyc, vis))
            return dh_fail(DH_EINVAL, "dh_scaffold_graph_probe: fork in linear walk");
        if ((int64_t)walk.size() > cap) return dh_fail(DH_EOVERFLOW, "dh_scaffold_graph_probe: cap");
        int32_t at = node_of(walk_start2[0], walk_start2[1]);
        for (size_t k = 0; k < walk.size(); k++) {
            const int32_t to = AGraph::target(walk[k], at);
            if (walk4) {
                walk4[4 * k] = contig_of(at);
                walk4[4 * k + 1] = part_of(at);
                walk4[4 * k + 2] = contig_of(to);
                walk4[4 * k + 3] = part_of(to);
            }
            at = to;
        }
        *walk_len = (int32_t)walk.size();
        if (cyclic) *cyclic = cyc ? 1 : 0;
    }
    return DH_OK;
}

// `dentist output` for any assembly graph:
//   * buildAssemblyGraph (output.d:305-348): default edges of the contigs; one edge per insertion that passed the gates --
//     a gap join between two contig ends (same orientation, anti-parallel, contig-skipping) or an extension edge, chosen
//     by opts->only (--only spanning | extending | both; extensions shorter than min_extension_length are skipped,
//     skipShortExtension :363-386); the n runs of the input scaffolds as unknown joins (appendUnkownJoins :350-361);
//     enforceJoinPolicy (common/scaffold.d:642-723): scaffoldGaps (0, the default) drops gap joins that do not sit on an
//     input gap, scaffolds (1) lets them back where both contig ends stay free, contigs (2) keeps all; *dropped = joins
//     skipped by the policy; normalizeUnkownJoins (:373-451).
//   * fixCropping (:931-1003): every contig is cropped at the splice sites of its incident insertions only.
//   * scaffoldStarts (:1209-1295) + linearWalk (:1021-1170): one FASTA record per walk, named after the contig of its
//     start node (scaffoldHeader :743-759, ids made unique as StringUniqifier :1035-1066 does, "\tisCyclic" for a cycle);
//     an anti-parallel insertion flips the strand of everything after it (writeNewScaffold :660-693); contig slices,
//     n runs, upper-cased insertions (:782-925; slice and strand of an insertion: common/insertions.d:230-284).
//     Two differences from the reference's letter, both where it would write wrong bases: a walk that starts on the END
//     node of a contig starts reverse-complemented (the reference starts every walk forward and asserts the opposite,
//     insertions.d:216-221), and the slice of an extension with a complement overlap is the part that extends the contig
//     (insertions.d:253-263 selects the aligned part for complement overlaps).
//   * AGP (:454-573): one line per contig slice / insertion / n run, object coordinates 1-based inclusive, components as the
//     reference writes them (contig begin in its input scaffold + crop begin, ... + crop end; the orientation column
//     follows :534 literally).  Closed-gaps BED (:879-891): every read id of the pile-up (`%(%d-%)`, ids 1-based).
// read_ids / read_ids_off[nins + 1] (int32, as dh_insertions_read_ids_off hands them out): the read ids (0-based) of
// every insertion's pile-up; NULL = the reference read alone.  nreads = entries of read_names (ids are checked against
// it; -1 = unknown, allowed without a name table).  read_names: FASTA ids of the reads for the AGP (NULL with agp_dazzler).
extern "C" int dh_output_assembly(const char *fasta_path, const char *bed_path, const char *agp_path,
                                  const uint8_t *contig_bases, const int64_t *contig_off, int32_t ncontigs,
                                  const int32_t *scaffold_of, const char *const *headers, const int32_t *gap_len,
                                  const dh_insertion *ins, int32_t nins, const uint8_t *ins_bases,
                                  const int32_t *read_ids, const int32_t *read_ids_off, int32_t nreads,
                                  const char *const *read_names, const dh_output_opts *opts, int32_t *dropped)
{
    if (!fasta_path || !contig_bases || !contig_off || !scaffold_of || !headers || !opts ||
        (nins > 0 && (!ins || !ins_bases)) || ncontigs < 0 || (read_ids && !read_ids_off))
        return dh_fail(DH_EINVAL, "dh_output_assembly: NULL argument");
    const dh_output_opts &o = *opts;
    if (o.join_policy < 0 || o.join_policy > 2) return dh_fail(DH_EINVAL, "dh_output_assembly: join_policy must be 0, 1 or 2");
    if (o.only < 0 || o.only > 3) return dh_fail(DH_EINVAL, "dh_output_assembly: only must be 0 (= spanning), 1 spanning, 2 extending or 3 both");
    const int32_t only = o.only == 0 ? 1 : o.only;
    if (agp_path && !o.agp_dazzler && !o.agp_skip_read_ids && !read_names)
        return dh_fail(DH_EINVAL, "dh_output_assembly: the AGP needs read names, agp_dazzler or agp_skip_read_ids");
    // read ids index read_names[]: offsets monotone, ids inside [0, nreads), names present (nreads < 0: ids unchecked,
    // legal only when no name table is consulted)
    const bool names_used = agp_path && !o.agp_dazzler && !o.agp_skip_read_ids;
    if (names_used && nreads < 0) return dh_fail(DH_EINVAL, "dh_output_assembly: read names need nreads");
    if (read_ids) {
        if (read_ids_off[0] < 0) return dh_fail(DH_EINVAL, "dh_output_assembly: negative read id offset");
        for (int32_t i = 0; i < nins; i++)
            if (read_ids_off[i + 1] < read_ids_off[i])
                return dh_fail(DH_EINVAL, "dh_output_assembly: read id offsets are not monotone");
        if (nreads >= 0)
            for (int32_t x = read_ids_off[0]; x < read_ids_off[nins]; x++)
                if (read_ids[x] < 0 || read_ids[x] >= nreads)
                    return dh_fail(DH_EINVAL, "dh_output_assembly: read id outside [0, nreads)");
    }
    if (names_used) {
        for (int32_t i = 0; i < nins; i++)
            if (!read_ids && ins[i].status == DH_PILE_OK && (ins[i].ref_read_id < 0 || ins[i].ref_read_id >= nreads))
                return dh_fail(DH_EINVAL, "dh_output_assembly: reference read id outside [0, nreads)");
        for (int32_t r = 0; r < nreads; r++)
            if (!read_names[r]) return dh_fail(DH_EINVAL, "dh_output_assembly: NULL read name");
    }
    // ---- the graph
    AGraph g;
    g.ncontigs = ncontigs;
    for (int32_t c = 0; c < ncontigs; c++) {
        Payload &p = g.edges[ekey(node_of(c, BEGIN), node_of(c, END))];
        p.kind = K_CONTIG;
        p.len = contig_off[c + 1] - contig_off[c];
    }
    for (int32_t i = 0; i < nins; i++) {
        if (ins[i].status != DH_PILE_OK) continue;
        EKey e;
        int32_t seeds[2], nf;
        if (!insertion_edge(ins[i], ncontigs, e, seeds, nf)) return dh_fail(DH_EINVAL, "dh_output_assembly: insertion outside the contigs");
        if (ins[i].ins_begin < 0 || ins[i].ins_begin > ins[i].ins_end || ins[i].ins_end > ins[i].cons_len || ins[i].cons_off < 0)
            return dh_fail(DH_EINVAL, "dh_output_assembly: insertion outside its consensus");
        if (nf == 1) {
            if (!(only & 2) || ins[i].ins_end - ins[i].ins_begin < o.min_extension_length) continue;
        } else if (!(only & 1))
            continue;
        if (g.edges.count(e)) return dh_fail(DH_EINVAL, nf == 1 ? "dh_output_assembly: two insertions for one extension" : "dh_output_assembly: two insertions for one gap");
        Payload &p = g.edges[e];
        p.kind = K_INS;
        p.ins = i;
    }
    for (int32_t c = 0; c + 1 < ncontigs; c++)
        if (scaffold_of[c] == scaffold_of[c + 1]) {
            Payload &p = g.edges[ekey(node_of(c, POST), node_of(c + 1, PRE))];
            p.kind = K_NGAP;
            p.len = gap_len ? gap_len[c] : 0;
        }
    g.index();
    const std::vector<std::pair<EKey, Payload>> forbidden = enforce_join_policy(g, o.join_policy);
    if (dropped) *dropped = (int32_t)forbidden.size();
    normalize_unknown_joins(g);
    // fixCropping
    for (int32_t c = 0; c < ncontigs; c++) {
        Payload &cp = g.edges[ekey(node_of(c, BEGIN), node_of(c, END))];
        cp.nov = 0;
        for (int32_t part : {BEGIN, END})
            for (const EKey &e : g.inc[(size_t)node_of(c, part)]) {
                const Payload &p = g.edges[e];
                if (p.kind != K_INS) continue;
                const dh_insertion &r = ins[p.ins];
                const int32_t f = r.contig_left == c ? 0 : 1;
                if (cp.nov >= 2) return dh_fail(DH_EINVAL, "dh_output_assembly: too many splice sites on a contig");
                cp.ov_seed[cp.nov] = part == BEGIN ? 0 : 1;
                cp.ov_pos[cp.nov] = f == 0 ? r.left_aepos : r.right_abpos;
                cp.nov++;
            }
    }
    std::vector<int32_t> starts;
    if (!scaffold_starts(g, starts)) return dh_fail(DH_EINVAL, "dh_output_assembly: fork in the assembly graph (two insertions on one contig end)");
    // position of every contig inside its input scaffold (ContigSegment.begin)
    std::vector<int64_t> cbegin((size_t)std::max(ncontigs, 1), 0);
    for (int32_t c = 1; c < ncontigs; c++)
        if (scaffold_of[c] == scaffold_of[c - 1])
            cbegin[(size_t)c] = cbegin[(size_t)c - 1] + (contig_off[c] - contig_off[c - 1]) + (gap_len ? gap_len[c - 1] : 0);
    FILE *f = fopen(fasta_path, "w");
    if (!f) return dh_fail(DH_EIO, std::string("cannot open ") + fasta_path);
    FILE *bed = nullptr, *agp = nullptr;
    auto close_all = [&]() {
        bool ok = fclose(f) == 0;
        if (bed && fclose(bed) != 0) ok = false;
        if (agp && fclose(agp) != 0) ok = false;
        return ok;
    };
    if (bed_path && !(bed = fopen(bed_path, "w"))) {
        close_all();
        return dh_fail(DH_EIO, std::string("cannot open ") + bed_path);
    }
    if (agp_path && !(agp = fopen(agp_path, "w"))) {
        close_all();
        return dh_fail(DH_EIO, std::string("cannot open ") + agp_path);
    }
    bool ok = true;
    if (agp) {  // writeAGPHeader, output.d:454-462
        ok = ok && fprintf(agp, "##agp-version\t%s\n", o.agp_version ? o.agp_version : "2.1") > 0;
        ok = ok && fprintf(agp, "# TOOL: %s\n", o.tool ? o.tool : "dentist-hip") > 0;
        ok = ok && fprintf(agp, "# INPUT_ASSEMBLY: %s\n", o.input_assembly ? o.input_assembly : "") > 0;
        ok = ok && fprintf(agp, "# object\tobject_beg\tobject_end\tpart_number\tcomponent_type\tcomponent_id/gap_length\t"
                                "component_beg/gap_type\tcomponent_end/linkage\torientation\tlinkage_evidence\n") > 0;
    }
    auto header_id = [&](int32_t c) {
        std::string id(headers[scaffold_of[c]] ? headers[scaffold_of[c]] : "");
        const size_t tab = id.find('\t');
        if (tab != std::string::npos) id.resize(tab);
        return id;
    };
    // StringUniqifier, output.d:1035-1066
    std::map<std::string, int64_t> dup;
    std::map<int32_t, std::string> ucache;
    auto uniq = [&](int32_t key, const std::string &label) {
        auto hit = ucache.find(key);
        if (hit != ucache.end()) return hit->second;
        int64_t n = dup.count(label) ? dup[label] : 0;
        std::string u = n == 0 ? label : label + "-" + std::to_string(n);
        while (dup.count(u)) u = label + "-" + std::to_string(++n);
        ucache[key] = u;
        dup[label] = n + 1;
        return u;
    };
    LineWriter w{f, o.line_width};
    std::vector<EKey> walk;
    std::vector<uint8_t> vis((size_t)4 * (size_t)std::max(ncontigs, 1), 0);
    for (int32_t start : starts) {
        bool cyc = false;
        if (!linear_walk(g, start, nullptr, walk, cyc, vis)) {
            close_all();
            return dh_fail(DH_EINVAL, "dh_output_assembly: fork in the assembly graph");
        }
        const std::string id = uniq(contig_of(start), header_id(contig_of(start)));
        ok = ok && fprintf(f, ">%s\tscaffold-%d%s\n", id.c_str(), contig_of(start) + 1, cyc ? "\tisCyclic" : "") > 0;
        int64_t coord = 1;  // 1-based scaffold coordinate of the next base (output.d currentScaffoldCoord)
        int32_t part = 1;   // currentScaffoldPartId
        bool comp = part_of(start) == END;  // globalComplement
        int32_t at = start;
        for (const EKey &e : walk) {
            const Payload &p = g.edges[e];
            const int32_t to = AGraph::target(e, at);
            if (p.kind == K_CONTIG) {
                const int32_t c = contig_of(at);
                int64_t from = 0, upto = p.len;  // getInfoForExistingContig, insertions.d:161-221
                for (int32_t x = 0; x < p.nov; x++) {
                    if (p.ov_seed[x] == 0)
                        from = p.ov_pos[x];
                    else
                        upto = p.ov_pos[x];
                }
                if (from > upto || upto > p.len || from < 0) {
                    close_all();
                    return dh_fail(DH_EINVAL, "dh_output_assembly: splice sites cross on a contig");
                }
                for (int64_t x = 0; x < upto - from; x++) {
                    uint8_t b = contig_bases[contig_off[c] + (comp ? upto - 1 - x : from + x)];
                    if (comp && b < 4) b = (uint8_t)(3 - b);
                    w.put(LOWER[b < 4 ? b : 4]);
                }
                if (agp) {  // writeAGPContig + writeAGPComponent, output.d:464-531
                    const std::string cid = o.agp_dazzler ? std::to_string(c + 1) : header_id(c);
                    ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tW\t%s\t%lld\t%lld\t%s\tna\n", id.c_str(), (long long)coord,
                                       (long long)(coord + (upto - from) - 1), part, cid.c_str(),
                                       (long long)(cbegin[(size_t)c] + from), (long long)(cbegin[(size_t)c] + upto), comp ? "+" : "-") > 0;
                }
                coord += upto - from;
            } else if (p.kind == K_NGAP) {
                for (int64_t x = 0; x < p.len; x++) w.put('n');
                if (agp)  // writeAGPGap, output.d:533-555
                    ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tN\t%lld\tscaffold\tyes\tna\tunspecified\n", id.c_str(),
                                       (long long)coord, (long long)(coord + p.len - 1), part, (long long)p.len) > 0;
                coord += p.len;
            } else if (p.kind == K_INS) {
                const dh_insertion &in = ins[p.ins];
                const uint8_t *cons = ins_bases + in.cons_off;
                // the slice on the stored consensus (getInfoForNewSequenceInsertion, insertions.d:230-284) ...
                const int64_t sb = in.comp ? in.cons_len - in.ins_end : in.ins_begin;
                const int64_t se = in.comp ? in.cons_len - in.ins_begin : in.ins_end;
                const int64_t n = se - sb;
                // ... and its strand: the complement flag of the overlap on the contig the walk comes from (on the other
                // flank of an anti-parallel join it is the opposite one) ^ globalComplement
                bool first_comp = in.comp != 0;
                if (!(in.join & DH_JOIN_EXTENSION) && contig_of(at) != in.contig_left && e_anti(e)) first_comp = !first_comp;
                const bool eff = first_comp != comp;
                for (int64_t x = 0; x < n; x++) {
                    uint8_t b = eff ? cons[se - 1 - x] : cons[sb + x];
                    if (eff && b < 4) b = (uint8_t)(3 - b);
                    w.put((o.highlight ? UPPER : LOWER)[b < 4 ? b : 4]);
                }
                // the read ids of the pile-up, 1-based, ascending (makeInsertion, processPileUps/package.d:789-798)
                std::vector<int32_t> ids;
                if (read_ids)
                    for (int32_t x = read_ids_off[p.ins]; x < read_ids_off[p.ins + 1]; x++) ids.push_back(read_ids[x] + 1);
                else
                    ids.push_back(in.ref_read_id + 1);
                std::sort(ids.begin(), ids.end());
                std::string idlist;
                for (size_t x = 0; x < ids.size(); x++) idlist += (x ? "-" : "") + std::to_string(ids[x]);
                if (agp) {  // writeAGPInsertion, output.d:489-512
                    std::string comp_id;
                    if (o.agp_skip_read_ids)
                        comp_id = std::to_string(ids.size()) + " reads";
                    else if (o.agp_dazzler)
                        comp_id = "reads-" + idlist;
                    else
                        for (size_t x = 0; x < ids.size(); x++) comp_id += (x ? " " : "") + std::string(read_names[ids[x] - 1]);
                    ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tO\t%s\t%lld\t%lld\t%s\tclone_contig\n", id.c_str(),
                                       (long long)coord, (long long)(coord + n - 1), part, comp_id.c_str(), (long long)sb,
                                       (long long)se, eff ? "+" : "-") > 0;
                }
                // output.d:879-891: currentScaffoldCoord - 1 and nextScaffoldCoord (= current + length)
                if (bed)
                    ok = ok && fprintf(bed, "%s\t%lld\t%lld\tcontigs-%d-%d|reads-%s\n", id.c_str(), (long long)(coord - 1),
                                       (long long)(coord + n), contig_of(at) + 1, contig_of(to) + 1, idlist.c_str()) > 0;
                coord += n;
                if (e_anti(e)) comp = !comp;
            }
            part++;
            at = to;
        }
        w.end_record();
    }
    ok = ok && w.ok;
    if (!close_all()) ok = false;
    return ok ? DH_OK : dh_fail(DH_EIO, std::string("short write to ") + fasta_path);
}

extern "C" void dh_default_output_opts(dh_output_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->line_width = 50;  // commandline.d:1697-1699
    o->highlight = 1;
    o->join_policy = 0;  // scaffoldGaps (commandline.d: --join-policy default)
    o->only = 1;         // --only spanning (commandline.d:2230-2250)
    o->min_extension_length = 100;  // commandline.d:2098-2100
    o->agp_version = "2.1";
    o->tool = "dentist-hip";
    o->input_assembly = "";
}

// the linear-scaffold subset as before: FASTA + BED listing the reference read of every closed gap
extern "C" int dh_output_fasta(const char *fasta_path, const char *bed_path, const uint8_t *contig_bases,
                               const int64_t *contig_off, int32_t ncontigs, const int32_t *scaffold_of,
                               const char *const *headers, const int32_t *gap_len, const dh_insertion *ins,
                               int32_t nins, const uint8_t *ins_bases, int32_t line_width, int32_t highlight)
{
    dh_output_opts o;
    dh_default_output_opts(&o);
    o.line_width = line_width;
    o.highlight = highlight;
    for (int32_t i = 0; i < nins; i++)  // (this entry point always refused joins between scaffolds)
        if (ins && ins[i].status == DH_PILE_OK && ins[i].join == 0 && ins[i].contig_left >= 0 && ins[i].contig_left + 1 < ncontigs &&
            scaffold_of && scaffold_of[ins[i].contig_left] != scaffold_of[ins[i].contig_left + 1])
            return dh_fail(DH_EINVAL, "dh_output_fasta: insertion does not join two contigs of one scaffold");
    return dh_output_assembly(fasta_path, bed_path, nullptr, contig_bases, contig_off, ncontigs, scaffold_of, headers, gap_len,
                              ins, nins, ins_bases, nullptr, nullptr, -1, nullptr, &o, nullptr);
}
