// dh_output.cpp -- host-side writer of the gap-closed assembly.
//
// Restates `dentist output` for assemblies whose insertions join consecutive contigs: assembly graph with
// the join policy source/dentist/commands/output.d:305-348 (common/scaffold.d:642-715), fixCropping
// :931-1003, header rule :743-759, contig slices :782-835, unclosed gaps as 'n' runs :837-862, upper-cased
// insertions :864-925, closed-gaps BED :879-891, AGP :454-573, line wrapping :232 (fastaLineWidth,
// commandline.d:1697-1699); splice coordinates are the dh_insertion fields (common/insertions.d:110-284).
// Not restated: extension insertions at scaffold ends, anti-parallel joins, cyclic scaffolds -- the process
// stage does not produce them.  No device work.
#include <algorithm>
#include <cstdio>
#include <cstring>
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

// `dentist output` for an assembly whose insertions close gaps between consecutive contigs:
//   * buildAssemblyGraph (output.d:305-348): one join per insertion that passed the gates; the join policy
//     (enforceJoinPolicy, common/scaffold.d:642-715) decides about insertions BETWEEN input scaffolds:
//     scaffoldGaps (0, the default) drops them, scaffolds (1) keeps one when both scaffold ends are still
//     free, contigs (2) keeps every one; kept joins merge the two scaffolds into one output record named
//     after its first contig (scaffoldHeader :743-759).  *dropped = insertions skipped by the policy.
//   * fixCropping (:931-1003): every contig is cropped at the splice sites of its incident insertions
//     only -- no insertion, no cropping; crossing splice sites on one contig are an error here (the
//     reference asserts).
//   * writers (:782-925) and the AGP (:454-573): one AGP line per contig slice / insertion / remaining gap,
//     object coordinates 1-based inclusive, components as the reference writes them (contig begin in its
//     input scaffold + crop begin, ... + crop end; the orientation column follows :534 literally).
//   * closed-gaps BED (:879-891): every read id of the pile-up (`%(%d-%)`, ids 1-based).
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
    std::vector<int32_t> closing((size_t)std::max(ncontigs, 1), -1);
    int32_t ndropped = 0;
    for (int32_t i = 0; i < nins; i++) {
        if (ins[i].status != DH_PILE_OK) continue;
        const int32_t c = ins[i].contig_left;
        if (c < 0 || c + 1 >= ncontigs) return dh_fail(DH_EINVAL, "dh_output_assembly: insertion outside the contigs");
        if (scaffold_of[c] != scaffold_of[c + 1] && o.join_policy == 0) {
            ndropped++;  // "skipping pile up due to joinPolicy" (output.d:337-344)
            continue;
        }
        if (closing[(size_t)c] >= 0) return dh_fail(DH_EINVAL, "dh_output_assembly: two insertions for one gap");
        closing[(size_t)c] = i;
    }
    if (dropped) *dropped = ndropped;
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
    LineWriter w{f, o.line_width};
    int32_t c = 0;
    while (c < ncontigs) {
        const std::string id = header_id(c);
        // the uniquified id is the id itself: an output scaffold starts with the first contig of an input scaffold
        ok = ok && fprintf(f, ">%s\tscaffold-%d\n", id.c_str(), c + 1) > 0;
        int64_t coord = 1;  // 1-based scaffold coordinate of the next base (output.d currentScaffoldCoord)
        int32_t part = 1;   // currentScaffoldPartId
        int32_t from = 0;   // the current contig is kept from here
        for (;; c++) {
            const int64_t clen = contig_off[c + 1] - contig_off[c];
            const bool scaffold_end = c + 1 >= ncontigs || scaffold_of[c + 1] != scaffold_of[c];
            int32_t ci = c + 1 < ncontigs ? closing[(size_t)c] : -1;
            // joinPolicy scaffolds: an insertion between two scaffolds stands when both ends are free -- always the case
            // for joins of consecutive contigs; contigs: every join stands
            const bool last = scaffold_end && ci < 0;
            const int64_t to = ci >= 0 ? ins[ci].left_aepos : clen;
            if (from > to || to > clen) {
                close_all();
                return dh_fail(DH_EINVAL, "dh_output_assembly: splice sites cross on a contig");
            }
            for (int64_t x = from; x < to; x++) {
                const uint8_t b = contig_bases[contig_off[c] + x];
                w.put(LOWER[b < 4 ? b : 4]);
            }
            if (agp) {  // writeAGPContig + writeAGPComponent, output.d:464-531
                const std::string cid = o.agp_dazzler ? std::to_string(c + 1) : header_id(c);
                ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tW\t%s\t%lld\t%lld\t-\tna\n", id.c_str(), (long long)coord,
                                   (long long)(coord + (to - from) - 1), part, cid.c_str(),
                                   (long long)(cbegin[(size_t)c] + from), (long long)(cbegin[(size_t)c] + to)) > 0;
            }
            coord += to - from;
            part++;
            from = 0;
            if (last) break;
            if (ci >= 0) {
                const dh_insertion &in = ins[ci];
                if (in.ins_begin < 0 || in.ins_begin > in.ins_end || in.ins_end > in.cons_len || in.cons_off < 0) {
                    close_all();
                    return dh_fail(DH_EINVAL, "dh_output_assembly: insertion outside its consensus");
                }
                const uint8_t *cons = ins_bases + in.cons_off;
                const int64_t n = (int64_t)in.ins_end - in.ins_begin;
                for (int64_t x = 0; x < n; x++) {
                    // oriented consensus: reverse complement of the stored sequence when comp is set
                    const int64_t p = in.ins_begin + x;
                    uint8_t b = in.comp ? cons[in.cons_len - 1 - p] : cons[p];
                    if (in.comp && b < 4) b = (uint8_t)(3 - b);
                    w.put((o.highlight ? UPPER : LOWER)[b < 4 ? b : 4]);
                }
                // the read ids of the pile-up, 1-based, ascending (makeInsertion, processPileUps/package.d:789-798)
                std::vector<int32_t> ids;
                if (read_ids)
                    for (int32_t x = read_ids_off[ci]; x < read_ids_off[ci + 1]; x++) ids.push_back(read_ids[x] + 1);
                else
                    ids.push_back(in.ref_read_id + 1);
                std::sort(ids.begin(), ids.end());
                std::string idlist;
                for (size_t x = 0; x < ids.size(); x++) idlist += (x ? "-" : "") + std::to_string(ids[x]);
                if (agp) {  // writeAGPInsertion, output.d:489-512; the slice is on the stored consensus (insertions.d:230-284)
                    std::string comp_id;
                    if (o.agp_skip_read_ids)
                        comp_id = std::to_string(ids.size()) + " reads";
                    else if (o.agp_dazzler)
                        comp_id = "reads-" + idlist;
                    else
                        for (size_t x = 0; x < ids.size(); x++) comp_id += (x ? " " : "") + std::string(read_names[ids[x] - 1]);
                    const int64_t sb = in.comp ? in.cons_len - in.ins_end : in.ins_begin;
                    const int64_t se = in.comp ? in.cons_len - in.ins_begin : in.ins_end;
                    ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tO\t%s\t%lld\t%lld\t%s\tclone_contig\n", id.c_str(),
                                       (long long)coord, (long long)(coord + n - 1), part, comp_id.c_str(), (long long)sb,
                                       (long long)se, in.comp ? "+" : "-") > 0;
                }
                // output.d:879-891: currentScaffoldCoord - 1 and nextScaffoldCoord (= current + length)
                if (bed)
                    ok = ok && fprintf(bed, "%s\t%lld\t%lld\tcontigs-%d-%d|reads-%s\n", id.c_str(), (long long)(coord - 1),
                                       (long long)(coord + n), c + 1, c + 2, idlist.c_str()) > 0;
                coord += n;
                part++;
                from = in.right_abpos;
            } else {
                const int32_t g = gap_len ? gap_len[c] : 0;
                for (int32_t x = 0; x < g; x++) w.put('n');
                if (agp)  // writeAGPGap, output.d:533-555
                    ok = ok && fprintf(agp, "%s\t%lld\t%lld\t%d\tN\t%d\tscaffold\tyes\tna\tunspecified\n", id.c_str(),
                                       (long long)coord, (long long)(coord + g - 1), part, g) > 0;
                coord += g;
                part++;
            }
        }
        w.end_record();
        c++;
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
        if (ins && ins[i].status == DH_PILE_OK && ins[i].contig_left >= 0 && ins[i].contig_left + 1 < ncontigs &&
            scaffold_of && scaffold_of[ins[i].contig_left] != scaffold_of[ins[i].contig_left + 1])
            return dh_fail(DH_EINVAL, "dh_output_fasta: insertion does not join two contigs of one scaffold");
    return dh_output_assembly(fasta_path, bed_path, nullptr, contig_bases, contig_off, ncontigs, scaffold_of, headers, gap_len,
                              ins, nins, ins_bases, nullptr, nullptr, -1, nullptr, &o, nullptr);
}
