// dh_binio.cpp -- the binary containers that travel between DENTIST's commands (host only):
//   pile-ups.db    written by `collect`, read by `process`   source/dentist/common/binio/pileupdb.d
//   insertions.db  written by `process`, read by `output`    source/dentist/common/binio/insertiondb.d
// Both: an index struct of size_t byte offsets, then homogeneous arrays of POD records; array
// fields are ArrayStorage{size_t ptr; size_t length} = (file offset, element count)
// (binio/common.d:208-281).  Records use the x86-64 layout of the D structs (natural alignment):
//   PileUpDbIndex{pileUps, readAlignments, seededAlignments, localAlignments, tracePoints, eof}      48 B
//   PileUp = ArrayStorage(ReadAlignment)   ReadAlignment = ArrayStorage(SeededAlignment)             16 B
//   SeededAlignmentStorage{size_t id; u32 contigAId, contigALength, contigBId, contigBLength;
//       u8 flags; ArrayStorage localAlignments; u16 tracePointDistance; u8 seed}                     56 B
//   LocalAlignmentStorage{u32 aBegin, aEnd, bBegin, bEnd, numDiffs; ArrayStorage tracePoints}        40 B
//   TracePointStorage{u16 numDiffs, numBasePairs}                                                     4 B
//   InsertionDbIndex{insertions, compressedBaseQuads, overlaps, localAlignments, tracePoints,
//       readIds, eof}                                                                                 56 B
//   InsertionStorage{ContigNode start, end (size_t contigId; u8 contigPart); u8 baseOffset;
//       size_t sequenceLength; ArrayStorage sequence; size_t contigLength; ArrayStorage overlaps;
//       ArrayStorage readIds}                                                                        104 B
//   sequence bytes: 4 bases per byte, a=0 c=1 t=2 g=3, first base in the low bits (common.d:324-345)
// (pileupdb.d:710-897, insertiondb.d:738-1031; sizes asserted by the reference's own unit tests as
// sums of T.sizeof, pileupdb.d:439-446.)
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dh_internal.h"

namespace {
#pragma pack(push, 1)
struct ArrSt {
    uint64_t ptr, length;
};
struct SeededSt {
    uint64_t id;
    uint32_t contigAId, contigALength, contigBId, contigBLength;
    uint8_t flags;
    uint8_t pad0[7];
    ArrSt localAlignments;
    uint16_t tracePointDistance;
    uint8_t seed;
    uint8_t pad1[5];
};
struct LocalSt {
    uint32_t aBegin, aEnd, bBegin, bEnd, numDiffs;
    uint8_t pad0[4];
    ArrSt tracePoints;
};
struct NodeSt {
    uint64_t contigId;
    uint8_t contigPart;
    uint8_t pad[7];
};
struct InsertionSt {
    NodeSt start, end;
    uint8_t baseOffset;
    uint8_t pad0[7];
    uint64_t sequenceLength;
    ArrSt sequence;
    uint64_t contigLength;
    ArrSt overlaps, readIds;
};
#pragma pack(pop)
static_assert(sizeof(ArrSt) == 16 && sizeof(SeededSt) == 56 && sizeof(LocalSt) == 40 && sizeof(InsertionSt) == 104,
              "D struct images");

struct Writer {
    FILE *f;
    bool ok = true;
    template <typename T>
    void put(const T *p, size_t n)
    {
        if (n && fwrite(p, sizeof(T), n, f) != n) ok = false;
    }
};

bool read_all(const char *path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    return true;
}

// the seeded-alignment / local-alignment / trace-point blocks shared by both containers
void write_chain_blocks(Writer &w, const dh_seeded *sa, int64_t nsa, const dh_chain_la *la, const uint16_t *tp,
                        uint64_t la_ptr, uint64_t tp_ptr)
{
    uint64_t lp = la_ptr;
    int64_t li = 0;
    for (int64_t i = 0; i < nsa; i++) {
        SeededSt s;
        memset(&s, 0, sizeof(s));
        s.id = (uint64_t)sa[i].id;
        s.contigAId = sa[i].contig_a_id;
        s.contigALength = sa[i].contig_a_len;
        s.contigBId = sa[i].contig_b_id;
        s.contigBLength = sa[i].contig_b_len;
        s.flags = sa[i].flags;
        s.localAlignments = ArrSt{lp, (uint64_t)sa[i].nla};
        s.tracePointDistance = sa[i].tspace;
        s.seed = sa[i].seed;
        w.put(&s, 1);
        lp += (uint64_t)sa[i].nla * sizeof(LocalSt);
        li += sa[i].nla;
    }
    uint64_t tpp = tp_ptr;
    for (int64_t i = 0; i < li; i++) {
        LocalSt l;
        memset(&l, 0, sizeof(l));
        l.aBegin = la[i].a_begin;
        l.aEnd = la[i].a_end;
        l.bBegin = la[i].b_begin;
        l.bEnd = la[i].b_end;
        l.numDiffs = la[i].diffs;
        l.tracePoints = ArrSt{tpp, (uint64_t)la[i].ntp};
        w.put(&l, 1);
        tpp += (uint64_t)la[i].ntp * 4;
    }
    int64_t ntp = 0;
    for (int64_t i = 0; i < li; i++) ntp += la[i].ntp;
    w.put(tp, (size_t)ntp * 2);
}
}  // namespace

struct dh_chaindb {
    // pile-ups.db: piles -> read alignments -> seeded alignments; insertions.db: insertions -> overlaps
    std::vector<int32_t> nra_of_pile, nsa_of_ra;
    std::vector<dh_insertion_rec> ins;
    std::vector<uint8_t> ins_bases;  // codes a,c,g,t = 0..3 (DAZZ order), concatenated
    std::vector<uint32_t> read_ids;
    std::vector<dh_seeded> sa;
    std::vector<dh_chain_la> la;
    std::vector<uint16_t> tp;
};

extern "C" void dh_chaindb_destroy(dh_chaindb *d) { delete d; }
extern "C" int32_t dh_chaindb_npiles(const dh_chaindb *d) { return d ? (int32_t)d->nra_of_pile.size() : 0; }
extern "C" const int32_t *dh_chaindb_pile_counts(const dh_chaindb *d) { return d ? d->nra_of_pile.data() : nullptr; }
extern "C" int32_t dh_chaindb_nread_alignments(const dh_chaindb *d) { return d ? (int32_t)d->nsa_of_ra.size() : 0; }
extern "C" const int32_t *dh_chaindb_read_alignment_counts(const dh_chaindb *d) { return d ? d->nsa_of_ra.data() : nullptr; }
extern "C" int64_t dh_chaindb_nseeded(const dh_chaindb *d) { return d ? (int64_t)d->sa.size() : 0; }
extern "C" const dh_seeded *dh_chaindb_seeded(const dh_chaindb *d) { return d ? d->sa.data() : nullptr; }
extern "C" int64_t dh_chaindb_nlas(const dh_chaindb *d) { return d ? (int64_t)d->la.size() : 0; }
extern "C" const dh_chain_la *dh_chaindb_las(const dh_chaindb *d) { return d ? d->la.data() : nullptr; }
extern "C" int64_t dh_chaindb_ntrace(const dh_chaindb *d) { return d ? (int64_t)d->tp.size() : 0; }
extern "C" const uint16_t *dh_chaindb_trace(const dh_chaindb *d) { return d ? d->tp.data() : nullptr; }
extern "C" int32_t dh_chaindb_ninsertions(const dh_chaindb *d) { return d ? (int32_t)d->ins.size() : 0; }
extern "C" const dh_insertion_rec *dh_chaindb_insertions(const dh_chaindb *d) { return d ? d->ins.data() : nullptr; }
extern "C" const uint8_t *dh_chaindb_bases(const dh_chaindb *d) { return d ? d->ins_bases.data() : nullptr; }
extern "C" const uint32_t *dh_chaindb_read_ids(const dh_chaindb *d) { return d ? d->read_ids.data() : nullptr; }

extern "C" int dh_pileupdb_write(const char *path, int32_t npiles, const int32_t *nra_of_pile, const int32_t *nsa_of_ra,
                                 const dh_seeded *sa, const dh_chain_la *la, const uint16_t *tp)
{
    if (!path || npiles < 0 || (npiles > 0 && (!nra_of_pile || !nsa_of_ra || !sa || !la)))
        return dh_fail(DH_EINVAL, "dh_pileupdb_write: bad argument");
    int64_t nra = 0, nsa = 0, nla = 0, ntp = 0;
    for (int32_t p = 0; p < npiles; p++) nra += nra_of_pile[p];
    for (int64_t r = 0; r < nra; r++) nsa += nsa_of_ra[r];
    for (int64_t s = 0; s < nsa; s++) nla += sa[s].nla;
    for (int64_t l = 0; l < nla; l++) ntp += la[l].ntp;
    if (ntp > 0 && !tp) return dh_fail(DH_EINVAL, "dh_pileupdb_write: trace is NULL");
    uint64_t index[6];
    index[0] = sizeof(index);
    index[1] = index[0] + 16ull * (uint64_t)npiles;
    index[2] = index[1] + 16ull * (uint64_t)nra;
    index[3] = index[2] + sizeof(SeededSt) * (uint64_t)nsa;
    index[4] = index[3] + sizeof(LocalSt) * (uint64_t)nla;
    index[5] = index[4] + 4ull * (uint64_t)ntp;
    FILE *f = fopen(path, "wb");
    if (!f) return dh_fail(DH_EIO, std::string("cannot open ") + path);
    Writer w{f};
    w.put(index, 6);
    uint64_t rp = index[1];
    for (int32_t p = 0; p < npiles; p++) {
        const ArrSt a{rp, (uint64_t)nra_of_pile[p]};
        w.put(&a, 1);
        rp += 16ull * (uint64_t)nra_of_pile[p];
    }
    uint64_t sp = index[2];
    for (int64_t r = 0; r < nra; r++) {
        const ArrSt a{sp, (uint64_t)nsa_of_ra[r]};
        w.put(&a, 1);
        sp += sizeof(SeededSt) * (uint64_t)nsa_of_ra[r];
    }
    write_chain_blocks(w, sa, nsa, la, tp, index[3], index[4]);
    if (fclose(f) != 0) w.ok = false;
    return w.ok ? DH_OK : dh_fail(DH_EIO, std::string("short write to ") + path);
}

static bool parse_chain_blocks(const std::vector<uint8_t> &b, uint64_t sa_ptr, uint64_t la_ptr, uint64_t tp_ptr, uint64_t tp_end,
                               dh_chaindb *d)
{
    if (sa_ptr > la_ptr || la_ptr > tp_ptr || tp_ptr > tp_end || tp_end > b.size()) return false;
    if ((la_ptr - sa_ptr) % sizeof(SeededSt) || (tp_ptr - la_ptr) % sizeof(LocalSt) || (tp_end - tp_ptr) % 4) return false;
    const size_t nsa = (la_ptr - sa_ptr) / sizeof(SeededSt), nla = (tp_ptr - la_ptr) / sizeof(LocalSt);
    uint64_t lp = la_ptr;
    for (size_t i = 0; i < nsa; i++) {
        SeededSt s;
        memcpy(&s, b.data() + sa_ptr + i * sizeof(SeededSt), sizeof(s));
        if (s.localAlignments.ptr != lp) return false;  // arrays are stored in order, back to back
        dh_seeded o;
        memset(&o, 0, sizeof(o));
        o.id = (int64_t)s.id;
        o.contig_a_id = s.contigAId;
        o.contig_a_len = s.contigALength;
        o.contig_b_id = s.contigBId;
        o.contig_b_len = s.contigBLength;
        o.flags = s.flags;
        o.seed = s.seed;
        o.tspace = s.tracePointDistance;
        o.nla = (int32_t)s.localAlignments.length;
        d->sa.push_back(o);
        lp += s.localAlignments.length * sizeof(LocalSt);
    }
    if (lp != tp_ptr) return false;
    uint64_t tpp = tp_ptr;
    for (size_t i = 0; i < nla; i++) {
        LocalSt l;
        memcpy(&l, b.data() + la_ptr + i * sizeof(LocalSt), sizeof(l));
        if (l.tracePoints.ptr != tpp) return false;
        dh_chain_la o;
        o.a_begin = l.aBegin;
        o.a_end = l.aEnd;
        o.b_begin = l.bBegin;
        o.b_end = l.bEnd;
        o.diffs = l.numDiffs;
        o.ntp = (int32_t)l.tracePoints.length;
        d->la.push_back(o);
        tpp += l.tracePoints.length * 4;
    }
    if (tpp != tp_end) return false;
    d->tp.resize((tp_end - tp_ptr) / 2);
    if (!d->tp.empty()) memcpy(d->tp.data(), b.data() + tp_ptr, tp_end - tp_ptr);
    return true;
}

extern "C" int dh_pileupdb_read(const char *path, dh_chaindb **out)
{
    if (!path || !out) return dh_fail(DH_EINVAL, "dh_pileupdb_read: NULL argument");
    std::vector<uint8_t> b;
    if (!read_all(path, b)) return dh_fail(DH_EIO, std::string("cannot open ") + path);
    uint64_t ix[6];
    if (b.size() < sizeof(ix)) return dh_fail(DH_EIO, "pile-ups db: unexpected end of file (index)");
    memcpy(ix, b.data(), sizeof(ix));
    // every array begins where the previous one ends and all of them lie inside the file
    if (ix[0] != sizeof(ix) || ix[5] != b.size() || ix[1] < ix[0] || ix[2] < ix[1] || ix[3] < ix[2] || ix[4] < ix[3] ||
        ix[5] < ix[4] || (ix[1] - ix[0]) % 16 || (ix[2] - ix[1]) % 16)
        return dh_fail(DH_EIO, "pile-ups db: corrupted index");
    dh_chaindb *d = new dh_chaindb();
    const size_t npiles = (ix[1] - ix[0]) / 16, nra = (ix[2] - ix[1]) / 16;
    uint64_t rp = ix[1], sp = ix[2];
    bool ok = true;
    for (size_t p = 0; p < npiles && ok; p++) {
        ArrSt a;
        memcpy(&a, b.data() + ix[0] + 16 * p, 16);
        ok = a.ptr == rp;
        d->nra_of_pile.push_back((int32_t)a.length);
        rp += 16 * a.length;
    }
    ok = ok && rp == ix[2];
    for (size_t r = 0; r < nra && ok; r++) {
        ArrSt a;
        memcpy(&a, b.data() + ix[1] + 16 * r, 16);
        ok = a.ptr == sp;
        d->nsa_of_ra.push_back((int32_t)a.length);
        sp += sizeof(SeededSt) * a.length;
    }
    ok = ok && sp == ix[3] && parse_chain_blocks(b, ix[2], ix[3], ix[4], ix[5], d);
    if (!ok) {
        delete d;
        return dh_fail(DH_EIO, "pile-ups db: corrupted array pointers");
    }
    *out = d;
    return DH_OK;
}

// DAZZ order (a,c,g,t = 0..3) -> CompressedBase (a=0, c=1, t=2, g=3)
static const uint8_t TO_CB[4] = {0, 1, 3, 2};

extern "C" int dh_insertiondb_write(const char *path, int32_t nins, const dh_insertion_rec *ins, const uint8_t *bases,
                                    const uint32_t *read_ids, const dh_seeded *sa, const dh_chain_la *la, const uint16_t *tp)
{
    if (!path || nins < 0 || (nins > 0 && !ins)) return dh_fail(DH_EINVAL, "dh_insertiondb_write: bad argument");
    int64_t nquads = 0, nsa = 0, nla = 0, ntp = 0, nids = 0, nbases = 0;
    for (int32_t i = 0; i < nins; i++) {
        if (ins[i].seq_len < 0 || ins[i].noverlaps < 0 || ins[i].nread_ids < 0)
            return dh_fail(DH_EINVAL, "dh_insertiondb_write: negative count");
        nquads += (ins[i].seq_len + 3) / 4;
        nbases += ins[i].seq_len;
        nsa += ins[i].noverlaps;
        nids += ins[i].nread_ids;
    }
    if ((nbases > 0 && !bases) || (nids > 0 && !read_ids) || (nsa > 0 && (!sa || !la)))
        return dh_fail(DH_EINVAL, "dh_insertiondb_write: NULL array");
    for (int64_t s = 0; s < nsa; s++) nla += sa[s].nla;
    for (int64_t l = 0; l < nla; l++) ntp += la[l].ntp;
    if (ntp > 0 && !tp) return dh_fail(DH_EINVAL, "dh_insertiondb_write: NULL array");
    uint64_t ix[7];
    ix[0] = sizeof(ix);
    ix[1] = ix[0] + sizeof(InsertionSt) * (uint64_t)nins;
    ix[2] = ix[1] + (uint64_t)nquads;
    ix[3] = ix[2] + sizeof(SeededSt) * (uint64_t)nsa;
    ix[4] = ix[3] + sizeof(LocalSt) * (uint64_t)nla;
    ix[5] = ix[4] + 4ull * (uint64_t)ntp;
    ix[6] = ix[5] + 4ull * (uint64_t)nids;
    FILE *f = fopen(path, "wb");
    if (!f) return dh_fail(DH_EIO, std::string("cannot open ") + path);
    Writer w{f};
    w.put(ix, 7);
    uint64_t qp = ix[1], op = ix[2], ip = ix[5];
    for (int32_t i = 0; i < nins; i++) {
        InsertionSt s;
        memset(&s, 0, sizeof(s));
        s.start.contigId = (uint64_t)ins[i].start_contig;
        s.start.contigPart = ins[i].start_part;
        s.end.contigId = (uint64_t)ins[i].end_contig;
        s.end.contigPart = ins[i].end_part;
        s.baseOffset = 0;
        s.sequenceLength = (uint64_t)ins[i].seq_len;
        s.sequence = ArrSt{qp, (uint64_t)((ins[i].seq_len + 3) / 4)};
        s.contigLength = (uint64_t)ins[i].contig_len;
        s.overlaps = ArrSt{op, (uint64_t)ins[i].noverlaps};
        s.readIds = ArrSt{ip, (uint64_t)ins[i].nread_ids};
        w.put(&s, 1);
        qp += s.sequence.length;
        op += sizeof(SeededSt) * s.overlaps.length;
        ip += 4 * s.readIds.length;
    }
    int64_t at = 0;
    std::vector<uint8_t> quads;
    for (int32_t i = 0; i < nins; i++) {
        quads.assign((size_t)((ins[i].seq_len + 3) / 4), 0);
        for (int64_t x = 0; x < ins[i].seq_len; x++) {
            const uint8_t c = bases[at + x];
            if (c > 3) {
                fclose(f);
                return dh_fail(DH_EINVAL, "dh_insertiondb_write: sequences must be acgt only");
            }
            quads[(size_t)(x >> 2)] |= (uint8_t)(TO_CB[c] << (2 * (x & 3)));
        }
        w.put(quads.data(), quads.size());
        at += ins[i].seq_len;
    }
    write_chain_blocks(w, sa, nsa, la, tp, ix[3], ix[4]);
    w.put(read_ids, (size_t)nids);
    if (fclose(f) != 0) w.ok = false;
    return w.ok ? DH_OK : dh_fail(DH_EIO, std::string("short write to ") + path);
}

extern "C" int dh_insertiondb_read(const char *path, dh_chaindb **out)
{
    if (!path || !out) return dh_fail(DH_EINVAL, "dh_insertiondb_read: NULL argument");
    std::vector<uint8_t> b;
    if (!read_all(path, b)) return dh_fail(DH_EIO, std::string("cannot open ") + path);
    uint64_t ix[7];
    if (b.size() < sizeof(ix)) return dh_fail(DH_EIO, "insertions db: unexpected end of file (index)");
    memcpy(ix, b.data(), sizeof(ix));
    if (ix[0] != sizeof(ix) || ix[6] != b.size() || ix[1] < ix[0] || (ix[1] - ix[0]) % sizeof(InsertionSt) || ix[2] < ix[1] ||
        ix[3] < ix[2] || ix[4] < ix[3] || ix[5] < ix[4] || ix[5] > ix[6] || (ix[6] - ix[5]) % 4)
        return dh_fail(DH_EIO, "insertions db: corrupted index");
    dh_chaindb *d = new dh_chaindb();
    const size_t nins = (ix[1] - ix[0]) / sizeof(InsertionSt);
    static const uint8_t FROM_CB[4] = {0, 1, 3, 2};
    uint64_t qp = ix[1], op = ix[2], ip = ix[5];
    bool ok = true;
    for (size_t i = 0; i < nins && ok; i++) {
        InsertionSt s;
        memcpy(&s, b.data() + ix[0] + i * sizeof(InsertionSt), sizeof(s));
        ok = s.sequence.ptr == qp && s.overlaps.ptr == op && s.readIds.ptr == ip && s.baseOffset < 4 &&
             s.sequence.length * 4 >= s.baseOffset + s.sequenceLength && qp + s.sequence.length <= ix[2] &&
             ip + 4 * s.readIds.length <= ix[6];
        if (!ok) break;
        dh_insertion_rec r;
        memset(&r, 0, sizeof(r));
        r.start_contig = (int64_t)s.start.contigId;
        r.start_part = s.start.contigPart;
        r.end_contig = (int64_t)s.end.contigId;
        r.end_part = s.end.contigPart;
        r.seq_len = (int64_t)s.sequenceLength;
        r.contig_len = (int64_t)s.contigLength;
        r.noverlaps = (int32_t)s.overlaps.length;
        r.nread_ids = (int32_t)s.readIds.length;
        d->ins.push_back(r);
        for (uint64_t x = 0; x < s.sequenceLength; x++) {
            const uint64_t g = s.baseOffset + x;
            d->ins_bases.push_back(FROM_CB[(b[qp + (g >> 2)] >> (2 * (g & 3))) & 3]);
        }
        for (uint64_t x = 0; x < s.readIds.length; x++) {
            uint32_t id;
            memcpy(&id, b.data() + ip + 4 * x, 4);
            d->read_ids.push_back(id);
        }
        qp += s.sequence.length;
        op += sizeof(SeededSt) * s.overlaps.length;
        ip += 4 * s.readIds.length;
    }
    ok = ok && qp == ix[2] && op == ix[3] && ip == ix[6] && parse_chain_blocks(b, ix[2], ix[3], ix[4], ix[5], d);
    if (!ok) {
        delete d;
        return dh_fail(DH_EIO, "insertions db: corrupted array pointers");
    }
    *out = d;
    return DH_OK;
}

// `dentist merge-insertions` (commands/mergeInsertions.d:42-164): the insertions of several insertions.db
// files (one per `process --batch`, snakemake/Snakefile:1315-1334) merged into one, ordered like the
// reference's Insertion.opCmp (util/math.d:527-545 on ContigNode tuples: start contig, start part, end
// contig, end part).  A file that is not sorted is sorted first (ensureSorted :66-72); equal keys keep the
// order of the files (the merger takes the first source whose head is strictly smallest, :120-138).
namespace {
struct InsRef {
    int32_t file, idx;
    int64_t base_off, id_off, sa_off;
};
inline bool ins_less(const dh_insertion_rec &a, const dh_insertion_rec &b)
{
    if (a.start_contig != b.start_contig) return a.start_contig < b.start_contig;
    if (a.start_part != b.start_part) return a.start_part < b.start_part;
    if (a.end_contig != b.end_contig) return a.end_contig < b.end_contig;
    return a.end_part < b.end_part;
}
} // namespace

extern "C" int dh_insertiondb_merge(const char *const *paths, int32_t npaths, const char *out_path, int64_t *ntotal)
{
    if (!out_path || npaths < 0 || (npaths > 0 && !paths)) return dh_fail(DH_EINVAL, "dh_insertiondb_merge: bad argument");
    std::vector<dh_chaindb *> dbs(npaths, nullptr);
    auto cleanup = [&]() { for (auto *d : dbs) delete d; };
    std::vector<std::vector<InsRef>> refs(npaths);
    std::vector<std::vector<int64_t>> la_off(npaths), tp_off(npaths); // per seeded alignment of a file
    int64_t total = 0;
    for (int32_t f = 0; f < npaths; f++) {
        if (!paths[f]) { cleanup(); return dh_fail(DH_EINVAL, "dh_insertiondb_merge: NULL path"); }
        int rc = dh_insertiondb_read(paths[f], &dbs[f]);
        if (rc != DH_OK) { cleanup(); return rc; }
        const dh_chaindb *d = dbs[f];
        int64_t lo = 0, to = 0;
        la_off[f].reserve(d->sa.size() + 1);
        tp_off[f].reserve(d->sa.size() + 1);
        for (size_t s = 0; s < d->sa.size(); s++) {
            la_off[f].push_back(lo);
            tp_off[f].push_back(to);
            for (int32_t l = 0; l < d->sa[s].nla; l++) to += 2ll * d->la[lo + l].ntp;
            lo += d->sa[s].nla;
        }
        la_off[f].push_back(lo);
        tp_off[f].push_back(to);
        int64_t bo = 0, io = 0, so = 0;
        for (size_t i = 0; i < d->ins.size(); i++) {
            refs[f].push_back({f, (int32_t)i, bo, io, so});
            bo += d->ins[i].seq_len;
            io += d->ins[i].nread_ids;
            so += d->ins[i].noverlaps;
        }
        auto less = [d](const InsRef &a, const InsRef &b) { return ins_less(d->ins[a.idx], d->ins[b.idx]); };
        if (!std::is_sorted(refs[f].begin(), refs[f].end(), less)) std::stable_sort(refs[f].begin(), refs[f].end(), less);
        total += (int64_t)d->ins.size();
    }
    if (total > INT32_MAX) { cleanup(); return dh_fail(DH_EINVAL, "dh_insertiondb_merge: more than 2^31 insertions"); }
    std::vector<dh_insertion_rec> ins;
    std::vector<uint8_t> bases;
    std::vector<uint32_t> ids;
    std::vector<dh_seeded> sa;
    std::vector<dh_chain_la> la;
    std::vector<uint16_t> tp;
    ins.reserve(total);
    std::vector<size_t> head(npaths, 0);
    for (;;) {
        int32_t best = -1;
        for (int32_t f = 0; f < npaths; f++) {
            if (head[f] >= refs[f].size()) continue;
            if (best < 0 || ins_less(dbs[f]->ins[refs[f][head[f]].idx], dbs[best]->ins[refs[best][head[best]].idx])) best = f;
        }
        if (best < 0) break;
        const InsRef &r = refs[best][head[best]++];
        const dh_chaindb *d = dbs[best];
        const dh_insertion_rec &rec = d->ins[r.idx];
        ins.push_back(rec);
        bases.insert(bases.end(), d->ins_bases.begin() + r.base_off, d->ins_bases.begin() + r.base_off + rec.seq_len);
        ids.insert(ids.end(), d->read_ids.begin() + r.id_off, d->read_ids.begin() + r.id_off + rec.nread_ids);
        sa.insert(sa.end(), d->sa.begin() + r.sa_off, d->sa.begin() + r.sa_off + rec.noverlaps);
        const int64_t l0 = la_off[best][r.sa_off], l1 = la_off[best][r.sa_off + rec.noverlaps];
        const int64_t t0 = tp_off[best][r.sa_off], t1 = tp_off[best][r.sa_off + rec.noverlaps];
        la.insert(la.end(), d->la.begin() + l0, d->la.begin() + l1);
        tp.insert(tp.end(), d->tp.begin() + t0, d->tp.begin() + t1);
    }
    cleanup();
    if (ntotal) *ntotal = total;
    return dh_insertiondb_write(out_path, (int32_t)ins.size(), ins.data(), bases.data(), ids.data(), sa.data(), la.data(),
                                tp.data());
}
