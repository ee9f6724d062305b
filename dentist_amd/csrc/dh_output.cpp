// dh_output.cpp -- host-side writer of the gap-closed assembly.
//
// Restates the linear-scaffold subset of `dentist output` that applies to gap closing inside
// existing scaffolds: header rule source/dentist/commands/output.d:743-759, contig slices
// :782-835, unclosed gaps as 'n' runs :837-862, upper-cased insertions :864-925, closed-gaps BED
// :879-891, line wrapping :232 (fastaLineWidth, commandline.d:1697-1699); splice coordinates are
// the dh_insertion fields (common/insertions.d:110-146).  No device work.
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

// contigs: base codes + offsets (ncontigs + 1); scaffold_of[c] = index of the input scaffold contig
// c belongs to (contigs of one scaffold are consecutive); headers[s] = FASTA header of input
// scaffold s without '>' (the id is cut at the first tab); gap_len[c] = length of the gap between
// contig c and c + 1 inside a scaffold (ignored at scaffold ends).  ins / ins_bases: the result of
// dh_process_pileups; a gap is closed by the insertion with status == DH_PILE_OK whose
// contig_left is c.  highlight != 0 upper-cases inserted bases.  bed_path may be NULL.
extern "C" int dh_output_fasta(const char *fasta_path, const char *bed_path, const uint8_t *contig_bases,
                               const int64_t *contig_off, int32_t ncontigs, const int32_t *scaffold_of,
                               const char *const *headers, const int32_t *gap_len, const dh_insertion *ins,
                               int32_t nins, const uint8_t *ins_bases, int32_t line_width, int32_t highlight)
{
    if (!fasta_path || !contig_bases || !contig_off || !scaffold_of || !headers || (nins > 0 && (!ins || !ins_bases)) ||
        ncontigs < 0)
        return dh_fail(DH_EINVAL, "dh_output_fasta: NULL argument");
    std::vector<int32_t> closing((size_t)std::max(ncontigs, 1), -1);
    for (int32_t i = 0; i < nins; i++) {
        if (ins[i].status != DH_PILE_OK) continue;
        const int32_t c = ins[i].contig_left;
        if (c < 0 || c + 1 >= ncontigs || scaffold_of[c] != scaffold_of[c + 1])
            return dh_fail(DH_EINVAL, "dh_output_fasta: insertion does not join two contigs of one scaffold");
        if (closing[(size_t)c] >= 0) return dh_fail(DH_EINVAL, "dh_output_fasta: two insertions for one gap");
        closing[(size_t)c] = i;
    }
    FILE *f = fopen(fasta_path, "w");
    if (!f) return dh_fail(DH_EIO, std::string("cannot open ") + fasta_path);
    FILE *bed = nullptr;
    if (bed_path) {
        bed = fopen(bed_path, "w");
        if (!bed) {
            fclose(f);
            return dh_fail(DH_EIO, std::string("cannot open ") + bed_path);
        }
    }
    LineWriter w{f, line_width};
    bool ok = true;
    int32_t c = 0;
    while (c < ncontigs) {
        const int32_t s = scaffold_of[c];
        std::string id(headers[s] ? headers[s] : "");
        const size_t tab = id.find('\t');
        if (tab != std::string::npos) id.resize(tab);
        // one output scaffold per input scaffold: the uniquified id is the id itself
        ok = ok && fprintf(f, ">%s\tscaffold-%d\n", id.c_str(), c + 1) > 0;
        int64_t coord = 1;  // 1-based scaffold coordinate of the next base (output.d currentScaffoldCoord)
        int32_t from = 0;   // the current contig is kept from here
        for (;; c++) {
            const int64_t clen = contig_off[c + 1] - contig_off[c];
            const bool last = c + 1 >= ncontigs || scaffold_of[c + 1] != s;
            const int32_t ci = last ? -1 : closing[(size_t)c];
            const int64_t to = ci >= 0 ? ins[ci].left_aepos : clen;
            if (from > to || to > clen) {
                fclose(f);
                if (bed) fclose(bed);
                return dh_fail(DH_EINVAL, "dh_output_fasta: splice coordinates outside the contig");
            }
            for (int64_t x = from; x < to; x++) {
                const uint8_t b = contig_bases[contig_off[c] + x];
                w.put(LOWER[b < 4 ? b : 4]);
            }
            coord += to - from;
            from = 0;
            if (last) break;
            if (ci >= 0) {
                const dh_insertion &in = ins[ci];
                if (in.ins_begin < 0 || in.ins_begin > in.ins_end || in.ins_end > in.cons_len || in.cons_off < 0) {
                    fclose(f);
                    if (bed) fclose(bed);
                    return dh_fail(DH_EINVAL, "dh_output_fasta: insertion outside its consensus");
                }
                const uint8_t *cons = ins_bases + in.cons_off;
                const int64_t n = (int64_t)in.ins_end - in.ins_begin;
                for (int64_t x = 0; x < n; x++) {
                    // oriented consensus: reverse complement of the stored sequence when comp is set
                    const int64_t p = in.ins_begin + x;
                    uint8_t b = in.comp ? cons[in.cons_len - 1 - p] : cons[p];
                    if (in.comp && b < 4) b = (uint8_t)(3 - b);
                    w.put((highlight ? UPPER : LOWER)[b < 4 ? b : 4]);
                }
                // output.d:879-891: currentScaffoldCoord - 1 and nextScaffoldCoord (= current + length)
                if (bed)
                    ok = ok && fprintf(bed, "%s\t%lld\t%lld\tcontigs-%d-%d|reads-%d\n", id.c_str(),
                                       (long long)(coord - 1), (long long)(coord + n), c + 1, c + 2,
                                       in.ref_read_id + 1) > 0;
                coord += n;
                from = in.right_abpos;
            } else {
                const int32_t g = gap_len ? gap_len[c] : 0;
                for (int32_t x = 0; x < g; x++) w.put('n');
                coord += g;
            }
        }
        w.end_record();
        c++;
    }
    ok = ok && w.ok;
    if (fclose(f) != 0) ok = false;
    if (bed && fclose(bed) != 0) ok = false;
    return ok ? DH_OK : dh_fail(DH_EIO, std::string("short write to ") + fasta_path);
}
