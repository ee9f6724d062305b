// sim.cpp -- seeded synthetic assembly / gap / read generator (plain C ABI, no GPU).
//
// Stands in for DAZZ_DB's `simulator` (absent here), which the reference's tests call as
// `simulator -m25000 -s12500 -e.13 -c20 -r<seed>` (tests/test-commands.sh:7-13, 102-105) and
// whose role in the BASELINE configs is spelled out in SURVEY.md section 8(d): i.i.d. ACGT
// assemblies, log-uniform gaps, fixed-length or log-normal reads, PacBio-like error profile
// (13 %: ins .60 / del .25 / sub .15).  Every read is generated from its own counter-based
// RNG stream (splitmix64 of seed and read index), so generation order and thread count do not
// change the data.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(uni() * (double)n); }
    double normal()
    {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};
}  // namespace

extern "C" {

// i.i.d. uniform bases, codes 0..3
void dhsim_genome(uint64_t seed, int64_t n, uint8_t *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t blk = 0; blk < (n + 4095) / 4096; blk++) {
        Rng r(seed * 0x100000001B3ull + (uint64_t)blk);
        int64_t lo = blk * 4096, hi = lo + 4096 < n ? lo + 4096 : n;
        for (int64_t i = lo; i < hi; i += 32) {
            uint64_t w = r.next();
            for (int64_t j = i; j < hi && j < i + 32; j++, w >>= 2) out[j] = (uint8_t)(w & 3);
        }
    }
}

// ngaps gaps, length log-uniform in [minlen, maxlen], at least `spacing` apart and from the
// ends; begin[]/end[] sorted ascending.  Returns number of gaps placed.
int32_t dhsim_gaps(uint64_t seed, int64_t genome_len, int32_t ngaps, int32_t minlen, int32_t maxlen,
                   int64_t spacing, int64_t *begin, int64_t *end)
{
    Rng r(seed);
    // one gap per equal slot keeps the spacing guarantee without rejection sampling
    const int64_t slot = genome_len / (ngaps + 1);
    int32_t n = 0;
    for (int32_t g = 0; g < ngaps; g++) {
        double lg = std::log((double)minlen) + r.uni() * (std::log((double)maxlen) - std::log((double)minlen));
        int64_t len = (int64_t)std::exp(lg);
        if (len < minlen) len = minlen;
        if (len > maxlen) len = maxlen;
        int64_t centre = slot * (g + 1);
        int64_t jitter = slot - spacing - len;
        if (jitter < 0) jitter = 0;
        int64_t b = centre - jitter / 2 + (int64_t)r.below((uint64_t)jitter + 1) - len / 2;
        if (b < spacing) b = spacing;
        if (b + len > genome_len - spacing) continue;
        if (n > 0 && b < end[n - 1] + spacing) continue;
        begin[n] = b;
        end[n] = b + len;
        n++;
    }
    return n;
}

// Reads sampled from genome[0, glen).  mean_len/sd_len: sd_len == 0 -> fixed length, otherwise
// log-normal with that mean and sd (DAZZ simulator's -m/-s), minimum min_len.
// out_bases must hold nreads * max_out bytes; read r is written at out_off[r] (filled here,
// contiguous) -- call once with out_bases == NULL to size (returns total bytes needed).
// truth[r*3+0..2] = start, end, strand.
int64_t dhsim_reads_from(uint64_t seed, const uint8_t *genome, int64_t glen, int32_t first, int32_t nreads,
                         int32_t mean_len, int32_t sd_len, int32_t min_len, double err, double p_ins,
                         double p_del, int64_t *out_off, uint8_t *out_bases, int64_t *truth);
int64_t dhsim_reads_sel(uint64_t seed, const uint8_t *genome, int64_t glen, const int64_t *ids, int32_t first,
                        int32_t nreads, int32_t mean_len, int32_t sd_len, int32_t min_len, double err, double p_ins,
                        double p_del, double hp_bias, int64_t *out_off, uint8_t *out_bases, int64_t *truth);
int64_t dhsim_reads(uint64_t seed, const uint8_t *genome, int64_t glen, int32_t nreads,
                    int32_t mean_len, int32_t sd_len, int32_t min_len, double err, double p_ins,
                    double p_del, int64_t *out_off, uint8_t *out_bases, int64_t *truth)
{
    return dhsim_reads_from(seed, genome, glen, 0, nreads, mean_len, sd_len, min_len, err, p_ins, p_del, out_off,
                            out_bases, truth);
}

// the reads [first, first + nreads) of the same stream: a rank of a sharded run generates only its
// share, identical to the corresponding reads of the whole set
int64_t dhsim_reads_from(uint64_t seed, const uint8_t *genome, int64_t glen, int32_t first, int32_t nreads,
                         int32_t mean_len, int32_t sd_len, int32_t min_len, double err, double p_ins,
                         double p_del, int64_t *out_off, uint8_t *out_bases, int64_t *truth)
{
    return dhsim_reads_sel(seed, genome, glen, nullptr, first, nreads, mean_len, sd_len, min_len, err, p_ins, p_del, 0.0,
                           out_off, out_bases, truth);
}

// where every read of the stream comes from (start, end, strand) without generating its bases: the
// first draws of a read's RNG stream.  Lets a sharded run find the reads of other shards that reach
// its gaps (they are then generated with dhsim_reads_sel).
void dhsim_read_truth(uint64_t seed, int64_t glen, int64_t first, int64_t nreads, int32_t mean_len, int32_t sd_len,
                      int32_t min_len, int64_t *truth)
{
    const double mu = sd_len > 0 ? std::log((double)mean_len * mean_len /
                                            std::sqrt((double)mean_len * mean_len + (double)sd_len * sd_len))
                                 : 0.0;
    const double sg = sd_len > 0 ? std::sqrt(std::log(1.0 + ((double)sd_len * sd_len) / ((double)mean_len * mean_len))) : 0.0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nreads; r++) {
        Rng g(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(first + r + 1)));
        int64_t len = mean_len;
        if (sd_len > 0) {
            len = (int64_t)std::exp(mu + sg * g.normal());
            if (len < min_len) len = min_len;
        }
        if (len > glen) len = glen;
        const int64_t start = (int64_t)g.below((uint64_t)(glen - len + 1));
        truth[r * 3 + 0] = start;
        truth[r * 3 + 1] = start + len;
        truth[r * 3 + 2] = (int64_t)(g.next() & 1);
    }
}

// the general form: reads ids[0 .. nreads) of the stream (ids == NULL: first .. first + nreads), and
// hp_bias > 0 = homopolymer-biased indels (the ONT-like profile of BASELINE configs[4]):
// inside a homopolymer run of the template the insertion and deletion rates grow by hp_bias per base
// of the run so far (capped at 5), half of the inserted bases there repeat the run's base, and the
// rates are scaled so that the expected error rate stays `err` on an i.i.d. template.
int64_t dhsim_reads_sel(uint64_t seed, const uint8_t *genome, int64_t glen, const int64_t *ids, int32_t first,
                        int32_t nreads, int32_t mean_len, int32_t sd_len, int32_t min_len, double err, double p_ins,
                        double p_del, double hp_bias, int64_t *out_off, uint8_t *out_bases, int64_t *truth)
{
    // E[min(run - 1, 5)] of an i.i.d. 4-letter template = sum_{k>=1} P(run >= k + 1) for k <= 5
    const double hp_mean = 0.25 + 0.0625 + 0.015625 + 0.00390625 + 0.0009765625;
    const double hp_norm = hp_bias > 0 ? 1.0 / (1.0 + hp_bias * hp_mean * (p_ins + p_del)) : 1.0;
    // pass 1: lengths of every read's output (sequential prefix sum needs them)
    std::vector<int32_t> outlen((size_t)nreads);
    const double mu = sd_len > 0 ? std::log((double)mean_len * mean_len /
                                            std::sqrt((double)mean_len * mean_len + (double)sd_len * sd_len))
                                 : 0.0;
    const double sg = sd_len > 0 ? std::sqrt(std::log(1.0 + ((double)sd_len * sd_len) /
                                                                ((double)mean_len * mean_len)))
                                 : 0.0;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            int64_t o = 0;
            for (int32_t r = 0; r < nreads; r++) {
                out_off[r] = o;
                o += outlen[(size_t)r];
            }
            out_off[nreads] = o;
            if (!out_bases) return o;
        }
#pragma omp parallel for schedule(dynamic, 256)
        for (int32_t r = 0; r < nreads; r++) {
            Rng g(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)((ids ? ids[r] : (int64_t)first + r) + 1)));
            int64_t len = mean_len;
            if (sd_len > 0) {
                len = (int64_t)std::exp(mu + sg * g.normal());
                if (len < min_len) len = min_len;
            }
            if (len > glen) len = glen;
            const int64_t start = (int64_t)g.below((uint64_t)(glen - len + 1));
            const int strand = (int)(g.next() & 1);
            uint8_t *dst = (pass == 1) ? out_bases + out_off[r] : nullptr;
            int32_t n = 0;
            int64_t p = 0;
            int32_t run = 0;
            uint8_t prev = 255;
            int64_t prev_p = -1;
            while (p < len) {
                // true base in read orientation
                const uint8_t tb = strand ? (uint8_t)(3 - genome[start + len - 1 - p]) : genome[start + p];
                double e_ins = err * p_ins, e_indel = err * (p_ins + p_del), e_all = err;
                bool in_run = false;
                if (hp_bias > 0) {
                    if (p != prev_p) {  // a new template position: length of the homopolymer run ending here
                        run = tb == prev ? run + 1 : 1;
                        prev = tb;
                        prev_p = p;
                    }
                    const double m = 1.0 + hp_bias * (double)(run - 1 < 5 ? run - 1 : 5);
                    in_run = run > 1;
                    e_ins = err * hp_norm * p_ins * m;
                    e_indel = e_ins + err * hp_norm * p_del * m;
                    e_all = e_indel + err * hp_norm * (1.0 - p_ins - p_del);
                }
                const double u = g.uni();
                if (u < e_ins) {  // insertion: emit a random base (inside a run: every other time the run's base), do not consume
                    const uint8_t b = (in_run && g.uni() < 0.5) ? tb : (uint8_t)(g.next() & 3);
                    if (dst) dst[n] = b;
                    n++;
                } else if (u < e_indel) {  // deletion
                    p++;
                } else if (u < e_all) {  // substitution by one of the three other bases
                    const uint8_t b = (uint8_t)((tb + 1 + g.below(3)) & 3);
                    if (dst) dst[n] = b;
                    n++;
                    p++;
                } else {
                    if (dst) dst[n] = tb;
                    n++;
                    p++;
                }
            }
            if (pass == 0) {
                outlen[(size_t)r] = n;
                if (truth) {
                    truth[(size_t)r * 3 + 0] = start;
                    truth[(size_t)r * 3 + 1] = start + len;
                    truth[(size_t)r * 3 + 2] = strand;
                }
            }
        }
    }
    return out_off[nreads];
}

}  // extern "C"
