// tile_host.cpp -- the DH-2 lane state machine of dentist_amd/csrc/dh_tile.h compiled for the CPU.
//
// TEST INFRASTRUCTURE: lets `-m "not gpu"` tests compare the very code k_tile runs (work fetch,
// candidate loop, tile set-up, bit-vector column step, scan, trace pairs, records) with the oracle's
// plain-DP restatement (oracle/align.c: extend_tiled).  Lanes run one after the other here; on the
// device 64 of them share a wavefront (dentist_amd/csrc/dh_tile.hip).
// Build: g++ -O2 -shared -fPIC -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ (Makefile target tests/native/libdh_tile_host.so)
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../dentist_amd/csrc/dh_tile.h"

using namespace dhtile;

static void pack2(const uint8_t *src, int64_t n, std::vector<uint32_t> &dst, int64_t pad_words)
{
    dst.assign((size_t)((n + 15) / 16 + 2 * pad_words), 0u);
    for (int64_t g = 0; g < n; g++) dst[(size_t)(pad_words + (g >> 4))] |= (uint32_t)(src[g] & 3u) << (2 * (g & 15));
}
static void planes(const uint8_t *src, int64_t n, std::vector<PlanePair> &dst, int64_t pad_words)
{
    dst.assign((size_t)((n + 31) / 32 + 2 * pad_words), PlanePair{0u, 0u});
    for (int64_t g = 0; g < n; g++) {
        PlanePair &p = dst[(size_t)(pad_words + (g >> 5))];
        p.x |= (uint32_t)(src[g] & 1u) << (g & 31);
        p.y |= (uint32_t)((src[g] >> 1) & 1u) << (g & 31);
    }
}
static void revcomp_all(const uint8_t *src, const int64_t *off, int32_t n, std::vector<uint8_t> &dst)
{
    dst.resize((size_t)off[n]);
    for (int32_t s = 0; s < n; s++)
        for (int64_t i = off[s], j = off[s + 1] - 1; i < off[s + 1]; i++, j--) dst[(size_t)i] = (uint8_t)(3 - (src[j] & 3u));
}

// every item (read * 2 + strand) of B against its candidates; results compacted in item order.
// out_la: capacity nitems * max_la; out_trace: capacity cap_trace u16; returns the number of records or < 0
// out_la2 / out_trace2 / n2 (optional): the transposed records of the mapping (`damapper -C`), compacted in item order

// one tile of the lane's running extension: set-up, the columns in sequence, the end of the tile
template <int WB>
static void run_tile(Lane &l, const Params &P)
{
    TileT<WB> t;
    uint32_t q[NTW];
    tile_setup<WB>(l, P, t, q);
    for (int32_t c = 1; c <= t.cols; c++) {
        typename BandVec<WB>::U p0, p1;
        uint32_t x;
        tile_window<WB>(q, c, p0, p1, x);
        if (P.tandem)
            tile_col<true, WB>(t, p0, p1, x);
        else
            tile_col<false, WB>(t, p0, p1, x);
    }
    tile_end<WB>(l, P, t);
}

extern "C" long dh_tile_host_align2(const uint8_t *abases, const int64_t *aoff, int32_t na, const uint8_t *bbases,
                                    const int64_t *boff, int32_t nb, const DhOpts *o, const DhCand *cand,
                                    const int32_t *ncand, int32_t nbmax, DhLa *out_la, uint16_t *out_trace,
                                    long cap_trace, unsigned long long *counters, DhLa *out_la2, uint16_t *out_trace2,
                                    long *n2)
{
    const int64_t PADW = 8;
    std::vector<uint8_t> arc, brc;
    revcomp_all(abases, aoff, na, arc);
    revcomp_all(bbases, boff, nb, brc);
    std::vector<uint32_t> apk, arcpk;
    std::vector<PlanePair> bpp, brcpp;
    pack2(abases, aoff[na], apk, PADW);
    pack2(arc.data(), aoff[na], arcpk, PADW);
    planes(bbases, boff[nb], bpp, PADW);
    planes(brc.data(), boff[nb], brcpp, PADW);
    // the copies of the transposed pairs: 2-bit packed B, plane-packed A
    std::vector<uint32_t> bpk, brcpk;
    std::vector<PlanePair> app, arcpp;
    pack2(bbases, boff[nb], bpk, PADW);
    pack2(brc.data(), boff[nb], brcpk, PADW);
    planes(abases, aoff[na], app, PADW);
    planes(arc.data(), aoff[na], arcpp, PADW);
    const int32_t nitems = 2 * nb, trmax = 2 * (2 * nbmax + 2);
    // symmetric mode: records in candidate-indexed slots (two per candidate), as on the device
    const bool sym = o->skip_self == 2;
    std::vector<int32_t> candoff((size_t)nitems + 1, 0);
    for (int32_t it = 0; it < nitems; it++) candoff[(size_t)it + 1] = candoff[(size_t)it] + (ncand[it] > 0 ? ncand[it] : 0);
    const size_t nslots = sym ? 2 * (size_t)candoff[(size_t)nitems] : (size_t)nitems * o->max_la;
    std::vector<DhLa> slots(nslots);
    memset(slots.data(), 0, sizeof(DhLa) * nslots);
    std::vector<uint16_t> tr(nslots * trmax, 0);
    std::vector<int32_t> nla((size_t)nitems, 0), ntr((size_t)nitems, 0), regs((size_t)MAXREG * REGF, 0);
    uint32_t queue = 0;
    int32_t status = 0;
    Params P = {};
    P.aoff = aoff;
    P.boff = boff;
    P.apk = apk.data() + PADW;
    P.arcpk = arcpk.data() + PADW;
    P.bpp = bpp.data() + PADW;
    P.brcpp = brcpp.data() + PADW;
    P.apk1 = bpk.data() + PADW;
    P.arcpk1 = brcpk.data() + PADW;
    P.bpp1 = app.data() + PADW;
    P.brcpp1 = arcpp.data() + PADW;
    std::vector<DhLa> slots2(out_la2 ? (size_t)2 * nb * o->max_la : 0);
    std::vector<uint16_t> tr2(out_la2 ? (size_t)2 * nb * o->max_la * (2 * (2 * nbmax + 2)) : 0, 0);
    std::vector<int32_t> nla2((size_t)2 * nb, 0), ntr2((size_t)2 * nb, 0);
    P.out_la2 = out_la2 ? slots2.data() : nullptr;
    P.out_trace2 = out_la2 ? tr2.data() : nullptr;
    P.out_nla2 = nla2.data();
    P.out_ntr2 = ntr2.data();
    P.o = *o;
    P.item0 = 0;
    P.nitems = nitems;
    P.cand = cand;
    P.ncand = ncand;
    P.queue = &queue;
    P.book_min = 1;
    P.units = nullptr;
    P.nunits = nullptr;
    P.candoff = sym ? candoff.data() : nullptr;
    P.regs = regs.data();
    P.nbmax = nbmax;
    P.trmax = trmax;
    P.out_la = slots.data();
    P.out_trace = tr.data();
    P.out_nla = nla.data();
    P.out_ntr = ntr.data();
    P.counters = counters;
    P.status = &status;
    Lane l;
    Cold cold;
    memset(&cold, 0, sizeof(cold));
    P.cold = &cold;
    lane_init(l, 0, &cold);
    for (;;) {
        while (l.st != L_RUN && l.st != L_DONE) {
            if (l.st == L_EXT_END)
                lane_ext_end(l, P);
            else if (l.st == L_CAND)
                lane_next_cand(l, P);
            else {  // L_FETCH
                const int32_t it = (int32_t)queue++;
                if (it >= nitems)
                    l.st = L_DONE;
                else
                    lane_fetch(l, P, it);
            }
        }
        if (l.st == L_DONE) break;
        if (P.o.width == 32)
            run_tile<32>(l, P);
        else
            run_tile<64>(l, P);
    }
    if (l.err) return -(long)l.err;
    counters[0] = l.cells;
    counters[1] = l.naln;
    long n = 0, tn = 0;
    if (sym) {
        for (size_t x = 0; x < nslots; x++) {
            DhLa la = slots[x];
            if (la.pad != 1) continue;
            const uint16_t *src = tr.data() + x * trmax + la.toff;
            if (tn + la.tlen > cap_trace) return -100;
            memcpy(out_trace + tn, src, sizeof(uint16_t) * (size_t)la.tlen);
            la.toff = tn;
            la.pad = 0;
            tn += la.tlen;
            out_la[n++] = la;
        }
    } else
    for (int32_t it = 0; it < nitems; it++)
        for (int32_t s = 0; s < nla[(size_t)it]; s++) {
            DhLa la = slots[(size_t)it * o->max_la + s];
            const uint16_t *src = tr.data() + ((size_t)it * o->max_la + s) * trmax + la.toff;
            if (tn + la.tlen > cap_trace) return -100;
            memcpy(out_trace + tn, src, sizeof(uint16_t) * (size_t)la.tlen);
            la.toff = tn;
            tn += la.tlen;
            out_la[n++] = la;
        }
    if (out_la2) {
        long m = 0, tm = 0;
        for (int32_t it = 0; it < nitems; it++)
            for (int32_t s = 0; s < nla2[(size_t)it]; s++) {
                DhLa la = slots2[(size_t)it * o->max_la + s];
                const uint16_t *src = tr2.data() + ((size_t)it * o->max_la + s) * trmax + la.toff;
                if (tm + la.tlen > cap_trace) return -100;
                memcpy(out_trace2 + tm, src, sizeof(uint16_t) * (size_t)la.tlen);
                la.toff = tm;
                tm += la.tlen;
                out_la2[m++] = la;
            }
        *n2 = m;
    }
    return n;
}

extern "C" long dh_tile_host_align(const uint8_t *abases, const int64_t *aoff, int32_t na, const uint8_t *bbases,
                                   const int64_t *boff, int32_t nb, const DhOpts *o, const DhCand *cand,
                                   const int32_t *ncand, int32_t nbmax, DhLa *out_la, uint16_t *out_trace,
                                   long cap_trace, unsigned long long *counters)
{
    return dh_tile_host_align2(abases, aoff, na, bbases, boff, nb, o, cand, ncand, nbmax, out_la, out_trace, cap_trace, counters,
                               nullptr, nullptr, nullptr);
}
