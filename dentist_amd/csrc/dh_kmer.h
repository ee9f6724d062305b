// dh_kmer.h -- device helpers shared by the seed kernels (dh_kernels.hip) and the pile-up k-mer join (dh_join.hip):
// modimer sampling on canonical k-mers, unaligned 8-byte loads, soft-mask test of a k-mer.
#ifndef DH_KMER_H
#define DH_KMER_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// modimer sampling (daligner -%): the same k-mers are kept on the A and on the B side, decided on the
// CANONICAL k-mer (the smaller of a k-mer and its reverse complement, both rolled along), so that a
// k-mer and its reverse complement are sampled together.
// Keep a k-mer iff h % mod == 0 with h = bits 32..63 of canon * 0x9E3779B97F4A7C15.  This test is evaluated for every
// base of every read, so it is arranged to need three multiplies (two when k <= 16) and no division.  (On gfx950
// v_mul_lo_u32 / v_mul_hi_u32 issue at the rate of a shift -- scripts/valu_probe.cpp, profiles/r03_valu_probe.txt: 0.56 G
// wave-instructions/s per SIMD for both -- not at a quarter of it: replacing two of the three by 24-bit multiplies for
// power-of-two mods, round 5, changed nothing: 56.4 against 56.3 ms of seeds per step.)
//  * h = mulhi(lo, C_lo) + lo * C_hi + hi * C_lo   (lo / hi = halves of the k-mer);
//  * h % mod == 0  <=>  rotr(h * inv(mod'), e) <= (2^32 - 1) / mod   for mod = mod' * 2^e, mod'
//    odd, inv = inverse of mod' modulo 2^32 (test for zero remainder, Hacker's Delight 10-17).
struct KmerSampler {
    uint32_t inv, thresh, rot;
    bool all, small_k, pow2;  // pow2: mod is a power of two -- h % mod == 0 is a mask test, one multiply less
};
__device__ __forceinline__ KmerSampler kmer_sampler(int32_t mod, int32_t k)
{
    KmerSampler s;
    s.all = mod <= 1;
    s.small_k = k <= 16;
    uint32_t d = s.all ? 1u : (uint32_t)mod, e = 0;
    while ((d & 1u) == 0u) {
        d >>= 1;
        e++;
    }
    uint32_t x = d;  // Newton: x <- x * (2 - d * x) doubles the number of correct low bits
    for (int it = 0; it < 5; it++) x *= 2u - d * x;
    s.inv = x;
    s.rot = e;
    s.pow2 = d == 1u;
    s.thresh = s.all ? 0xFFFFFFFFu : 0xFFFFFFFFu / (uint32_t)mod;
    return s;
}
__device__ __forceinline__ bool kmer_sampled(uint64_t km, const KmerSampler &s)
{
    if (s.all) return true;  // (uniform: no sampling, no hash)
    const uint32_t lo = (uint32_t)km, hi = (uint32_t)(km >> 32);
    uint32_t h = __umulhi(lo, 0x7F4A7C15u) + lo * 0x9E3779B9u;
    if (!s.small_k) h += hi * 0x7F4A7C15u;
    if (s.pow2) return (h & ((1u << s.rot) - 1u)) == 0u;  // (uniform branch)
    const uint32_t t = h * s.inv;
    const uint32_t r = __builtin_rotateright32(t, s.rot);  // (rot == 0: t itself)
    return s.all | (r <= s.thresh);
}

__device__ __forceinline__ uint64_t load8(const uint8_t *p)
{
    uint64_t x;
    __builtin_memcpy(&x, p, 8);  // unaligned global_load_dwordx2 (DB buffers are padded)
    return x;
}

// ---- soft masks (daligner / damapper -m<track>, DBdust): one bit per base of the concatenated base
// array (DbView.mask_bits, bit g = base g is masked).  A k-mer is neither indexed nor looked up when
// it touches a masked base: k <= 28 bits read with one unaligned 8-byte load.
__device__ __forceinline__ bool mask_touch(const uint8_t *__restrict__ bits, int64_t g, int32_t k)
{
    const uint64_t w = load8(bits + (g >> 3)) >> (g & 7);
    return (w & ((1ull << k) - 1ull)) != 0ull;
}

#endif
