// valu_probe.cpp -- the VALU issue ceiling of this part, measured (round-2 verdict, task 2a).
//
// k_wave2's "VALU-issue bound" claim rested on an assumed 4 cycles per wave64 VALU instruction;
// /opt/skills/guides/MI355X_MICROARCH.md:52-53 states 2 cycles (SIMD-32).  This probe measures, per
// instruction class the alignment kernels use, the wave-instructions per second one SIMD sustains with
// W = 1..8 resident waves per SIMD, each wave running 8 independent dependency chains (so that the
// dependent-issue latency does not limit a single wave), and prints
//     op, waves/SIMD, G wave-instr/s per SIMD, cycles per instruction at the measured shader clock.
// Build and run on the GPU box:
//     hipcc -O3 --offload-arch=gfx950 -o scripts/valu_probe scripts/valu_probe.cpp && scripts/valu_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                 \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

#define ITERS 4096
#define CHAINS 8
#define UNROLL 4  // instructions per chain per loop iteration

// one asm statement = the instruction applied to the 8 chains; X(i) names chain i's register(s)
#define OP8(T)                                                                                       \
    asm volatile(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)                                             \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), \
                   "+v"(r[7])                                                                        \
                 : "v"(s), "v"(q)                                                                    \
                 : "vcc")
#define OP8_64(T)                                                                                    \
    asm volatile(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)                                             \
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), \
                   "+v"(w[7])                                                                        \
                 : "v"(s), "v"(q)                                                                    \
                 : "vcc")

#define T_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define T_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 1\n"
#define T_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define T_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define T_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define T_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define T_MAX(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define T_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define T_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define T_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define T_XNOR(i) "v_xnor_b32 %" #i ", %" #i ", %8\n"
#define T_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_DPPROW(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CMP(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"
#define T_SUBREV(i) "v_subrev_u32 %" #i ", %8, %" #i "\n"
#define T_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define T_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define T_SAD(i) "v_sad_u32 %" #i ", %" #i ", %8, %9\n"
// 64-bit
#define T_ADD64(i) "v_add_co_u32 %L" #i ", vcc, %L" #i ", %8\nv_addc_co_u32 %H" #i ", vcc, %H" #i ", %9, vcc\n"
#define T_LSHR64(i) "v_lshrrev_b64 %" #i ", 1, %" #i "\n"
#define T_LSHL64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define T_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"

enum {
    K_ADD, K_AND, K_XOR, K_LSHL, K_ALIGNBIT, K_BFI, K_OR3, K_ANDOR, K_ADD3, K_LSHLADD, K_MAX, K_CNDMASK, K_BCNT,
    K_FFBL, K_NOT, K_XNOR, K_DPP, K_DPPROW, K_ADDDPP, K_CMP, K_MULLO, K_MUL24, K_PERM, K_BPERM, K_SUBREV, K_MINU,
    K_MED3, K_SAD, K_PKADD, K_LSHR64, K_LSHL64, K_N
};
static const char *names[K_N] = {
    "v_add_u32", "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_alignbit_b32", "v_bfi_b32", "v_or3_b32", "v_and_or_b32",
    "v_add3_u32", "v_lshl_add_u32", "v_max_i32", "v_cndmask_b32", "v_bcnt_u32_b32", "v_ffbl_b32", "v_not_b32",
    "v_xnor_b32", "v_mov_dpp quad_perm", "v_mov_dpp row_shr", "v_add_u32_dpp row_shr", "v_cmp_lt_i32", "v_mul_lo_u32",
    "v_mul_u32_u24", "v_perm_b32", "ds_bpermute_b32", "v_subrev_u32", "v_min_u32", "v_med3_i32", "v_sad_u32",
    "v_pk_add_u16", "v_lshrrev_b64", "v_lshlrev_b64"};

template <int K>
__global__ void __launch_bounds__(64) k_probe(uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    uint32_t r[CHAINS];
    uint64_t w[CHAINS];
    const uint32_t s = seed * 2654435761u + threadIdx.x, q = seed ^ (threadIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
        r[i] = s + i * 977u;
        w[i] = ((uint64_t)(q + i) << 32) | (s + i * 13u);
    }
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (K == K_ADD) OP8(T_ADD);
            if (K == K_AND) OP8(T_AND);
            if (K == K_XOR) OP8(T_XOR);
            if (K == K_LSHL) OP8(T_LSHL);
            if (K == K_ALIGNBIT) OP8(T_ALIGNBIT);
            if (K == K_BFI) OP8(T_BFI);
            if (K == K_OR3) OP8(T_OR3);
            if (K == K_ANDOR) OP8(T_ANDOR);
            if (K == K_ADD3) OP8(T_ADD3);
            if (K == K_LSHLADD) OP8(T_LSHLADD);
            if (K == K_MAX) OP8(T_MAX);
            if (K == K_CNDMASK) OP8(T_CNDMASK);
            if (K == K_BCNT) OP8(T_BCNT);
            if (K == K_FFBL) OP8(T_FFBL);
            if (K == K_NOT) OP8(T_NOT);
            if (K == K_XNOR) OP8(T_XNOR);
            if (K == K_DPP) OP8(T_DPP);
            if (K == K_DPPROW) OP8(T_DPPROW);
            if (K == K_ADDDPP) OP8(T_ADDDPP);
            if (K == K_CMP) OP8(T_CMP);
            if (K == K_MULLO) OP8(T_MULLO);
            if (K == K_MUL24) OP8(T_MUL24);
            if (K == K_PERM) OP8(T_PERM);
            if (K == K_SUBREV) OP8(T_SUBREV);
            if (K == K_MINU) OP8(T_MINU);
            if (K == K_MED3) OP8(T_MED3);
            if (K == K_SAD) OP8(T_SAD);
            if (K == K_PKADD) OP8(T_PKADD);
            if (K == K_BPERM) {
                OP8(T_BPERM);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (K == K_LSHR64) OP8_64(T_LSHR64);
            if (K == K_LSHL64) OP8_64(T_LSHL64);
        }
    }
    const unsigned long long t1 = clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc ^= r[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int K>
static void run(int ncu, uint32_t *d_out, unsigned long long *d_clk)
{
    for (int wps = 1; wps <= 8; wps++) {
        if (wps == 7) continue;
        const int grid = ncu * 4 * wps;  // one 64-thread block = one wave; 4 SIMDs per CU
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_probe<K>, dim3(grid), dim3(64), 0, 0, d_out, 1u, d_clk);  // warm-up
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<K>, dim3(grid), dim3(64), 0, 0, d_out, 7u, d_clk);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk = 0;
        CHK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
        const double n_inst = (double)ITERS * UNROLL * CHAINS;  // per wave
        const double per_simd = n_inst * wps / (ms * 1e-3);     // wave-instructions / s / SIMD
        // clock64() on gfx950 = s_memtime = shader cycles of wave 0 of block 0 over its whole loop
        const double cyc_per_inst_wave0 = (double)clk / n_inst;
        printf("%-22s waves/SIMD %d  %8.3f ms  %7.3f G wave-instr/s/SIMD  chip %8.1f G/s  cycles/instr @2.4GHz %.2f"
               "  (wave0: %.2f shader cycles per own instr => %.2f per SIMD slot)\n",
               names[K], wps, ms, per_simd * 1e-9, per_simd * 1e-9 * ncu * 4, 2.4e9 / per_simd, cyc_per_inst_wave0,
               cyc_per_inst_wave0 / wps);
    }
}

template <int K>
static void run_all(int ncu, uint32_t *d_out, unsigned long long *d_clk)
{
    run<K>(ncu, d_out, d_clk);
    if constexpr (K + 1 < K_N) run_all<K + 1>(ncu, d_out, d_clk);
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, ncu, p.clockRate);
    uint32_t *d_out;
    unsigned long long *d_clk;
    CHK(hipMalloc(&d_out, (size_t)ncu * 4 * 8 * 64 * 4));
    CHK(hipMalloc(&d_clk, 8));
    run_all<0>(ncu, d_out, d_clk);
    return 0;
}
