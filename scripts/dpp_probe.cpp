// Probe of the DPP controls the wave kernel relies on (gfx950): prints source lane per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void probe(int *out, uint64_t *o64, const uint8_t *buf)
{
    const int lane = threadIdx.x;
    out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x134, 0xF, 0xF, false);  // wave_rol:1
    int v = (lane * 37) % 101;
    int m = v;
    m = max(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x141, 0xF, 0xF, false));  // row_half_mirror
    m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x140, 0xF, 0xF, false));  // row_mirror
    int r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
    int r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
    out[2 * 64 + lane] = max(max(r0, r1), max(r2, r3));
    out[3 * 64 + lane] = v;
    uint64_t x;
    __builtin_memcpy(&x, buf + lane, 8);  // unaligned 8-byte load
    o64[lane] = x;
}
int main()
{
    int *d; uint64_t *d64; uint8_t *db;
    hipMalloc(&d, 4 * 64 * 4); hipMalloc(&d64, 64 * 8); hipMalloc(&db, 256);
    uint8_t hb[256]; for (int i = 0; i < 256; i++) hb[i] = (uint8_t)i;
    hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, d64, db);
    int h[4 * 64]; uint64_t h64[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(h64, d64, sizeof(h64), hipMemcpyDeviceToHost);
    printf("ror1:"); for (int i = 0; i < 64; i++) printf(" %d", h[i]); printf("\n");
    printf("rol1:"); for (int i = 0; i < 64; i++) printf(" %d", h[64 + i]); printf("\n");
    int mx = 0; for (int i = 0; i < 64; i++) mx = h[192 + i] > mx ? h[192 + i] : mx;
    int ok = 1; for (int i = 0; i < 64; i++) ok &= h[128 + i] == mx;
    printf("max reduce ok=%d (max %d)\n", ok, mx);
    int ok2 = 1; for (int i = 0; i < 64; i++) { uint64_t e; memcpy(&e, hb + i, 8); ok2 &= e == h64[i]; }
    printf("unaligned u64 load ok=%d\n", ok2);
    return 0;
}
