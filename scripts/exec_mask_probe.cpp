// Does a wave64 VALU instruction get cheaper when whole 16-lane quarters are masked off?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_mask_probe scripts/exec_mask_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(int *out, int nactive, int lo, int iters)
{
    const int lane = threadIdx.x;
    int a = lane, b = lane * 3 + 1, c = lane ^ 5, d = lane + 7;
    if (lane >= lo && lane < lo + nactive) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                a = a * 3 + b;  // v_mad / add chains, 4 independent
                b = b ^ (c + u);
                c = c + d;
                d = d ^ a;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a + b + c + d;
}
int main()
{
    int *d;
    hipMalloc(&d, 256 * 64 * 64 * sizeof(int));
    const int cases[][2] = {{64, 0}, {32, 0}, {16, 0}, {16, 8}, {19, 10}, {19, 40}, {8, 0}, {1, 0}};
    for (auto &cs : cases) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256 * 32), dim3(64), 0, 0, d, cs[0], cs[1], 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * 32), dim3(64), 0, 0, d, cs[0], cs[1], 20000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("active lanes %2d starting at %2d: %.3f ms\n", cs[0], cs[1], ms);
    }
    return 0;
}
