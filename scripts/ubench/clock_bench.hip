// What does one s_memtime tick mean, and what clock do short kernels actually run at?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters, float seed)
{
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = wall_clock64();
    float a = seed;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0000001f, 0.5f);   // dependent chain
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x*4+2] = (long long)a; }
}
int main()
{
    long long* d; hipMalloc(&d, 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall_clock rate %d kHz, device clock attr %d kHz\n", rate, clk);
    for (int iters : {2000, 20000, 200000, 2000000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("iters %8d: event %9.2f us | s_memtime ticks %10lld (%.1f MHz vs event) | wall_clock ticks %8lld (=%.2f us) | ticks/iter %.2f\n",
                   iters, ms * 1e3, h[0], h[0] / (ms * 1e3), h[1], h[1] * 1e3 / rate, (double)h[0] / iters);
        }
    }
    return 0;
}
