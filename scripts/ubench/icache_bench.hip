// Does a long straight-line kernel pay for instruction fetch on every launch?
// One pass over NI unrolled FMAs is timed twice inside the same kernel (second pass = warm I-cache),
// on back-to-back launches of the same kernel, for 1 and 16 wavefronts per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NI>
__device__ __forceinline__ void body(float& a, float& b, float& c, float& d, float m, float k)
{
#pragma unroll
    for (int i = 0; i < NI / 4; ++i) {
        a = __builtin_fmaf(a, m, k); b = __builtin_fmaf(b, m, k);
        c = __builtin_fmaf(c, m, k); d = __builtin_fmaf(d, m, k);
    }
}
template <int NI>
__global__ void __launch_bounds__(1024) straight(long long* out, float m, float k, int passes)
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    long long t[5];
    t[0] = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {          // not unrolled: the same code bytes run again
        body<NI>(a, b, c, d, m, k);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        t[p + 1] = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) {
        for (int p = 0; p < passes; ++p) out[blockIdx.x * 4 + p] = t[p + 1] - t[p];
        out[blockIdx.x * 4 + 3] = (long long)(a + b + c + d);
    }
}
template <int NI>
void run(long long* d, int threads)
{
    long long h[256 * 4];
    for (int rep = 0; rep < 4; ++rep)
        hipLaunchKernelGGL(straight<NI>, dim3(256), dim3(threads), 0, 0, d, 1.0000001f, 0.5f, 3);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double p0 = 0, p1 = 0, p2 = 0;
    for (int g = 0; g < 256; ++g) { p0 += h[g * 4]; p1 += h[g * 4 + 1]; p2 += h[g * 4 + 2]; }
    printf("NI %5d (%5.1f KB code) threads %4d: pass0 %8.0f ticks  pass1 %8.0f  pass2 %8.0f  (cold - warm = %.2f us @2.4GHz)\n",
           NI, NI * 4 / 1024.0, threads, p0 / 256, p1 / 256, p2 / 256, (p0 - p1) / 256 / 2400.0);
}
int main()
{
    long long* d; hipMalloc(&d, 256 * 4 * 8);
    for (int threads : {64, 1024}) {
        run<256>(d, threads); run<1024>(d, threads); run<2048>(d, threads); run<4096>(d, threads);
    }
    return 0;
}
