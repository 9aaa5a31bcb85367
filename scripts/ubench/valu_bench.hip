// VALU throughput on gfx950: v_fma_f32 vs v_pk_fma_f32 (is packed fp32 2 FMAs per issue slot?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s)
{
    float a[16]; f2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2*i], a[2*i+1]};
    f2 m = {s, s * 1.0001f}; f2 c = {0.25f, 0.125f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(c.x));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m), "v"(c));
        }
    }
    float r = 0;
    if (MODE == 0) { for (int i = 0; i < 16; ++i) r += a[i]; } else { for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main()
{
    float* o; hipMalloc(&o, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 100000;
    for (int wpc : {1, 2, 4, 8}) {
        int blocks = 256 * wpc;
        float ms[2];
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.000001f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.000001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[mode], e0, e1);
            }
        }
        double fma = (double)blocks * 256 * 16 * iters;
        printf("waves/SIMD %d: v_fma_f32 %.3f ms = %.1f TFLOP/s | v_pk_fma_f32 %.3f ms = %.1f TFLOP/s\n", wpc,
               ms[0], 2 * fma / ms[0] / 1e9, ms[1], 2 * fma / ms[1] / 1e9);
    }
    return 0;
}
