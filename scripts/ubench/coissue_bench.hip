// Do a matrix-instruction wavefront and a vector-ALU wavefront on the same SIMD overlap?  512-thread workgroups (8
// wavefronts = 2 per SIMD), one per CU: wavefronts 0-3 run a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 (CHAINS
// independent accumulators), wavefronts 4-7 a stream of independent v_fma_f32 / ds_read traffic.  Times: each role
// alone, both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ void __launch_bounds__(512) k(float* out, int iters, int do_mfma, int do_valu, float x)
{
    const int wv = threadIdx.x >> 6;
    if (wv < 4) {
        if (!do_mfma) return;
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)x; b[i] = (__bf16)(x + 1.f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 36 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        }
        float t = 0.f;
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) t += acc[c][i];
        if (t == 12345.f) out[threadIdx.x] = t;
    } else {
        if (!do_valu) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 75; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], x, 1.0f);       // 600 independent-ish VALU ops
        }
        float t = 0.f;
        for (int i = 0; i < 8; ++i) t += v[i];
        if (t == 12345.f) out[threadIdx.x] = t;
    }
}

template <int CHAINS>
float run(float* d, int m, int v)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(512), 0, 0, d, 2000, m, v, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
int main()
{
    float* d; hipMalloc(&d, 4096 * 4);
    printf("per iteration: 36 MFMAs per matrix wavefront, 600 v_fma_f32 per vector wavefront; 2000 iterations; clocks at 2.4 GHz\n");
    {
        const float m = run<1>(d, 1, 0), v = run<1>(d, 0, 1), b = run<1>(d, 1, 1);
        printf("1 chain : mfma alone %.3f ms (%.0f clk/iter)  valu alone %.3f ms (%.0f clk/iter)  both %.3f ms (%.0f clk/iter)\n", m, m * 1.2e3, v, v * 1.2e3, b, b * 1.2e3);
    }
    {
        const float m = run<2>(d, 1, 0), v = run<2>(d, 0, 1), b = run<2>(d, 1, 1);
        printf("2 chains: mfma alone %.3f ms (%.0f clk/iter)  valu alone %.3f ms (%.0f clk/iter)  both %.3f ms (%.0f clk/iter)\n", m, m * 1.2e3, v, v * 1.2e3, b, b * 1.2e3);
    }
    {
        const float m = run<4>(d, 1, 0), v = run<4>(d, 0, 1), b = run<4>(d, 1, 1);
        printf("4 chains: mfma alone %.3f ms (%.0f clk/iter)  valu alone %.3f ms (%.0f clk/iter)  both %.3f ms (%.0f clk/iter)\n", m, m * 1.2e3, v, v * 1.2e3, b, b * 1.2e3);
    }
    return 0;
}
