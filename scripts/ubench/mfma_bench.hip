// Calibration for the MFMA-busy counter: a kernel that does nothing but v_mfma_f32_32x32x2_f32 on every
// SIMD of the chip (2 wavefronts per SIMD, two independent accumulators each).  Its
// SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE ratio is "100 % of the fp32 matrix pipe"; the achieved
// TFLOP/s is what the pipe really delivers at the clock the chip holds under this load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) mfma_only(float* out, int iters, float a, float b)
{
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 1.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_only, dim3(256), dim3(512), 0, 0, d, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 16 * 4096.0;
        printf("mfma_only: %.3f ms, %.1f TFLOP/s fp32 (v_mfma_f32_32x32x2_f32), %.0f MFMAs per wavefront\n", ms, flop / ms / 1e9, iters * 16.0);
    }
    return 0;
}
