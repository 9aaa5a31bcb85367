// What does a chain of DEPENDENT small launches cost on this chip?  Models the decoder's step projections (512 tokens x 512
// outputs, K = 512, three-term form: 256 workgroups of 256 threads, each reading a 64 KB slice of X shared with 15 others
// and a 64 KB slice of W shared with 15 others, writing 4 KB of the next launch's X):
//   empty      nothing but the launch
//   store      every workgroup writes its 4 KB
//   load       + reads its two 64 KB slices (16-byte loads, all in flight together), one dependent pass
//   load+mfma  + 24 v_mfma_f32_32x32x16_bf16 per wavefront and the 4-way reduction through LDS
// Each variant: 2000 launches back to back on one stream, output of launch i is the X of launch i + 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MAP: 0 launch order; 1 each XCD (workgroup id % 8) an 8 x 4 sub-grid of (token blocks, output tiles); 2 a 4 x 8 sub-grid;
// 3 a 16 x 2 strip (all of X, an eighth of W)
template <int MODE, int MAP = 0>
__global__ void __launch_bounds__(256) k(const float* __restrict__ X, const u32x4* __restrict__ W, float* __restrict__ out)
{
    __shared__ float sRed[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int tb = blockIdx.x, ct = blockIdx.y;
    if (MAP) {
        const int lin = blockIdx.x + 16 * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
        if (MAP == 1) { tb = (xcd & 1) * 8 + (slot & 7); ct = (xcd >> 1) * 4 + (slot >> 3); }
        if (MAP == 2) { tb = (xcd & 3) * 4 + (slot & 3); ct = (xcd >> 2) * 8 + (slot >> 2); }
        if (MAP == 3) { tb = slot & 15; ct = xcd * 2 + (slot >> 4); }
    }
    if (MODE == 0) { if (tid == 9999) out[0] = 1.0f; return; }
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    if (MODE >= 2) {
        const float4* xp = reinterpret_cast<const float4*>(X + (size_t)(tb * 32 + l31) * 512 + 16 * (wv * 8) + 8 * half);
        const u32x4* wp = W + ((size_t)(ct * 32 + wv * 8) * 4 + half) * 32 + l31;
        float4 xa[8][2];
        u32x4 ah[8], al[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xa[j][0] = xp[4 * j]; xa[j][1] = xp[4 * j + 1]; ah[j] = wp[j * 128]; al[j] = wp[j * 128 + 64]; }
        if (MODE == 2) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xa[j][0].x + xa[j][1].w + __builtin_bit_cast(float, ah[j][0]) + __builtin_bit_cast(float, al[j][3]);
            v[0] = s;
        } else {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x4 b;
                b[0] = __builtin_bit_cast(unsigned, xa[j][0].x); b[1] = __builtin_bit_cast(unsigned, xa[j][0].z);
                b[2] = __builtin_bit_cast(unsigned, xa[j][1].x); b[3] = __builtin_bit_cast(unsigned, xa[j][1].z);
                const bf16x8 A = __builtin_bit_cast(bf16x8, ah[j]), A2 = __builtin_bit_cast(bf16x8, al[j]), B = __builtin_bit_cast(bf16x8, b);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[r];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = (sRed[0][4 * wv + e][lane] + sRed[1][4 * wv + e][lane]) + (sRed[2][4 * wv + e][lane] + sRed[3][4 * wv + e][lane]);
        }
    }
    // 4 outputs per lane: token tb * 32 + l31, outputs ct * 32 + 8 wv + 4 half ..
    *reinterpret_cast<float4*>(out + (size_t)(tb * 32 + l31) * 512 + ct * 32 + 8 * wv + 4 * half) =
        make_float4(v[0] * 1e-9f, v[1] * 1e-9f, v[2] * 1e-9f, v[3] * 1e-9f);
}

template <int MODE, int MAP = 0> void run(const char* name, float* a, float* b, u32x4* w)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k<MODE, MAP>), dim3(16, 16), dim3(256), 0, 0, (i & 1) ? b : a, w, (i & 1) ? a : b);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<MODE, MAP>), dim3(16, 16), dim3(256), 0, 0, (i & 1) ? b : a, w + (size_t)(i % 36) * 65536, (i & 1) ? a : b);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %6.2f us per launch\n", name, ms * 1e3f / n);
    }
}

// the same chain captured once into a hipGraph and replayed
template <int MODE> void run_graph(const char* name, float* a, float* b, u32x4* w)
{
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    const int n = 2000;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(16, 16), dim3(256), 0, st, (i & 1) ? b : a, w + (size_t)(i % 36) * 65536, (i & 1) ? a : b);
    hipStreamEndCapture(st, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("%s: graph instantiate failed\n", name); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %6.2f us per launch (hipGraph of %d kernel nodes)\n", name, ms * 1e3f / n, n);
    }
}

int main()
{
    float *a, *b; u32x4* w;
    hipMalloc(&a, 512 * 512 * 4); hipMalloc(&b, 512 * 512 * 4);
    hipMalloc(&w, (size_t)36 * 65536 * 16);                 // 36 weights of 1 MB: a layer-step's worth cycling, as in the decoder
    hipMemset(a, 0, 512 * 512 * 4); hipMemset(b, 0, 512 * 512 * 4); hipMemset(w, 0, (size_t)36 * 65536 * 16);
    run<0>("empty", a, b, w);
    run<1>("store", a, b, w);
    run<2>("load", a, b, w);
    run<3>("load+mfma", a, b, w);
    run<2, 1>("load 8x4", a, b, w);
    run<2, 2>("load 4x8", a, b, w);
    run<2, 3>("load 16x2", a, b, w);
    run_graph<1>("store", a, b, w);
    run_graph<3>("load+mfma", a, b, w);
    return 0;
}
