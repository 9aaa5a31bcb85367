// Attention score of the TPS++ regressor (fp32 MFMA, register-chained):
//     f = feat_linear.1(feat_linear.0(de_feat as (n pixels, C)))        Linear C->32 -> Linear 32->128
//     score[pt][px] = tanh( (f[px] . p[pt]) * scale )                  p = p_linear(point features)
// One workgroup = one image x 128 pixels; a lane owns one pixel and keeps every intermediate in the
// registers in which v_mfma_f32_32x32x2_f32 delivers it (see tpspp_dgab.hip); only the three weight
// slabs live in LDS (the third one, p, is per image).  The result is written in the (N, F, n) layout
// the warp kernels read coalesced -- the (N, n, F) view the module returns is its transpose.
//
// Reference: Transformation_Parameter_Estimation.get_score / atten_score,
// mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:293-312 (einsum 'bmc,bnc->bmn', * 64^-0.5, tanh).
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
constexpr int C = 64, M1 = 32, M2 = 128, PT = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ScoreParams {
    const float* de;       // (N, 64, n)
    const float* w1_s;     // [64 k-slots (natural channel order)][32 out]
    const float* b1;       // (32)
    const float* w2_s;     // [32 k-slots (MFMA order)][128 out]
    const float* b2;       // (128)
    const float* p;        // (N, 32 points, 128)
    float* score_t;        // (N, 32, n)
    int n;
    float scale;
};

__device__ __forceinline__ constexpr int feat16(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ void __launch_bounds__(256, 2)
score_kernel(const ScoreParams P)
{
    __shared__ __attribute__((aligned(16))) float sW1[C * M1];        // 8 KB
    __shared__ __attribute__((aligned(16))) float sW2[M1 * M2];       // 16 KB
    __shared__ __attribute__((aligned(16))) float sP[M2 * PT];        // 16 KB, slab order
    __shared__ float sB[M1 + M2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < C * M1 / 4; i += 256) reinterpret_cast<float4*>(sW1)[i] = reinterpret_cast<const float4*>(P.w1_s)[i];
    for (int i = tid; i < M1 * M2 / 4; i += 256) reinterpret_cast<float4*>(sW2)[i] = reinterpret_cast<const float4*>(P.w2_s)[i];
    if (tid < M1) sB[tid] = P.b1[tid];
    if (tid < M2) sB[M1 + tid] = P.b2[tid];
    // p slab: slot (2*ks + half), ks = h2*16 + r  <->  feature 32*h2 + feat16(r, half);  sP[slot][pt]
    for (int e = tid; e < M2 * PT; e += 256) {
        const int slot = e / PT, pt = e - slot * PT;
        const int ks = slot >> 1, half = slot & 1;
        const int f = 32 * (ks >> 4) + feat16(ks & 15, half);
        sP[e] = P.p[((size_t)b * PT + pt) * M2 + f];
    }
    __syncthreads();

    const int lane = tid & (kWave - 1), wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int px = blockIdx.x * 128 + wv * 32 + l31;
    const int pxc = px < P.n ? px : P.n - 1;
    const float* xb = P.de + (size_t)b * C * P.n + pxc;

    // ---- stage 1: t (32) = W1 x + b1; k-slot (ks, half) = channel 2*ks + half ----
    float xin[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) xin[ks] = xb[(size_t)(2 * ks + half) * P.n];
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks)
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[(2 * ks + half) * M1 + l31], xin[ks], t, 0, 0, 0);
    float tin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tin[r] = t[r] + sB[feat16(r, half)];

    // ---- stage 2: f (128) = W2 t + b2 ----
    f32x16 f[4];
#pragma unroll
    for (int h2 = 0; h2 < 4; ++h2)
#pragma unroll
        for (int i = 0; i < 16; ++i) f[h2][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2)
            f[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(sW2[(2 * ks + half) * M2 + 32 * h2 + l31], tin[ks],
                                                         f[h2], 0, 0, 0);
    }
    // ---- stage 3: score (32 points) = P f ----
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
    for (int h2 = 0; h2 < 4; ++h2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ks = h2 * 16 + r;
            const float fv = f[h2][r] + sB[M1 + 32 * h2 + feat16(r, half)];
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sP[(2 * ks + half) * PT + l31], fv, sacc, 0, 0, 0);
        }
    }
    if (px < P.n) {
        float* o = P.score_t + (size_t)b * PT * P.n + px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = sacc[r] * P.scale;
            o[(size_t)feat16(r, half) * P.n] = tanhf(v);
        }
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_score_fwd(const float* de_feat, const float* w1_slab, const float* b1,
                                 const float* w2_slab, const float* b2, const float* p, float scale,
                                 float* score_t, int N, int n, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(de_feat && w1_slab && b1 && w2_slab && b2 && p && score_t, "tpspp_score_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && n > 0 && N <= 65535, "tpspp_score_fwd: bad sizes");
    if (N == 0) return TPSPP_OK;
    ScoreParams P;
    P.de = de_feat; P.w1_s = w1_slab; P.b1 = b1; P.w2_s = w2_slab; P.b2 = b2; P.p = p; P.score_t = score_t;
    P.n = n; P.scale = scale;
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)((n + 127) / 128), (unsigned)N), dim3(256), 0,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_score_fwd");
}
