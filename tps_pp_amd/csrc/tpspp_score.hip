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

// ---- the same chain with the three-term bf16 split ("bf16x3", tpspp_conv_bf16.hip): 60 bf16 MFMAs per 32 pixels
// instead of 160 fp32 ones; every operand hi = bf16(v), lo = bf16(v - hi), every product hi*hi + hi*lo + lo*hi in fp32.
// Result registers chain into the next layer's operand as in tpspp_front_bf16.hip (k-slots in the order
// [0,1,2,3,8,9,10,11,4,5,6,7,12,..,15] inside every 16 features: the weight slabs are permuted on the host, the
// per-image point matrix while it is staged).  ~5e-6 of scale before the tanh.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ScoreXParams {
    const float* de;       // (N, 64, n)
    const u32x4* w1;       // [hi|lo][4 k-steps][2][32 out][8]   natural k order
    const float* b1;
    const u32x4* w2;       // [hi|lo][2 k-steps][2][128 out][8]  chain k order
    const float* b2;
    const float* p;        // (N, 32 points, 128)
    float* score_t;        // (N, 32, n)
    int n;
    float scale;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(v0, v1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(v0 - h0, v1 - h1);
}

__device__ __forceinline__ f32x16 mfma3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x16 acc)
{
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, al);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bh), b1 = __builtin_bit_cast(bf16x8, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
    return acc;
}

// 16 result registers of one 32-row tile (+ bias) -> the hi / lo operands of its two k-steps (chain order)
__device__ __forceinline__ void tile_to_operands(const f32x16& t, const float* __restrict__ bias, int half,
                                                 u32x4 (&oh)[2], u32x4 (&ol)[2])
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t[4 * g + e] + bias[8 * g + 4 * half + e];
        unsigned h, l;
        split2(v[0], v[1], h, l);
        oh[g >> 1][2 * (g & 1)] = h; ol[g >> 1][2 * (g & 1)] = l;
        split2(v[2], v[3], h, l);
        oh[g >> 1][2 * (g & 1) + 1] = h; ol[g >> 1][2 * (g & 1) + 1] = l;
    }
}

__global__ void __launch_bounds__(256, 2)
score_x3_kernel(const ScoreXParams P)
{
    constexpr int W1U = 4 * 2 * 32, W2U = 2 * 2 * 128, PU = 8 * 2 * 32;     // 16-B units of one (hi or lo) slab
    __shared__ u32x4 sW1[2 * W1U];        // 8 KB
    __shared__ u32x4 sW2[2 * W2U];        // 16 KB
    __shared__ u32x4 sP[2 * PU];          // 16 KB: [hi|lo][8 k-steps][2][32 points][8]
    __shared__ float sB[M1 + M2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < 2 * W1U; i += 256) sW1[i] = P.w1[i];
    for (int i = tid; i < 2 * W2U; i += 256) sW2[i] = P.w2[i];
    if (tid < M1) sB[tid] = P.b1[tid];
    if (tid < M2) sB[M1 + tid] = P.b2[tid];
    // the point matrix of this image, split and laid out as the A operand of stage 3: unit (j, h, pt) holds
    // features 16 j + perm[8 h + e], e = 0..7, of point pt
    for (int u = tid; u < PU; u += 256) {
        const int pt = u & 31, h = (u >> 5) & 1, j = u >> 6;
        const float* pp = P.p + ((size_t)b * PT + pt) * M2 + 16 * j + 4 * h;     // e < 4: 4h + e; e >= 4: 8 + 4h + (e-4)
        const float4 v0 = *reinterpret_cast<const float4*>(pp), v1 = *reinterpret_cast<const float4*>(pp + 8);
        unsigned hh[4], ll[4];
        split2(v0.x, v0.y, hh[0], ll[0]); split2(v0.z, v0.w, hh[1], ll[1]);
        split2(v1.x, v1.y, hh[2], ll[2]); split2(v1.z, v1.w, hh[3], ll[3]);
        u32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = hh[e]; c[e] = ll[e]; }
        sP[u] = a; sP[PU + u] = c;
    }
    __syncthreads();

    const int lane = tid & (kWave - 1), wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int px = blockIdx.x * 128 + wv * 32 + l31;
    const int pxc = px < P.n ? px : P.n - 1;
    const float* xb = P.de + (size_t)b * C * P.n + pxc;

    // ---- stage 1: t (32) = W1 x + b1; k-step j, slot 8h + e = channel 16 j + 8 h + e ----
    float xin[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) xin[j][e] = xb[(size_t)(16 * j + 8 * half + e) * P.n];
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32x4 bh, bl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2(xin[j][2 * q], xin[j][2 * q + 1], h, l);
            bh[q] = h; bl[q] = l;
        }
        t = mfma3(sW1[(2 * j + half) * 32 + l31], sW1[W1U + (2 * j + half) * 32 + l31], bh, bl, t);
    }
    u32x4 th[2], tl[2];
    tile_to_operands(t, sB, half, th, tl);

    // ---- stage 2: f (128) = W2 t + b2: four 32-feature tiles, each straight into stage 3 ----
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        f32x16 f;
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            f = mfma3(sW2[(2 * j + half) * 128 + 32 * tt + l31], sW2[W2U + (2 * j + half) * 128 + 32 * tt + l31],
                      th[j], tl[j], f);
        u32x4 fh[2], fl[2];
        tile_to_operands(f, sB + M1 + 32 * tt, half, fh, fl);
        // ---- stage 3: score (32 points) += P[:, features of this tile] f ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int J = 2 * tt + j;
            sacc = mfma3(sP[(2 * J + half) * 32 + l31], sP[PU + (2 * J + half) * 32 + l31], fh[j], fl[j], sacc);
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

TPSPP_EXPORT int tpspp_score_x3_fwd(const float* de_feat, const void* w1_slab, const float* b1,
                                    const void* w2_slab, const float* b2, const float* p, float scale,
                                    float* score_t, int N, int n, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(de_feat && w1_slab && b1 && w2_slab && b2 && p && score_t, "tpspp_score_x3_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && n > 0 && N <= 65535, "tpspp_score_x3_fwd: bad sizes");
    if (N == 0) return TPSPP_OK;
    ScoreXParams P;
    P.de = de_feat; P.w1 = static_cast<const u32x4*>(w1_slab); P.b1 = b1; P.w2 = static_cast<const u32x4*>(w2_slab);
    P.b2 = b2; P.p = p; P.score_t = score_t; P.n = n; P.scale = scale;
    hipLaunchKernelGGL(score_x3_kernel, dim3((unsigned)((n + 127) / 128), (unsigned)N), dim3(256), 0,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_score_x3_fwd");
}
