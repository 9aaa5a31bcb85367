// The two small per-image stages of the TPS++ regressor that work on the 2x16 bottleneck map
// (64 channels x 32 points = 8 KB per image); plain VALU kernels, one workgroup per image:
//
//   cbam_kernel        CBAM(64, ratio 16): channel attention (shared 1x1 MLP 64->4->64 on the avg- and
//                      max-pooled vectors, sigmoid), then spatial attention (3x3 conv on [mean_c, max_c],
//                      sigmoid)                                  -- tps_pp.py:27-82, used at :163
//   tpe_points_kernel  control points: fc1 = Linear 64->256, ReLU, Linear 256->2, ReLU per point,
//                      fc2 = Linear 64->64 on the flattened (32x2) vector  -- tps_pp.py:270-285,321-323
//                      and the point side of the score: p = Linear 64->32 -> Linear 32->128 per point
//                                                                 -- tps_pp.py:253-256,305
// Everything is fp32 with fp32 accumulation; these feed the control points, which the TPS solve
// amplifies by up to ~223x, so no reduced precision anywhere.
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
constexpr int CH = 64, NPT = 32, PH = 2, PW = 16;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one wavefront per image; lane = channel, 32 registers = the pixels of that channel
__global__ void __launch_bounds__(64)
cbam_kernel(const float* __restrict__ x, const float* __restrict__ w0,   // (4, 64)
            const float* __restrict__ w2,                                // (64, 4)
            const float* __restrict__ wsp, const float* __restrict__ bsp, // (1, 2, 3, 3), (1)
            float* __restrict__ out)
{
    __shared__ float sMap[2][PH][PW];
    __shared__ float sSa[NPT];
    const int c = threadIdx.x;
    const float* xp = x + ((size_t)blockIdx.x * CH + c) * NPT;
    float v[NPT];
#pragma unroll
    for (int p = 0; p < NPT; ++p) v[p] = xp[p];
    float s = 0.0f, mx = v[0];
#pragma unroll
    for (int p = 0; p < NPT; ++p) { s += v[p]; mx = fmaxf(mx, v[p]); }
    const float avg = s * (1.0f / NPT);
    // shared MLP on both pooled vectors
    float oa = 0.0f, om = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ha = fmaxf(wave_sum(w0[j * CH + c] * avg), 0.0f);
        const float hm = fmaxf(wave_sum(w0[j * CH + c] * mx), 0.0f);
        oa = fmaf(w2[c * 4 + j], ha, oa);
        om = fmaf(w2[c * 4 + j], hm, om);
    }
    const float ca = sigmoidf(oa + om);
#pragma unroll
    for (int p = 0; p < NPT; ++p) v[p] = ca * v[p];
    // spatial attention: channel mean / max per pixel
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
        const float m = wave_sum(v[p]) * (1.0f / CH);
        const float xm = wave_max(v[p]);
        if (c == 0) { sMap[0][p / PW][p % PW] = m; sMap[1][p / PW][p % PW] = xm; }
    }
    __syncthreads();
    if (c < NPT) {
        const int y = c / PW, xx = c % PW;
        float a = bsp[0];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = y + ky - 1, ix = xx + kx - 1;
                    if (iy >= 0 && iy < PH && ix >= 0 && ix < PW)
                        a = fmaf(wsp[(ch * 3 + ky) * 3 + kx], sMap[ch][iy][ix], a);
                }
        sSa[c] = sigmoidf(a);
    }
    __syncthreads();
    float* op = out + ((size_t)blockIdx.x * CH + c) * NPT;
#pragma unroll
    for (int p = 0; p < NPT; ++p) op[p] = sSa[p] * v[p];
}

struct PointsParams {
    const float* en;                       // (N, 64, 32): bottleneck map before CBAM, en[b][c][pt]
    const float* fc1a_w; const float* fc1a_b;   // (256, 64), (256)
    const float* fc1b_w; const float* fc1b_b;   // (2, 256), (2)
    const float* fc2_w; const float* fc2_b;     // (64, 64), (64)
    const float* pl0_w; const float* pl0_b;     // (32, 64), (32)
    const float* pl1_w; const float* pl1_b;     // (128, 32), (128)
    float* ctrl;                           // (N, 32, 2)
    float* p;                              // (N, 32, 128)
};

__global__ void __launch_bounds__(256)
tpe_points_kernel(const PointsParams P)
{
    __shared__ float sEn[NPT][CH + 1];         // [pt][c]
    __shared__ float sPart[4][NPT][2];         // per-wavefront partial sums of the 256 -> 2 layer
    __shared__ float sV[NPT * 2];              // fc1 output, flattened (pt, 2)
    __shared__ float sT1[NPT][32 + 1];         // p_linear hidden
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    for (int e = tid; e < CH * NPT; e += 256) {
        const int c = e / NPT, pt = e - c * NPT;
        sEn[pt][c] = P.en[(size_t)b * CH * NPT + e];
    }
    __syncthreads();
    // ---- fc1: hidden unit h = tid; z[pt] = relu(W1a[h] . en[pt] + b); then 256 -> 2 reduction ----
    {
        float w[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) w[c] = P.fc1a_w[tid * CH + c];
        const float bh = P.fc1a_b[tid];
        const float wb0 = P.fc1b_w[tid], wb1 = P.fc1b_w[256 + tid];
        for (int pt = 0; pt < NPT; ++pt) {
            float z = bh;
#pragma unroll
            for (int c = 0; c < CH; ++c) z = fmaf(w[c], sEn[pt][c], z);
            z = fmaxf(z, 0.0f);
            const float s0 = wave_sum(z * wb0), s1 = wave_sum(z * wb1);
            if (lane == 0) { sPart[wv][pt][0] = s0; sPart[wv][pt][1] = s1; }
        }
    }
    // ---- p_linear stage 1: t1[pt][j], 1024 outputs, 4 per thread ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256, pt = e >> 5, j = e & 31;
        float a = P.pl0_b[j];
        for (int c = 0; c < CH; ++c) a = fmaf(P.pl0_w[j * CH + c], sEn[pt][c], a);
        sT1[pt][j] = a;
    }
    __syncthreads();
    if (tid < NPT * 2) {
        const int pt = tid >> 1, o = tid & 1;
        const float s = ((sPart[0][pt][o] + sPart[1][pt][o]) + sPart[2][pt][o]) + sPart[3][pt][o] + P.fc1b_b[o];
        sV[tid] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    // ---- fc2: ctrl[o] = W2[o] . v + b ----
    if (tid < NPT * 2) {
        float a = P.fc2_b[tid];
        for (int i = 0; i < NPT * 2; ++i) a = fmaf(P.fc2_w[tid * (NPT * 2) + i], sV[i], a);
        P.ctrl[(size_t)b * NPT * 2 + tid] = a;
    }
    // ---- p_linear stage 2: p[pt][o], 4096 outputs, 16 per thread ----
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256, pt = e >> 7, o = e & 127;
        float a = P.pl1_b[o];
#pragma unroll
        for (int j = 0; j < 32; ++j) a = fmaf(P.pl1_w[o * 32 + j], sT1[pt][j], a);
        P.p[(size_t)b * NPT * 128 + e] = a;
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_cbam_fwd(const float* x, const float* mlp0_w, const float* mlp2_w,
                                const float* sp_w, const float* sp_b, float* out, int N,
                                tpspp_stream_t stream)
{
    TPSPP_REQUIRE(x && mlp0_w && mlp2_w && sp_w && sp_b && out, "tpspp_cbam_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0, "tpspp_cbam_fwd: bad batch");
    if (N == 0) return TPSPP_OK;
    hipLaunchKernelGGL(cbam_kernel, dim3((unsigned)N), dim3(64), 0, tpspp::as_stream(stream), x, mlp0_w, mlp2_w,
                       sp_w, sp_b, out);
    return tpspp::check_launch("tpspp_cbam_fwd");
}

TPSPP_EXPORT int tpspp_tpe_points_fwd(const float* en_feat, const float* fc1a_w, const float* fc1a_b,
                                      const float* fc1b_w, const float* fc1b_b, const float* fc2_w,
                                      const float* fc2_b, const float* pl0_w, const float* pl0_b,
                                      const float* pl1_w, const float* pl1_b, float* ctrl, float* p, int N,
                                      tpspp_stream_t stream)
{
    TPSPP_REQUIRE(en_feat && fc1a_w && fc1a_b && fc1b_w && fc1b_b && fc2_w && fc2_b && pl0_w && pl0_b &&
                  pl1_w && pl1_b && ctrl && p, "tpspp_tpe_points_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0, "tpspp_tpe_points_fwd: bad batch");
    if (N == 0) return TPSPP_OK;
    PointsParams P;
    P.en = en_feat; P.fc1a_w = fc1a_w; P.fc1a_b = fc1a_b; P.fc1b_w = fc1b_w; P.fc1b_b = fc1b_b;
    P.fc2_w = fc2_w; P.fc2_b = fc2_b; P.pl0_w = pl0_w; P.pl0_b = pl0_b; P.pl1_w = pl1_w; P.pl1_b = pl1_b;
    P.ctrl = ctrl; P.p = p;
    hipLaunchKernelGGL(tpe_points_kernel, dim3((unsigned)N), dim3(256), 0, tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_tpe_points_fwd");
}
