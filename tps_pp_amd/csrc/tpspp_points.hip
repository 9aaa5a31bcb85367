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

// Persistent workgroups (one per CU), 2 x 256 threads: each half takes an image per trip (32 points x 8 groups), both share
// the weights in LDS and give every SIMD a second wavefront to issue from while one waits on its accumulation chain.
// The first version gave a thread one hidden unit of fc1 and let it walk the 32 points, reading the bottleneck map as
// 64 single-word LDS broadcasts per point and reducing the 256 -> 2 layer with two wavefront-wide shuffles per point:
// ~2400 LDS-pipe instructions per thread and image, 87 us per 512 images for 0.74 GFLOP.  Here the thread owns a POINT
// (its 64 channels in registers) and a group of 32 hidden units whose weight rows it reads as 16-byte LDS broadcasts
// (the 64 KB of fc1 weights stay in LDS for the life of the workgroup); the 256 -> 2 layer is a per-thread sum over its
// 32 units plus one 8-way LDS reduction; p_linear's second layer keeps a thread's weight row in registers.  Every dot
// product keeps its accumulation order (bias first, inputs ascending).
constexpr int EP = CH + 4;                     // pitch of a point's channel row in LDS: 16-byte rows, banks shifted by 4
struct PointsLds {
    float w1a[256][CH];                        // fc1 first layer (hidden, channel)
    float b1a[256], wb0[256], wb1[256];        // its bias; the two rows of the 256 -> 2 layer
    float pl0[32][CH];                         // p_linear first layer
    struct Img {
        float en[NPT][EP];                     // the image's bottleneck map [point][channel]
        float part[8][NPT][2];
        float v[NPT * 2];
        float t1[NPT][32];
    } img[2];
};

__global__ void __launch_bounds__(512)
tpe_points_kernel(const PointsParams P, int N)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PointsLds& S = *reinterpret_cast<PointsLds*>(smem_raw);
    for (int e = threadIdx.x; e < 256 * CH / 4; e += 512)
        reinterpret_cast<float4*>(&S.w1a[0][0])[e] = reinterpret_cast<const float4*>(P.fc1a_w)[e];
    for (int e = threadIdx.x; e < 32 * CH / 4; e += 512)
        reinterpret_cast<float4*>(&S.pl0[0][0])[e] = reinterpret_cast<const float4*>(P.pl0_w)[e];
    const int tid = threadIdx.x & 255, sub = threadIdx.x >> 8;
    if (sub == 0) { S.b1a[tid] = P.fc1a_b[tid]; S.wb0[tid] = P.fc1b_w[tid]; S.wb1[tid] = P.fc1b_w[256 + tid]; }
    PointsLds::Img& I = S.img[sub];
    const int pt = tid & 31, grp = tid >> 5;
    // p_linear second layer: this thread's output unit and its weight row (kept for every image)
    const int o2 = tid & 127;
    float w2[32];
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const float4 q = *reinterpret_cast<const float4*>(P.pl1_w + o2 * 32 + j);
        w2[j] = q.x; w2[j + 1] = q.y; w2[j + 2] = q.z; w2[j + 3] = q.w;
    }
    const float b2 = P.pl1_b[o2];

    for (int b0 = 2 * blockIdx.x; b0 < N; b0 += 2 * gridDim.x) {
        const bool live = b0 + sub < N;        // an odd batch leaves the second half idle on the last trip (it still syncs)
        const int b = live ? b0 + sub : b0;
        __syncthreads();                       // weights in place (first trip); previous image's readers are done
        for (int e = tid; e < CH * NPT; e += 256) {
            const int c = e / NPT, q = e - c * NPT;
            I.en[q][c] = P.en[(size_t)b * CH * NPT + e];
        }
        __syncthreads();
        float en[CH];
#pragma unroll
        for (int c = 0; c < CH; c += 4) {
            const float4 q = *reinterpret_cast<const float4*>(&I.en[pt][c]);
            en[c] = q.x; en[c + 1] = q.y; en[c + 2] = q.z; en[c + 3] = q.w;
        }
        // ---- fc1: z[h] = relu(W1a[h] . en + b[h]) for the group's 32 hidden units, folded into the 256 -> 2 layer ----
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
        for (int j = 0; j < 32; ++j) {
            const int h = grp * 32 + j;
            float z = S.b1a[h];
#pragma unroll
            for (int c = 0; c < CH; c += 4) {
                const float4 w = *reinterpret_cast<const float4*>(&S.w1a[h][c]);
                z = fmaf(w.x, en[c], z); z = fmaf(w.y, en[c + 1], z); z = fmaf(w.z, en[c + 2], z); z = fmaf(w.w, en[c + 3], z);
            }
            z = fmaxf(z, 0.0f);
            s0 = fmaf(z, S.wb0[h], s0);
            s1 = fmaf(z, S.wb1[h], s1);
        }
        I.part[grp][pt][0] = s0; I.part[grp][pt][1] = s1;
        // ---- p_linear stage 1: t1[pt][j] for j = 4 grp .. 4 grp + 3 ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = grp * 4 + i;
            float a = P.pl0_b[j];
#pragma unroll
            for (int c = 0; c < CH; c += 4) {
                const float4 w = *reinterpret_cast<const float4*>(&S.pl0[j][c]);
                a = fmaf(w.x, en[c], a); a = fmaf(w.y, en[c + 1], a); a = fmaf(w.z, en[c + 2], a); a = fmaf(w.w, en[c + 3], a);
            }
            I.t1[pt][j] = a;
        }
        __syncthreads();
        if (tid < NPT * 2) {
            const int q = tid >> 1, o = tid & 1;
            float s = I.part[0][q][o];
#pragma unroll
            for (int g = 1; g < 8; ++g) s += I.part[g][q][o];
            I.v[tid] = fmaxf(s + P.fc1b_b[o], 0.0f);
        }
        __syncthreads();
        // ---- fc2: ctrl[o] = W2[o] . v + b ----
        if (tid < NPT * 2) {
            float a = P.fc2_b[tid];
            for (int i = 0; i < NPT * 2; ++i) a = fmaf(P.fc2_w[tid * (NPT * 2) + i], I.v[i], a);
            if (live) P.ctrl[(size_t)b * NPT * 2 + tid] = a;
        }
        // ---- p_linear stage 2: p[q][o2] for the points q = (tid >> 7) + 2 i ----
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int q = (tid >> 7) + 2 * i;
            float a = b2;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                const float4 t = *reinterpret_cast<const float4*>(&I.t1[q][j]);
                a = fmaf(w2[j], t.x, a); a = fmaf(w2[j + 1], t.y, a); a = fmaf(w2[j + 2], t.z, a); a = fmaf(w2[j + 3], t.w, a);
            }
            if (live) P.p[(size_t)b * NPT * 128 + q * 128 + o2] = a;
        }
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
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tpe_points_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const int pairs = (N + 1) / 2;
    const int grid = pairs < 256 ? pairs : 256;                      // persistent: one workgroup per CU
    hipLaunchKernelGGL(tpe_points_kernel, dim3((unsigned)grid), dim3(512), sizeof(PointsLds), tpspp::as_stream(stream), P, N);
    return tpspp::check_launch("tpspp_tpe_points_fwd");
}
