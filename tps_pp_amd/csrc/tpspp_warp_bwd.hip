// Backward of the fused TPS warp: gradients w.r.t. the sampled inputs,
// the control points and the attention score, so the rectifier can sit inside a training graph.
//
// Forward (tpspp_warp.hip):  T = inv_delta_C [C'; 0];  row(p) = [1, P.x, P.y, rbf_k (0.5 s_k + 1)];
//                            grid(p) = row(p) T;  out_i = grid_sample(in_i, grid)  (bilinear, border,
//                            align_corners=True), i = 0 (feature map) and optionally 1 (image).
// Backward, two kernels:
//   A (thread = pixel x 8 channels of one input): dL/d(ix, iy) from the four taps (ATen's CPU formulation:
//     gx += ((ne - nw) s + (se - sw) n) g,  gy += ((sw - nw) e + (se - ne) w) g), times (size-1)/2 and
//     the border-clip derivative (0 where the coordinate was clamped); dL/d in_i scattered with float
//     atomics (taps of neighbouring pixels overlap), the chunk's coordinate gradient added to g_grid;
//   B (workgroup = image):
//   dL/ds[p][k] = 0.5 rbf[p][k] (g_grid(p) . T[3+k]);
//   dL/dT[k] = sum_p row(p)[k] g_grid(p)  (register partials, wavefront shuffles, LDS across wavefronts);
//   dL/dC' = (inv_delta_C^T dL/dT)[:F].
// Replaces: autograd through backbones/tps_pp/tps_pp.py:467-496,597-615 and
//           preprocessor/tps_preprocessor.py:71-83,270-282 (torch.bmm / F.grid_sample backward).
// Bound: L2 atomics + HBM (every g_out element read once, 4 atomics per tap set).
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxK = 64;          // F + 3 <= 64, as in the forward

struct BwdParams {
    const float* g_out[2];
    const float* in[2];
    float* g_in[2];
    int C[2], H[2], W[2];
    int nin;
    const float* grid;             // (N, n, 2) from the forward
    const float* T;                // (N, K, 2)
    const float* inv_delta_c;      // (K, K)
    const float* p_hat; int p_hat_ld;
    const float* p_hat_t;          // (cols, n) or null
    const float* p_xy;             // (n, 2) or null: null = classic table [1, x, y, rbf]
    const float* score;            // (N, n, F) / (N, F, n) or null
    int score_t;
    float* g_ctrl;                 // (N, F, 2)
    float* g_score;                // same layout as score, or null
    int N, F, n;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// ---- kernel A: sampler backward.  Thread = one output pixel x kCPT channels of one input ---------------
// grid = (pixel blocks, channel chunks of input 0 then input 1, images).  dL/d input goes out as float
// atomics (fire-and-forget), the coordinate gradient of the chunk is added to g_grid[b][p] (zeroed by
// the host side), so the parallelism is N * n * C / kCPT threads instead of N * n.  unsafeAtomicAdd = the
// hardware's return-less global_atomic_add_f32 (valid on ordinary device allocations; plain atomicAdd on a
// float compiles to a compare-and-swap loop here, 10x slower).
constexpr int kCPT = 8;

__global__ void __launch_bounds__(256)
warp_bwd_sample_kernel(const BwdParams P, int chunks0, float* __restrict__ g_grid)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n) return;
    const int b = blockIdx.z;
    const int i = (int)blockIdx.y >= chunks0 ? 1 : 0;
    const int c_lo = ((int)blockIdx.y - (i ? chunks0 : 0)) * kCPT;
    const int H = P.H[i], W = P.W[i], C = P.C[i];
    const int c_hi = min(C, c_lo + kCPT);
    const float2 g = reinterpret_cast<const float2*>(P.grid)[(size_t)b * P.n + p];
    float ix = ((g.x + 1.0f) * 0.5f) * (float)(W - 1);
    float iy = ((g.y + 1.0f) * 0.5f) * (float)(H - 1);
    float mx = (float)(W - 1) * 0.5f, my = (float)(H - 1) * 0.5f;
    if (ix <= 0.0f) { ix = 0.0f; mx = 0.0f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); mx = 0.0f; }
    if (iy <= 0.0f) { iy = 0.0f; my = 0.0f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); my = 0.0f; }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float w = ix - fx, e = 1.0f - w, nn = iy - fy, s = 1.0f - nn;
    const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
    const bool inx = (x0 + 1) < W, iny = (y0 + 1) < H, inxy = inx && iny;
    const int o00 = y0 * W + x0;
    const int o01 = inx ? o00 + 1 : o00, o10 = iny ? o00 + W : o00, o11 = inxy ? o00 + W + 1 : o00;
    const size_t plane = (size_t)H * W;
    const float* in = P.in[i] + ((size_t)b * C + c_lo) * plane;
    float* gi = P.g_in[i] ? P.g_in[i] + ((size_t)b * C + c_lo) * plane : nullptr;
    const float* go = P.g_out[i] + ((size_t)b * C + c_lo) * P.n + p;
    float gv[kCPT], v00[kCPT], v01[kCPT], v10[kCPT], v11[kCPT];
#pragma unroll
    for (int c = 0; c < kCPT; ++c) {                       // all loads of the thread in flight together
        const int cc = c_lo + c < c_hi ? c : 0;
        const float* pl = in + (size_t)cc * plane;
        gv[c] = go[(size_t)cc * P.n];
        v00[c] = pl[o00]; v01[c] = pl[o01]; v10[c] = pl[o10]; v11[c] = pl[o11];
    }
    float gx = 0.0f, gy = 0.0f;
#pragma unroll
    for (int c = 0; c < kCPT; ++c) {
        if (c_lo + c < c_hi) {
            const float a01 = inx ? v01[c] : 0.0f, a10 = iny ? v10[c] : 0.0f, a11 = inxy ? v11[c] : 0.0f;
            gx += ((a01 - v00[c]) * s + (a11 - a10) * nn) * gv[c];
            gy += ((a10 - v00[c]) * e + (a11 - a01) * w) * gv[c];
            if (gi) {
                float* gp = gi + (size_t)c * plane;
                unsafeAtomicAdd(gp + o00, nw * gv[c]);
                if (inx) unsafeAtomicAdd(gp + o01, ne * gv[c]);
                if (iny) unsafeAtomicAdd(gp + o10, sw * gv[c]);
                if (inxy) unsafeAtomicAdd(gp + o11, se * gv[c]);
            }
        }
    }
    float* gg = g_grid + ((size_t)b * P.n + p) * 2;
    unsafeAtomicAdd(gg, gx * mx);
    unsafeAtomicAdd(gg + 1, gy * my);
}

// ---- kernel A': the same with the input-gradient planes accumulated in LDS ------------------------------
// Global float atomics top out at ~40 G/s here (the TPS_PP geometry would issue 268 M of them per batch
// of 512: 6 ms).  When `cpt` whole input planes fit in 64 KB of LDS, a workgroup owns (image, input,
// every G-th channel chunk), walks ALL output pixels for each of its chunks, accumulates dL/d input with
// LDS atomics (ds_add_f32) and writes each plane once, coalesced, without a preceding memset; the
// coordinate gradients of its chunks are summed per pixel in LDS too (a thread always owns the same
// pixels) and reach g_grid as ONE pair of global atomics per pixel and workgroup.
// grid = (G groups of input 0 then G groups of input 1, images).
__global__ void __launch_bounds__(1024)
warp_bwd_sample_lds_kernel(const BwdParams P, int G, int cpt0, int cpt1, float* __restrict__ g_grid)
{
    extern __shared__ float smem[];
    float* sgg = smem;                                     // [n][2] coordinate gradients of this workgroup
    float* acc = smem + 2 * P.n;                           // [cpt][H*W]
    const int b = blockIdx.y;
    const int i = (int)blockIdx.x >= G ? 1 : 0;
    const int grp = (int)blockIdx.x - (i ? G : 0);
    const int cpt = i ? cpt1 : cpt0;
    const int H = P.H[i], W = P.W[i], C = P.C[i];
    const int plane = H * W;
    const int chunks = (C + cpt - 1) / cpt;
    const bool want = P.g_in[i] != nullptr;
    for (int e = threadIdx.x; e < 2 * P.n; e += blockDim.x) sgg[e] = 0.0f;
    for (int ch = grp; ch < chunks; ch += G) {
        const int c_lo = ch * cpt;
        const int nc = min(cpt, C - c_lo);
        if (want) {
            for (int e = threadIdx.x; e < nc * plane; e += blockDim.x) acc[e] = 0.0f;
        }
        __syncthreads();
        const float* in = P.in[i] + ((size_t)b * C + c_lo) * plane;
        for (int p = threadIdx.x; p < P.n; p += blockDim.x) {
            const float2 g = reinterpret_cast<const float2*>(P.grid)[(size_t)b * P.n + p];
            float ix = ((g.x + 1.0f) * 0.5f) * (float)(W - 1);
            float iy = ((g.y + 1.0f) * 0.5f) * (float)(H - 1);
            float mx = (float)(W - 1) * 0.5f, my = (float)(H - 1) * 0.5f;
            if (ix <= 0.0f) { ix = 0.0f; mx = 0.0f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); mx = 0.0f; }
            if (iy <= 0.0f) { iy = 0.0f; my = 0.0f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); my = 0.0f; }
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float w = ix - fx, e = 1.0f - w, nn = iy - fy, s = 1.0f - nn;
            const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
            const bool inx = (x0 + 1) < W, iny = (y0 + 1) < H, inxy = inx && iny;
            const int o00 = y0 * W + x0;
            const int o01 = inx ? o00 + 1 : o00, o10 = iny ? o00 + W : o00, o11 = inxy ? o00 + W + 1 : o00;
            const float* go = P.g_out[i] + ((size_t)b * C + c_lo) * P.n + p;
            float gv[kCPT], v00[kCPT], v01[kCPT], v10[kCPT], v11[kCPT];
#pragma unroll
            for (int c = 0; c < kCPT; ++c) {
                const int cc = c < nc ? c : 0;
                const float* pl = in + (size_t)cc * plane;
                gv[c] = go[(size_t)cc * P.n];
                v00[c] = pl[o00]; v01[c] = pl[o01]; v10[c] = pl[o10]; v11[c] = pl[o11];
            }
            float gx = 0.0f, gy = 0.0f;
#pragma unroll
            for (int c = 0; c < kCPT; ++c) {
                if (c < nc) {
                    const float a01 = inx ? v01[c] : 0.0f, a10 = iny ? v10[c] : 0.0f, a11 = inxy ? v11[c] : 0.0f;
                    gx += ((a01 - v00[c]) * s + (a11 - a10) * nn) * gv[c];
                    gy += ((a10 - v00[c]) * e + (a11 - a01) * w) * gv[c];
                    if (want) {
                        float* ap = acc + c * plane;
                        atomicAdd(ap + o00, nw * gv[c]);
                        if (inx) atomicAdd(ap + o01, ne * gv[c]);
                        if (iny) atomicAdd(ap + o10, sw * gv[c]);
                        if (inxy) atomicAdd(ap + o11, se * gv[c]);
                    }
                }
            }
            sgg[2 * p] += gx * mx;                         // this thread owns pixel p in every chunk
            sgg[2 * p + 1] += gy * my;
        }
        __syncthreads();
        if (want) {
            float* gi = P.g_in[i] + ((size_t)b * C + c_lo) * plane;
            for (int e = threadIdx.x; e < nc * plane; e += blockDim.x) gi[e] = acc[e];
        }
    }
    __syncthreads();
    float* gg = g_grid + (size_t)b * P.n * 2;
    for (int e = threadIdx.x; e < 2 * P.n; e += blockDim.x) unsafeAtomicAdd(gg + e, sgg[e]);
}

// ---- kernel B: parameter gradients from g_grid, one workgroup per image -----------------------------------
template <int KMAX>
__global__ void __launch_bounds__(256)
warp_bwd_params_kernel(const BwdParams P, const float* __restrict__ g_grid)
{
    __shared__ float sT[kMaxK * 2];
    __shared__ float sRed[4][kMaxK * 2];
    __shared__ float sGT[kMaxK * 2];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
    const int K = P.F + 3;
    for (int i = tid; i < 2 * K; i += blockDim.x) sT[i] = P.T[(size_t)b * K * 2 + i];
    __syncthreads();

    float aT[KMAX][2];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) aT[k][0] = aT[k][1] = 0.0f;

    for (int p = tid; p < P.n; p += blockDim.x) {
        const float2 gg = reinterpret_cast<const float2*>(g_grid)[(size_t)b * P.n + p];
        const float ggx = gg.x, ggy = gg.y;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                float v;
                if (P.p_xy) {                                  // TPS_PP table: rbf only, [1, P.x, P.y] separate
                    if (k == 0) v = 1.0f;
                    else if (k < 3) v = P.p_xy[(size_t)p * 2 + (k - 1)];
                    else v = P.p_hat_t ? P.p_hat_t[(size_t)(k - 3) * P.n + p] : P.p_hat[(size_t)p * P.p_hat_ld + (k - 3)];
                } else {
                    v = P.p_hat_t ? P.p_hat_t[(size_t)k * P.n + p] : P.p_hat[(size_t)p * P.p_hat_ld + k];
                }
                if (k >= 3 && P.score) {
                    const size_t so = P.score_t ? ((size_t)b * P.F + (k - 3)) * P.n + p
                                                : ((size_t)b * P.n + p) * P.F + (k - 3);
                    const float sc = P.score[so];
                    if (P.g_score) P.g_score[so] = 0.5f * v * (ggx * sT[2 * k] + ggy * sT[2 * k + 1]);
                    v = v * (0.5f * sc + 1.0f);
                }
                aT[k][0] = fmaf(v, ggx, aT[k][0]);
                aT[k][1] = fmaf(v, ggy, aT[k][1]);
            }
        }
    }
    // ---- dL/dT: wavefront shuffle sums, then across the four wavefronts ----
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float a = wave_sum(aT[k][0]), c = wave_sum(aT[k][1]);
            if (lane == 0) { sRed[wv][2 * k] = a; sRed[wv][2 * k + 1] = c; }
        }
    }
    __syncthreads();
    const int nw_ = blockDim.x >> 6;
    for (int i = tid; i < 2 * K; i += blockDim.x) {
        float a = 0.0f;
        for (int w2 = 0; w2 < nw_; ++w2) a += sRed[w2][i];
        sGT[i] = a;
    }
    __syncthreads();
    // ---- dL/dC'[f] = sum_k inv_delta_C[k][f] dL/dT[k] ----
    for (int i = tid; i < 2 * P.F; i += blockDim.x) {
        const int f = i >> 1, xy = i & 1;
        float a = 0.0f;
        for (int k = 0; k < K; ++k) a = fmaf(P.inv_delta_c[(size_t)k * K + f], sGT[2 * k + xy], a);
        P.g_ctrl[(size_t)b * P.F * 2 + i] = a;
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_warp_bwd(const float* g_out0, const float* in0, int C0, int H0, int W0,
                                const float* g_out1, const float* in1, int C1, int H1, int W1,
                                const float* grid, const float* T, const float* inv_delta_c,
                                const float* p_hat, int p_hat_ld, const float* p_xy, const float* score,
                                const float* p_hat_t_or_null, int table_flags, int N, int F, int Ho, int Wo,
                                float* g_in0, float* g_in1, float* g_ctrl, float* g_score, float* g_grid_ws,
                                tpspp_stream_t stream)
{
    TPSPP_REQUIRE(g_out0 && in0 && grid && T && inv_delta_c && p_hat && g_ctrl && g_grid_ws, "tpspp_warp_bwd: null pointer");
    TPSPP_REQUIRE((g_out1 == nullptr) == (in1 == nullptr), "tpspp_warp_bwd: g_out1 and in1 come together");
    TPSPP_REQUIRE(N >= 0 && F > 0 && F + 3 <= kMaxK && Ho > 0 && Wo > 0, "tpspp_warp_bwd: bad sizes (F <= %d)", kMaxK - 3);
    TPSPP_REQUIRE(C0 > 0 && H0 > 0 && W0 > 0 && (!in1 || (C1 > 0 && H1 > 0 && W1 > 0)), "tpspp_warp_bwd: bad input sizes");
    TPSPP_REQUIRE(p_hat_ld >= (p_xy ? F : F + 3), "tpspp_warp_bwd: p_hat_ld too small");
    TPSPP_REQUIRE(!g_score || score, "tpspp_warp_bwd: g_score without score");
    TPSPP_REQUIRE(!g_in1 || in1, "tpspp_warp_bwd: g_in1 without in1");
    if (N == 0) return TPSPP_OK;
    hipStream_t st = tpspp::as_stream(stream);
    BwdParams P;
    P.g_out[0] = g_out0; P.in[0] = in0; P.g_in[0] = g_in0; P.C[0] = C0; P.H[0] = H0; P.W[0] = W0;
    P.g_out[1] = g_out1; P.in[1] = in1; P.g_in[1] = g_in1; P.C[1] = C1; P.H[1] = H1; P.W[1] = W1;
    P.nin = in1 ? 2 : 1;
    P.grid = grid; P.T = T; P.inv_delta_c = inv_delta_c; P.p_hat = p_hat; P.p_hat_ld = p_hat_ld;
    P.p_hat_t = p_hat_t_or_null; P.p_xy = p_xy; P.score = score;
    P.score_t = (score && (table_flags & TPSPP_SCORE_TRANSPOSED)) ? 1 : 0;
    P.g_ctrl = g_ctrl; P.g_score = g_score; P.N = N; P.F = F; P.n = Ho * Wo;
    if (hipMemsetAsync(g_grid_ws, 0, (size_t)N * P.n * 2 * sizeof(float), st) != hipSuccess)
        return tpspp::check_launch("tpspp_warp_bwd(memset)");
    TPSPP_REQUIRE(N <= 65535, "tpspp_warp_bwd: N > 65535");
    const dim3 block(256);
    // LDS-accumulating sampler backward when whole planes fit (64 KB per workgroup), else global atomics
    const size_t kLdsBudget = 64 * 1024;
    const size_t plane0 = (size_t)H0 * W0 * sizeof(float), plane1 = in1 ? (size_t)H1 * W1 * sizeof(float) : 0;
    const size_t gg_bytes = (size_t)P.n * 2 * sizeof(float);
    if (plane0 <= kLdsBudget && plane1 <= kLdsBudget && gg_bytes <= 32 * 1024) {
        const int cpt0 = (int)(kLdsBudget / plane0 < (size_t)kCPT ? kLdsBudget / plane0 : (size_t)kCPT);
        const int cpt1 = in1 ? (int)(kLdsBudget / plane1 < (size_t)kCPT ? kLdsBudget / plane1 : (size_t)kCPT) : 1;
        const int chunks0 = (C0 + cpt0 - 1) / cpt0, chunks1 = in1 ? (C1 + cpt1 - 1) / cpt1 : 0;
        const int most = chunks0 > chunks1 ? chunks0 : chunks1;
        int G = (2048 + N * P.nin - 1) / (N * P.nin);      // enough workgroups for ~4 per CU
        G = G < 1 ? 1 : (G > most ? most : G);
        const size_t accb = (size_t)cpt0 * plane0 > (size_t)cpt1 * plane1 ? (size_t)cpt0 * plane0 : (size_t)cpt1 * plane1;
        const size_t lds = gg_bytes + accb;
        static bool attr_done[tpspp::kMaxDevices] = {};
        if (tpspp::first_use_on_device(attr_done)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_bwd_sample_lds_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(warp_bwd_sample_lds_kernel, dim3((unsigned)(G * P.nin), (unsigned)N), dim3(1024), lds, st, P,
                           G, cpt0, cpt1, g_grid_ws);
    } else {
        if (g_in0 && hipMemsetAsync(g_in0, 0, (size_t)N * C0 * H0 * W0 * sizeof(float), st) != hipSuccess)
            return tpspp::check_launch("tpspp_warp_bwd(memset)");
        if (g_in1 && hipMemsetAsync(g_in1, 0, (size_t)N * C1 * H1 * W1 * sizeof(float), st) != hipSuccess)
            return tpspp::check_launch("tpspp_warp_bwd(memset)");
        const int chunks0 = (C0 + kCPT - 1) / kCPT, chunks1 = in1 ? (C1 + kCPT - 1) / kCPT : 0;
        hipLaunchKernelGGL(warp_bwd_sample_kernel, dim3((unsigned)((P.n + 255) / 256), (unsigned)(chunks0 + chunks1), (unsigned)N),
                           block, 0, st, P, chunks0, g_grid_ws);
    }
    const dim3 grid_dim((unsigned)N);
    if (F + 3 <= 24)      hipLaunchKernelGGL(warp_bwd_params_kernel<24>, grid_dim, block, 0, st, P, g_grid_ws);
    else if (F + 3 <= 36) hipLaunchKernelGGL(warp_bwd_params_kernel<36>, grid_dim, block, 0, st, P, g_grid_ws);
    else                  hipLaunchKernelGGL(warp_bwd_params_kernel<64>, grid_dim, block, 0, st, P, g_grid_ws);
    return tpspp::check_launch("tpspp_warp_bwd");
}
