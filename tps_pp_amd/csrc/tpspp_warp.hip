// TPS grid build + bilinear warp for gfx950 (MI355X): the hot path of TPS++ rectification.
//
// One fused kernel (tps_warp_kernel) does, per workgroup = (tile of output pixels) x (group of
// images):
//   1. stage inv_delta_C in LDS; one wavefront per image solves T = inv_delta_C @ [C';0]: lane i owns
//      row i, the control points are broadcast lane->wave in ascending k with v_readlane (an ORDERED
//      broadcast, never a tree reduction: the fp32 sum must be the k-ascending FMA chain the
//      reference's torch.bmm performs -- BASELINE.md section 2); T goes to LDS, never to HBM;
//   2. every thread keeps the P_hat row of ITS output pixel in registers for the whole image group
//      (P_hat is batch-shared: 294 KB classic / 131 KB TPS_PP, it is read once per group, not once per
//      image) and expands the sampling grid for one image after the other with T read as LDS
//      broadcasts;
//   3. the grid stays in registers; the four bilinear taps of every channel are gathered through
//      L1/L2 (the 32x100 / 32x128 planes are small and the TPS map is smooth, so a wavefront's taps
//      fall in a handful of 128-B lines) and the output row is written fully coalesced.
// HBM traffic = each input plane once + each output plane once; roofline = HBM bandwidth.
//
// Reference call sites replaced: preprocessor/tps_preprocessor.py:71-83 (build_P_prime +
// grid_sample) and backbones/tps_pp/tps_pp.py:597-615 (build_P_prime with score + 2x grid_sample).
//
// Compiled with -ffp-contract=off: the ONLY fused operations are the explicit fmaf() below.
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxK = 64;  // F + 3 <= 64: one lane per row of T in the wave-level solve

struct WarpParams {
    const float* in0; int C0, H0, W0;
    const float* in1; int C1, H1, W1;
    const float* ctrl;
    const float* score;
    const float* inv_delta_c;
    const float* p_hat; int p_hat_ld;
    const float* p_xy;
    int N, F, n;          // n = Ho*Wo
    float* out0; float* out1; float* grid; int32_t* idx;
    int G;                // images per workgroup
    int tiles, chunks;    // grid = tiles * chunks workgroups
    int xcd_map;          // 1: chunks % 8 == 0 -> all tiles of a chunk share an XCD (its L2)
};

struct Taps {
    int o00, o01, o10, o11;   // offsets inside one H x W plane (clamped: always readable)
    float nw, ne, sw, se;
    bool inx, iny;            // is the east column / south row inside the plane
    int x0, y0;
};

// ATen bilinear, padding_mode='border', align_corners=True; weight form and rounding of the CPU
// vector kernel (oracle/tps_oracle.c, weight_form 2).
__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W)
{
    Taps t;
    float ix = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
    float iy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    const float limx = (float)(W - 1), limy = (float)(H - 1);
    ix = (ix > 0.0f) ? ix : 0.0f;   // NaN -> 0
    iy = (iy > 0.0f) ? iy : 0.0f;
    ix = (ix < limx) ? ix : limx;
    iy = (iy < limy) ? iy : limy;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float w = ix - fx, e = 1.0f - w, nn = iy - fy, s = 1.0f - nn;
    t.nw = s * e; t.ne = s * w; t.sw = nn * e; t.se = nn * w;
    t.inx = (x0 + 1) < W;
    t.iny = (y0 + 1) < H;
    const int x1 = t.inx ? x0 + 1 : x0;
    const int y1 = t.iny ? y0 + 1 : y0;
    t.o00 = y0 * W + x0; t.o01 = y0 * W + x1;
    t.o10 = y1 * W + x0; t.o11 = y1 * W + x1;
    t.x0 = x0; t.y0 = y0;
    return t;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ pl, const Taps& t)
{
    float v00 = pl[t.o00];
    float v01 = pl[t.o01];
    float v10 = pl[t.o10];
    float v11 = pl[t.o11];
    v01 = t.inx ? v01 : 0.0f;
    v10 = t.iny ? v10 : 0.0f;
    v11 = (t.inx && t.iny) ? v11 : 0.0f;
    float acc = v00 * t.nw;
    acc = fmaf(v01, t.ne, acc);
    acc = fmaf(v10, t.sw, acc);
    acc = fmaf(v11, t.se, acc);
    return acc;
}

__device__ __forceinline__ void sample_planes(const float* __restrict__ in, float* __restrict__ out,
                                              int C, int HW, int n, const Taps& t)
{
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        const float r0 = bilerp(in + (size_t)(c + 0) * HW, t);
        const float r1 = bilerp(in + (size_t)(c + 1) * HW, t);
        const float r2 = bilerp(in + (size_t)(c + 2) * HW, t);
        const float r3 = bilerp(in + (size_t)(c + 3) * HW, t);
        out[(size_t)(c + 0) * n] = r0;
        out[(size_t)(c + 1) * n] = r1;
        out[(size_t)(c + 2) * n] = r2;
        out[(size_t)(c + 3) * n] = r3;
    }
    if (c + 3 == C) {   // the 3-channel image case: keep all 12 taps in flight
        const float r0 = bilerp(in + (size_t)(c + 0) * HW, t);
        const float r1 = bilerp(in + (size_t)(c + 1) * HW, t);
        const float r2 = bilerp(in + (size_t)(c + 2) * HW, t);
        out[(size_t)(c + 0) * n] = r0;
        out[(size_t)(c + 1) * n] = r1;
        out[(size_t)(c + 2) * n] = r2;
        return;
    }
    for (; c < C; ++c) out[(size_t)c * n] = bilerp(in + (size_t)c * HW, t);
}

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One wavefront: T[i] = sum_q inv[i][q] * Cz[q], q ascending, FMA chain from 0.  Lane i owns row i.
// `inv` may be LDS or global.  Returns (Tx, Ty) of row `lane` (garbage for lane >= K).
__device__ __forceinline__ float2 wave_solve_T(const float* inv, const float* __restrict__ ctrl_b,
                                               int F, int K, int lane)
{
    float cx = 0.0f, cy = 0.0f;             // rows F..F+2 of [C';0] are the appended zeros
    if (lane < F) {
        const float2 c = reinterpret_cast<const float2*>(ctrl_b)[lane];
        cx = c.x; cy = c.y;
    }
    const int row = lane < K ? lane : K - 1;
    const float* h = inv + row * K;
    float ax = 0.0f, ay = 0.0f;
    for (int q = 0; q < K; ++q) {
        const float hv = h[q];
        const float bx = readlane_f(cx, q);
        const float by = readlane_f(cy, q);
        ax = fmaf(hv, bx, ax);
        ay = fmaf(hv, by, ay);
    }
    return make_float2(ax, ay);
}

// FCT > 0: F known at compile time, the pixel's P_hat row lives in registers across the image group.
// FCT == 0: any F (F + 3 <= 64); the row is re-read (L1/L2) for every image.
template <int FCT, bool PXY, bool SCORE>
__global__ void __launch_bounds__(256)
tps_warp_kernel(const WarpParams P)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = FCT > 0 ? FCT : P.F;
    const int K = F + 3;
    float* sInv = smem;                                         // K*K
    float2* sT = reinterpret_cast<float2*>(smem + ((K * K + 3) & ~3));   // G*K

    int chunk, tile;
    {
        const int b = blockIdx.x;
        if (P.xcd_map) {
            const int j = b >> 3;
            chunk = (b & 7) + 8 * (j / P.tiles);
            tile = j % P.tiles;
        } else {
            chunk = b / P.tiles;
            tile = b % P.tiles;
        }
    }
    const int tid = threadIdx.x;
    const int p = tile * blockDim.x + tid;
    const bool live = p < P.n;
    const int pc = live ? p : P.n - 1;
    const int b0 = chunk * P.G;
    const int gcount = min(P.G, P.N - b0);

    // ---- this pixel's P_hat row: issue the loads first, they fly during the T-solve ----
    constexpr int FR = FCT > 0 ? FCT : 1;
    float rbf[FR];
    float r0 = 1.0f, r1, r2;
    {
        const float* ph = P.p_hat + (size_t)pc * P.p_hat_ld;
        if (PXY) {
            const float2 xy = reinterpret_cast<const float2*>(P.p_xy)[pc];
            r1 = xy.x; r2 = xy.y;
        } else {
            r0 = ph[0]; r1 = ph[1]; r2 = ph[2];
            ph += 3;
        }
        if (FCT > 0) {
#pragma unroll
            for (int k = 0; k < FR; ++k) rbf[k] = ph[k];
        }
    }

    // ---- inv_delta_C -> LDS ----
    for (int i = tid; i < K * K; i += blockDim.x) sInv[i] = P.inv_delta_c[i];
    __syncthreads();

    // ---- T-solve: one wavefront per image of the group ----
    {
        const int lane = tid & (kWave - 1), wv = tid / kWave, nw = blockDim.x / kWave;
        for (int g = wv; g < gcount; g += nw) {
            const float2 t = wave_solve_T(sInv, P.ctrl + (size_t)(b0 + g) * F * 2, F, K, lane);
            if (lane < K) sT[g * K + lane] = t;
        }
    }
    __syncthreads();
    if (!live) return;

    const int HW0 = P.H0 * P.W0, HW1 = P.H1 * P.W1;
    for (int g = 0; g < gcount; ++g) {
        const int b = b0 + g;
        const float2* t = sT + g * K;
        float ax = 0.0f, ay = 0.0f;
        {
            const float2 t0 = t[0], t1 = t[1], t2 = t[2];
            ax = fmaf(r0, t0.x, ax); ay = fmaf(r0, t0.y, ay);
            ax = fmaf(r1, t1.x, ax); ay = fmaf(r1, t1.y, ay);
            ax = fmaf(r2, t2.x, ax); ay = fmaf(r2, t2.y, ay);
        }
        const float* srow = SCORE ? P.score + ((size_t)b * P.n + p) * F : nullptr;
        if (FCT > 0) {
            if (SCORE && (FCT % 4 == 0)) {
#pragma unroll
                for (int k4 = 0; k4 < FR / 4; ++k4) {
                    const float4 s4 = reinterpret_cast<const float4*>(srow)[k4];
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k4 * 4 + j;
                        float gq = sv[j] * 0.5f;
                        gq = gq + 1.0f;
                        const float m = rbf[k] * gq;
                        const float2 tk = t[3 + k];
                        ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < FR; ++k) {
                    float m = rbf[k];
                    if (SCORE) {
                        float gq = srow[k] * 0.5f;
                        gq = gq + 1.0f;
                        m = m * gq;
                    }
                    const float2 tk = t[3 + k];
                    ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
                }
            }
        } else {
            const float* ph = P.p_hat + (size_t)p * P.p_hat_ld + (PXY ? 0 : 3);
            for (int k = 0; k < F; ++k) {
                float m = ph[k];
                if (SCORE) {
                    float gq = srow[k] * 0.5f;
                    gq = gq + 1.0f;
                    m = m * gq;
                }
                const float2 tk = t[3 + k];
                ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
            }
        }
        if (P.grid) reinterpret_cast<float2*>(P.grid)[(size_t)b * P.n + p] = make_float2(ax, ay);

        const Taps t0 = make_taps(ax, ay, P.H0, P.W0);
        if (P.idx) reinterpret_cast<int2*>(P.idx)[(size_t)b * P.n + p] = make_int2(t0.x0, t0.y0);
        sample_planes(P.in0 + (size_t)b * P.C0 * HW0, P.out0 + (size_t)b * P.C0 * P.n + p, P.C0,
                      HW0, P.n, t0);
        if (P.in1) {
            const Taps t1 = make_taps(ax, ay, P.H1, P.W1);
            sample_planes(P.in1 + (size_t)b * P.C1 * HW1, P.out1 + (size_t)b * P.C1 * P.n + p,
                          P.C1, HW1, P.n, t1);
        }
    }
}

// ---- stand-alone pieces (same device functions; used for the un-fused API and in tests) ----------
__global__ void __launch_bounds__(256)
solve_T_kernel(const float* __restrict__ inv, const float* __restrict__ ctrl, int N, int F,
               float* __restrict__ T)
{
    const int K = F + 3;
    const int lane = threadIdx.x & (kWave - 1);
    const int b = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= N) return;   // whole wavefronts leave together
    const float2 t = wave_solve_T(inv, ctrl + (size_t)b * F * 2, F, K, lane);
    if (lane < K) reinterpret_cast<float2*>(T)[(size_t)b * K + lane] = t;
}

__global__ void __launch_bounds__(256)
build_grid_kernel(const float* __restrict__ p_hat, int p_hat_ld, const float* __restrict__ p_xy,
                  const float* __restrict__ score, const float* __restrict__ T, int N, int n, int F,
                  float* __restrict__ grid)
{
    const int K = F + 3;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= n) return;
    const float2* t = reinterpret_cast<const float2*>(T) + (size_t)b * K;
    const float* ph = p_hat + (size_t)p * p_hat_ld;
    float r0 = 1.0f, r1, r2;
    if (p_xy) { r1 = p_xy[2 * p]; r2 = p_xy[2 * p + 1]; }
    else { r0 = ph[0]; r1 = ph[1]; r2 = ph[2]; ph += 3; }
    float ax = 0.0f, ay = 0.0f;
    ax = fmaf(r0, t[0].x, ax); ay = fmaf(r0, t[0].y, ay);
    ax = fmaf(r1, t[1].x, ax); ay = fmaf(r1, t[1].y, ay);
    ax = fmaf(r2, t[2].x, ax); ay = fmaf(r2, t[2].y, ay);
    const float* srow = score ? score + ((size_t)b * n + p) * F : nullptr;
    for (int k = 0; k < F; ++k) {
        float m = ph[k];
        if (srow) {
            float gq = srow[k] * 0.5f;
            gq = gq + 1.0f;
            m = m * gq;
        }
        const float2 tk = t[3 + k];
        ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
    }
    reinterpret_cast<float2*>(grid)[(size_t)b * n + p] = make_float2(ax, ay);
}

__global__ void __launch_bounds__(256)
grid_sample_kernel(const float* __restrict__ in, const float* __restrict__ grid, int N, int C, int H,
                   int W, int n, float* __restrict__ out, int32_t* __restrict__ idx)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= n) return;
    const float2 g = reinterpret_cast<const float2*>(grid)[(size_t)b * n + p];
    const Taps t = make_taps(g.x, g.y, H, W);
    if (idx) reinterpret_cast<int2*>(idx)[(size_t)b * n + p] = make_int2(t.x0, t.y0);
    sample_planes(in + (size_t)b * C * H * W, out + (size_t)b * C * n + p, C, H * W, n, t);
}

int g_tune_G = 0, g_tune_tpb = 0;

template <int FCT>
void launch_warp(const WarpParams& P, dim3 grid, dim3 block, size_t lds, hipStream_t st)
{
    const bool pxy = P.p_xy != nullptr, sc = P.score != nullptr;
    if (pxy && sc)       hipLaunchKernelGGL((tps_warp_kernel<FCT, true, true>), grid, block, lds, st, P);
    else if (pxy && !sc) hipLaunchKernelGGL((tps_warp_kernel<FCT, true, false>), grid, block, lds, st, P);
    else if (!pxy && sc) hipLaunchKernelGGL((tps_warp_kernel<FCT, false, true>), grid, block, lds, st, P);
    else                 hipLaunchKernelGGL((tps_warp_kernel<FCT, false, false>), grid, block, lds, st, P);
}

}  // namespace

TPSPP_EXPORT int tpspp_warp_set_tuning(int images_per_group, int threads_per_group)
{
    TPSPP_REQUIRE(images_per_group >= 0 && images_per_group <= 64, "images_per_group out of range");
    TPSPP_REQUIRE(threads_per_group == 0 || (threads_per_group % 64 == 0 && threads_per_group >= 64 &&
                                             threads_per_group <= 256),
                  "threads_per_group must be 0 or a multiple of 64 in [64, 256]");
    g_tune_G = images_per_group;
    g_tune_tpb = threads_per_group;
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_solve_T(const float* inv_delta_c, const float* ctrl, int N, int F, float* T,
                               tpspp_stream_t stream)
{
    TPSPP_REQUIRE(inv_delta_c && ctrl && T, "tpspp_solve_T: null pointer");
    TPSPP_REQUIRE(N >= 0 && F > 0 && F + 3 <= kMaxK, "tpspp_solve_T: need N >= 0, 0 < F <= %d", kMaxK - 3);
    if (N == 0) return TPSPP_OK;
    const int wpb = 4;
    hipLaunchKernelGGL(solve_T_kernel, dim3((N + wpb - 1) / wpb), dim3(wpb * kWave), 0,
                       tpspp::as_stream(stream), inv_delta_c, ctrl, N, F, T);
    return tpspp::check_launch("tpspp_solve_T");
}

TPSPP_EXPORT int tpspp_build_grid(const float* p_hat, int p_hat_ld, const float* p_xy,
                                  const float* score, const float* T, int N, int n, int F,
                                  float* grid, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(p_hat && T && grid, "tpspp_build_grid: null pointer");
    TPSPP_REQUIRE(N >= 0 && n > 0 && F > 0, "tpspp_build_grid: bad sizes");
    TPSPP_REQUIRE(p_hat_ld >= (p_xy ? F : F + 3), "tpspp_build_grid: p_hat_ld too small");
    TPSPP_REQUIRE(N <= 65535, "tpspp_build_grid: N > 65535");
    if (N == 0) return TPSPP_OK;
    hipLaunchKernelGGL(build_grid_kernel, dim3((n + 255) / 256, N), dim3(256), 0,
                       tpspp::as_stream(stream), p_hat, p_hat_ld, p_xy, score, T, N, n, F, grid);
    return tpspp::check_launch("tpspp_build_grid");
}

TPSPP_EXPORT int tpspp_grid_sample(const float* in, const float* grid, int N, int C, int H, int W,
                                   int Ho, int Wo, float* out, int32_t* idx_or_null,
                                   tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && grid && out, "tpspp_grid_sample: null pointer");
    TPSPP_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "tpspp_grid_sample: bad sizes");
    TPSPP_REQUIRE(N <= 65535, "tpspp_grid_sample: N > 65535");
    if (N == 0) return TPSPP_OK;
    const int n = Ho * Wo;
    hipLaunchKernelGGL(grid_sample_kernel, dim3((n + 255) / 256, N), dim3(256), 0,
                       tpspp::as_stream(stream), in, grid, N, C, H, W, n, out, idx_or_null);
    return tpspp::check_launch("tpspp_grid_sample");
}

TPSPP_EXPORT int tpspp_warp_fwd(const float* in0, int C0, int H0, int W0,
                                const float* in1, int C1, int H1, int W1,
                                const float* ctrl, const float* score,
                                const float* inv_delta_c, const float* p_hat, int p_hat_ld,
                                const float* p_xy, int N, int F, int Ho, int Wo,
                                float* out0, float* out1, float* grid_or_null, int32_t* idx_or_null,
                                tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in0 && ctrl && inv_delta_c && p_hat && out0, "tpspp_warp_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C0 > 0 && H0 > 0 && W0 > 0 && Ho > 0 && Wo > 0, "tpspp_warp_fwd: bad sizes");
    TPSPP_REQUIRE(F > 0 && F + 3 <= kMaxK, "tpspp_warp_fwd: need 0 < F <= %d", kMaxK - 3);
    TPSPP_REQUIRE(p_hat_ld >= (p_xy ? F : F + 3), "tpspp_warp_fwd: p_hat_ld too small");
    TPSPP_REQUIRE((in1 == nullptr) == (out1 == nullptr), "tpspp_warp_fwd: in1/out1 must come together");
    if (in1) TPSPP_REQUIRE(C1 > 0 && H1 > 0 && W1 > 0, "tpspp_warp_fwd: bad in1 sizes");
    if (N == 0) return TPSPP_OK;

    WarpParams P;
    P.in0 = in0; P.C0 = C0; P.H0 = H0; P.W0 = W0;
    P.in1 = in1; P.C1 = in1 ? C1 : 0; P.H1 = in1 ? H1 : 1; P.W1 = in1 ? W1 : 1;
    P.ctrl = ctrl; P.score = score; P.inv_delta_c = inv_delta_c;
    P.p_hat = p_hat; P.p_hat_ld = p_hat_ld; P.p_xy = p_xy;
    P.N = N; P.F = F; P.n = Ho * Wo;
    P.out0 = out0; P.out1 = out1; P.grid = grid_or_null; P.idx = idx_or_null;

    const int tpb = g_tune_tpb > 0 ? g_tune_tpb : 256;
    P.tiles = (P.n + tpb - 1) / tpb;
    int G = g_tune_G;
    if (G <= 0) {
        // largest group that still leaves >= ~3 workgroups per CU (256 CUs) in flight
        G = 16;
        while (G > 1 && (long)P.tiles * ((N + G - 1) / G) < 768) G >>= 1;
    }
    if (G > N) G = N;
    P.G = G;
    P.chunks = (N + G - 1) / G;
    P.xcd_map = (P.chunks % 8 == 0) ? 1 : 0;
    const int K = F + 3;
    const size_t lds = (size_t)(((K * K + 3) & ~3) + 2 * G * K) * sizeof(float);
    const dim3 grid((unsigned)(P.tiles * P.chunks)), block(tpb);
    hipStream_t st = tpspp::as_stream(stream);
    if (F == 20)      launch_warp<20>(P, grid, block, lds, st);
    else if (F == 32) launch_warp<32>(P, grid, block, lds, st);
    else              launch_warp<0>(P, grid, block, lds, st);
    return tpspp::check_launch("tpspp_warp_fwd");
}
