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
#include <new>
#include "tpspp_warp_dev.h"
#include "tpspp_warp_stream.h"
#include "tpspp_warp_pair.h"
#include "tpspp_warp_img.h"
#include "tpspp_warp_geo_launch.h"

#include <cstring>

namespace {

using namespace tpspp_dev;

struct WarpParams {
    const float* in0; int C0, H0, W0;
    const float* in1; int C1, H1, W1;
    const float* ctrl;
    const float* score; int score_t;   // score_t: 1 = (N, F, n) layout
    const float* inv_delta_c;
    const float* p_hat; int p_hat_ld;
    const float* p_xy;
    const float* p_hat_t; // (cols, n) transposed copy of p_hat or nullptr
    int N, F, n;          // n = Ho*Wo
    float* out0; float* out1; float* grid; int32_t* idx;
    int G;                // images per workgroup
    int tiles, chunks;    // grid = tiles * chunks workgroups
    int xcd_map;          // 1: chunks % 8 == 0 -> all tiles of a chunk share an XCD (its L2)
};


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
        const float* pt = P.p_hat_t ? P.p_hat_t + pc : nullptr;   // column pc of (cols, n): coalesced
        if (PXY) {
            const float2 xy = reinterpret_cast<const float2*>(P.p_xy)[pc];
            r1 = xy.x; r2 = xy.y;
        } else {
            if (pt) { r0 = pt[0]; r1 = pt[P.n]; r2 = pt[2 * (size_t)P.n]; pt += 3 * (size_t)P.n; }
            else    { r0 = ph[0]; r1 = ph[1]; r2 = ph[2]; }
            ph += 3;
        }
        if (FCT > 0) {
            if (pt) {
#pragma unroll
                for (int k = 0; k < FR; ++k) rbf[k] = pt[(size_t)k * P.n];
            } else {
#pragma unroll
                for (int k = 0; k < FR; ++k) rbf[k] = ph[k];
            }
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
        // score element (b, p, k) sits at sbase[k * sk]: sk = 1 in the reference's (N, n, F) layout
        const float* srow = nullptr;
        size_t sk = 1;
        if (SCORE) {
            if (P.score_t) { srow = P.score + (size_t)b * F * P.n + p; sk = (size_t)P.n; }
            else           { srow = P.score + ((size_t)b * P.n + p) * F; }
        }
        if (FCT > 0) {
            if (SCORE && (FCT % 4 == 0) && !P.score_t) {
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
                        float gq = srow[k * sk] * 0.5f;
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
                    float gq = srow[k * sk] * 0.5f;
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


// ---- LDS-staged kernel: small single-input images (classic TPS-STN, 2*C*H*W*4 B <= ~150 KB) -------
// One workgroup rectifies TWO images (adjacent in memory) with specialised wavefronts:
//   * loader wavefronts (the last kLdsLoaders of the group) do nothing but stream the image pair
//     HBM -> LDS with global_load_lds_dwordx4 (1 KB per wavefront instruction, no register round
//     trip), all of it issued in the first few hundred cycles;
//   * compute wavefronts solve T (wavefronts 0 and 1, one image each), expand the sampling grid of
//     their pixels for BOTH images from the transposed P_hat (one coalesced 256-B row segment per
//     instruction; each value feeds four FMA chains: x/y of image A and of image B) while the images
//     are still in flight, then take the bilinear taps from LDS and store fully coalesced rows.
// vmcnt retires in order per wavefront; keeping the long-latency image fetch in other wavefronts
// than the P_hat reads is what lets the grid expansion overlap the HBM latency.
// Every input byte crosses HBM -> CU exactly once.
struct LdsParams {
    const float* in; int C, H, W;
    const float* ctrl; const float* inv_delta_c; const float* p_hat_t;
    int N, n, Ho, Wo;
    float* out; float* grid; int32_t* idx;
    int compute_waves;  // wavefronts [0, compute_waves) own pixels, the rest load
    int band_px;        // output pixels per band (a workgroup = one image pair x one band)
    int bands;
    int zero_off;       // mirror kernel: float offset (from the staged pair) of the zero words for OOB taps
    long long* trace;   // optional: 8 shader-clock stamps per workgroup (tpspp_warp_set_trace)
};



constexpr int kLdsMaxWaves = 16;
constexpr int kLdsLoaders = 3;
constexpr int kLdsFirstBurst = 8;   // DMA pieces per loader issued before the T barrier


constexpr int kPrefetchK = 4;   // P_hat^T rows fetched before T is known (hides one L2 round trip)

// HC/WC: input height/width known at compile time (0 = read them from the parameters)
template <int F, int C, int PPT, int HC, int WC>
__global__ void __launch_bounds__(kLdsMaxWaves * kWave, (PPT <= 2 ? 8 : 4))   // PPT <= 2: two groups per CU
tps_warp_lds_kernel(const LdsParams P)
{
    constexpr int K = F + 3;
    const int H = HC > 0 ? HC : P.H;
    const int W = WC > 0 ? WC : P.W;
    constexpr int PF = kPrefetchK < K ? kPrefetchK : K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [ sT: K x float4 (TxA,TyA,TxB,TyB) | sInv: K*K (padded) | image pair, contiguous, + slack ]
    float4* sT = reinterpret_cast<float4*>(smem);
    float* sInv = smem + 4 * K;
    float* sImg = sInv + ((K * K + 3) & ~3);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    // band-major block order: the bands of one pair are `npairs` blocks apart, i.e. on the same XCD
    // whenever npairs % 8 == 0, so the second band's image fetch hits that XCD's L2
    const int npairs = (P.N + 1) >> 1;
    const int band = blockIdx.x / npairs;
    const int pair = blockIdx.x - band * npairs;
    const int b0 = pair * 2;
    const bool hasB = (b0 + 1) < P.N;
    const int HW = H * W;
    const int img_elems = C * HW;                          // multiple of 4 (checked on the host)

    if (wv == 0) stamp(P.trace, 0);
    if (wv >= P.compute_waves) {
        // ================= loader wavefronts =================
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;               // tail lanes re-read a valid address
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, 0);
        };
        // a first burst that the memory queue accepts without stalling the issue, then the group's
        // T barrier (so the compute wavefronts are not held up behind a full queue), then the rest
        int piece = wv - P.compute_waves;
        for (int i = 0; i < kLdsFirstBurst && piece < pieces; ++i, piece += kLdsLoaders) dma(piece);
        lds_only_barrier();      // matches the T barrier of the compute wavefronts
        for (; piece < pieces; piece += kLdsLoaders) dma(piece);
        if (wv == P.compute_waves) stamp(P.trace, 5);            // DMA issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv == P.compute_waves) stamp(P.trace, 6);            // DMA landed
        __syncthreads();         // this wavefront's DMA has landed; release the group
        return;
    }

    // ================= compute wavefronts =================
    const int tpc = P.compute_waves * kWave;               // pixel stride between a thread's slots
    const int band_lo = band * P.band_px;
    const int band_hi = min(P.n, band_lo + P.band_px);

    // ---- T-solve inputs: wavefront 0 -> image A, wavefront 1 -> image B; lane i owns row i ----
    // inv_delta_C is read COALESCED (every workgroup of the launch wants these same 2 KB at the same
    // moment: a row-per-lane read would put ~20x more requests on the few L2 lines that hold them)
    // and turned into one row per lane through LDS.  Both wavefronts write identical values.
    constexpr int KK = K * K;
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;                            // rows F..F+2 of [C';0] stay zero
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
        if (lane < F) {
            const int b = (wv == 1 && hasB) ? b0 + 1 : b0;
            const float2 c = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[lane];
            cx = c.x; cy = c.y;
        }
    }

    // ---- this thread's pixels; first P_hat^T rows requested before T exists ----
    // (pixel index kept as an unsigned BYTE offset: loads and stores then use the scalar-base +
    //  32-bit-VGPR-offset addressing mode and need no 64-bit address arithmetic on the VALU)
    unsigned poff[PPT];
    bool live[PPT];
    float pre[PF][PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = band_lo + tid + j * tpc;
        live[j] = p < band_hi;
        poff[j] = 4u * (unsigned)(live[j] ? p : band_hi - 1);
    }
    const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
    const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            pre[k][j] = *reinterpret_cast<const float*>(pht + k * row_bytes + poff[j]);

    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;   // stride K is odd: no conflicts
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
    }
    lds_only_barrier();
    if (wv == 0) stamp(P.trace, 1);                               // T ready

    // ---- sampling grid for both images: k-ascending FMA chains, T broadcast from LDS ----
    float gxA[PPT], gyA[PPT], gxB[PPT], gyB[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) gxA[j] = gyA[j] = gxB[j] = gyB[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const float4 t = sT[k];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            gxA[j] = fmaf(pre[k][j], t.x, gxA[j]);
            gyA[j] = fmaf(pre[k][j], t.y, gyA[j]);
            gxB[j] = fmaf(pre[k][j], t.z, gxB[j]);
            gyB[j] = fmaf(pre[k][j], t.w, gyB[j]);
        }
    }
    // unroll 8 keeps 8*PPT independent coalesced row reads in flight; a full unroll makes the
    // scheduler hoist every load and spill
#pragma unroll 8
    for (int k = PF; k < K; ++k) {
        const float4 t = sT[k];
        const char* row = pht + k * row_bytes;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float ph = *reinterpret_cast<const float*>(row + poff[j]);
            gxA[j] = fmaf(ph, t.x, gxA[j]);
            gyA[j] = fmaf(ph, t.y, gyA[j]);
            gxB[j] = fmaf(ph, t.z, gxB[j]);
            gyB[j] = fmaf(ph, t.w, gyB[j]);
        }
    }

    // pin the finished grid here (the optimiser would otherwise sink the chains below the barrier)
#pragma unroll
    for (int j = 0; j < PPT; ++j) asm volatile("" : "+v"(gxA[j]), "+v"(gyA[j]), "+v"(gxB[j]), "+v"(gyB[j]));

    // ---- the image pair must have landed (loaders wait on their DMA before this barrier) ----
    if (wv == 0) stamp(P.trace, 2);                               // grid expanded
    __syncthreads();
    if (wv == 0) stamp(P.trace, 3);                               // images in LDS

    // ---- bilinear taps from LDS, coalesced stores ----
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const bool store = live[j] && (im == 0 || hasB);
            const int b = b0 + im;
            const float gx = im ? gxB[j] : gxA[j], gy = im ? gyB[j] : gyA[j];
            const Taps t = make_taps(gx, gy, H, W);
            if (P.grid && store)
                *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid + (size_t)b * P.n * 2) +
                                           2 * poff[j]) = make_float2(gx, gy);
            if (P.idx && store)
                *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx + (size_t)b * P.n * 2) +
                                         2 * poff[j]) = make_int2(t.x0, t.y0);
            const float* img = sImg + im * img_elems;
            char* o = reinterpret_cast<char*>(P.out + (size_t)b * C * P.n);   // wave-uniform base
            const bool inxy = t.inx && t.iny;
            float res[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* pl = img + c * HW;
                // the east neighbour is read unconditionally (the staged pair is followed by
                // slack) and masked afterwards; o10 is already clamped to the last row
                const float v00 = pl[t.o00];
                float v01 = pl[t.o00 + 1];
                float v10 = pl[t.o10];
                float v11 = pl[t.o10 + 1];
                v01 = t.inx ? v01 : 0.0f;
                v10 = t.iny ? v10 : 0.0f;
                v11 = inxy ? v11 : 0.0f;
                float acc = v00 * t.nw;
                acc = fmaf(v01, t.ne, acc);
                acc = fmaf(v10, t.sw, acc);
                acc = fmaf(v11, t.se, acc);
                res[c] = acc;
            }
            if (store) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    *reinterpret_cast<float*>(o + c * row_bytes + poff[j]) = res[c];
            }
        }
    }
    if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(P.trace, 4);                                        // this wavefront's stores retired
    }
}

// ---- LDS-staged kernel, mirror-symmetric table --------------------------------------------------
// The reference's RBF table has an exact 4-fold mirror symmetry (the fiducials sit symmetrically on
// the top/bottom edges, the pixel centres symmetrically in the rectified image):
//     row(Ho-1-r, c)   = [1,  P.x, -P.y, rbf[(k + F/2) mod F]]
//     row(r, Wo-1-c)   = [1, -P.x,  P.y, rbf[mirror of k inside its half]]
// bit for bit in fp32 (the caller verifies that on the host before passing TPSPP_TABLE_MIRROR4).
// A thread therefore loads ONE table row (F+3 coalesced reads, issued at kernel entry) and expands
// the grid of its FOUR mirror pixels for BOTH images of the pair: 16 FMA chains per loaded value
// instead of 4, i.e. a quarter of the table traffic through the vector memory pipeline, which is
// what bounds the un-mirrored kernel at batch 512 (two images per CU against a 294 KB table).
// Each chain is still the k-ascending fp32 FMA chain from zero of its own pixel.
template <int F>
__device__ __forceinline__ constexpr int perm_y(int k) { return (k + F / 2) % F; }
template <int F>
__device__ __forceinline__ constexpr int perm_x(int k) { return k < F / 2 ? F / 2 - 1 - k : F + F / 2 - 1 - k; }

template <int F, int C, int HC, int WC, bool AUX>
__global__ void __launch_bounds__(kLdsMaxWaves * kWave)
tps_warp_lds_mirror_kernel(const LdsParams P)
{
    constexpr int K = F + 3;
    const int H = HC > 0 ? HC : P.H;
    const int W = WC > 0 ? WC : P.W;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;                            // K*K (padded to 4)
    float* sImg = sInv + ((K * K + 3) & ~3);               // image pair, contiguous, + slack
    float* sZero = sImg + P.zero_off;                      // C zero words, H*W apart, behind the DMA pieces

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int npairs = (P.N + 1) >> 1;
    const int band = blockIdx.x / npairs;                  // band-major: see tps_warp_lds_kernel
    const int pair = blockIdx.x - band * npairs;
    const int b0 = pair * 2;
    const bool hasB = (b0 + 1) < P.N;
    const int HW = H * W;
    const int img_elems = C * HW;

    if (wv == 0) stamp(P.trace, 0);
    if (wv >= P.compute_waves) {
        // ================= loader wavefronts =================
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;               // tail lanes re-read a valid address
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, 0);
        };
        int piece = wv - P.compute_waves;
        for (int i = 0; i < kLdsFirstBurst && piece < pieces; ++i, piece += kLdsLoaders) dma(piece);
        lds_only_barrier();      // matches the T barrier of the compute wavefronts
        for (; piece < pieces; piece += kLdsLoaders) dma(piece);
        if (wv == P.compute_waves) stamp(P.trace, 5);            // DMA issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv == P.compute_waves) stamp(P.trace, 6);            // DMA landed
        __syncthreads();
        return;
    }

    // ================= compute wavefronts =================
    // this thread's quadrant pixel (r, c), r < Ho/2, c < Wo/2, and its three mirror images
    const int halfW = P.Wo >> 1;
    const int nq = (P.Ho >> 1) * halfW;
    const int q_hi = min(nq, (band + 1) * P.band_px);
    const int qp_raw = band * P.band_px + tid;
    const bool live = qp_raw < q_hi;
    const int qp = live ? qp_raw : q_hi - 1;
    const int r = qp / halfW, c = qp - r * halfW;
    unsigned poff[4];                                      // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * P.Wo + c);                           // (r, c)
    poff[1] = 4u * (unsigned)(r * P.Wo + (P.Wo - 1 - c));              // x-mirror
    poff[2] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + c);              // y-mirror
    poff[3] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + (P.Wo - 1 - c)); // both

    // ---- T-solve inputs: wavefront 0 -> image A, wavefront 1 -> image B; lane i owns row i ----
    // inv_delta_C is read COALESCED (every workgroup of the launch wants these same 2 KB at the same
    // moment: a row-per-lane read would put ~20x more requests on the few L2 lines that hold them)
    // and turned into one row per lane through LDS.  Both wavefronts write identical values.
    constexpr int KK = K * K;
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;                            
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
        if (lane < F) {
            const int b = (wv == 1 && hasB) ? b0 + 1 : b0;
            const float2 cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[lane];
            cx = cc.x; cy = cc.y;
        }
    }

    // ---- the table row of (r, c): all K values requested now, consumed after the T barrier ----
    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }

    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;   // stride K is odd: no conflicts
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;                           // read by out-of-image taps
    lds_only_barrier();
    if (wv == 0) stamp(P.trace, 1);                               // T ready

    // ---- 16 FMA chains: 4 mirror pixels x (image A, image B) x (x, y), each k-ascending ----
    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
        float val[4];
        if constexpr (q == 0) {
            val[0] = val[1] = val[2] = val[3] = v[0];
        } else if constexpr (q == 1) {                     // P.x flips under the x-mirror
            val[0] = v[1]; val[1] = -v[1]; val[2] = v[1]; val[3] = -v[1];
        } else if constexpr (q == 2) {                     // P.y flips under the y-mirror
            val[0] = v[2]; val[1] = v[2]; val[2] = -v[2]; val[3] = -v[2];
        } else {
            constexpr int k = q - 3;
            val[0] = v[3 + k];
            val[1] = v[3 + perm_x<F>(k)];
            val[2] = v[3 + perm_y<F>(k)];
            val[3] = v[3 + perm_x<F>(perm_y<F>(k))];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
    // pin the finished grid HERE: without this the optimiser sinks the chains below the image
    // barrier, next to their first use, and the grid expansion no longer overlaps the image fetch
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));
    }

    if (wv == 0) stamp(P.trace, 2);                               // grid expanded
    __syncthreads();                                              // image pair is in LDS
    if (wv == 0) stamp(P.trace, 3);

    // ---- bilinear taps from LDS, coalesced stores ----
    // Every store is  wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: no 64-bit
    // vector address arithmetic in this VALU-bound phase.
    const size_t row_bytes = (size_t)P.n * 4;
    if (live) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            if (im == 1 && !hasB) break;                   // wave-uniform
            const int b = b0 + im;
            const float* img = sImg + im * img_elems;
            const float* zero_im = sZero;                  // + ch*HW: this image's zero words
            typedef __attribute__((address_space(1))) char gchar;
            typedef __attribute__((address_space(1))) float gfloat;
            gchar* oc[C];                                  // per-plane bases, kept in SGPR pairs
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                oc[ch] = (gchar*)(P.out) + ((size_t)b * C + ch) * row_bytes;
                asm volatile("" : "+s"(oc[ch]));
            }
            char* og = AUX ? reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes : nullptr;
            char* oi = AUX ? reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes : nullptr;
            // Four tap pointers per pixel; a tap outside the image points at a zero word kept behind the
            // staged pair (one per channel plane, HW apart), so a channel is four LDS reads with immediate
            // offsets and no per-channel select or address arithmetic.  All 16*C reads of the image's four
            // mirror pixels are issued before the first is consumed: with ~3 wavefronts per SIMD the
            // LDS latency would otherwise be paid once per pixel.
            const float* tp[4][4];
            float tw[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
                if constexpr (AUX) {
                    if (P.grid) *reinterpret_cast<float2*>(og + 2u * poff[m]) = make_float2(gx[m][im], gy[m][im]);
                    if (P.idx) *reinterpret_cast<int2*>(oi + 2u * poff[m]) = make_int2(t.x0, t.y0);
                }
                const bool inxy = t.inx && t.iny;
                tp[m][0] = img + t.o00;
                tp[m][1] = t.inx ? tp[m][0] + 1 : zero_im;
                tp[m][2] = t.iny ? img + t.o10 : zero_im;
                tp[m][3] = inxy ? img + t.o10 + 1 : zero_im;
                tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
            }
            float tv[4][C][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    float acc = tv[m][ch][0] * tw[m][0];
                    acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                    acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                    acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                    *(gfloat*)(oc[ch] + poff[m]) = acc;
                }
            }
        }
    }
    if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(P.trace, 4);
    }
}


__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ src, int ld, int n, int cols, float* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // index into dst (cols, n)
    if (i >= n * cols) return;
    const int c = i / n, p = i - c * n;
    dst[i] = src[(size_t)p * ld + c];
}

int g_tune_G = 0, g_tune_tpb = 0, g_tune_kernel = 0, g_tune_bands = 0, g_tune_mirror = 0;
long long* g_trace = nullptr;

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

TPSPP_EXPORT int tpspp_warp_set_tuning(int images_per_group, int threads_per_group, int kernel_choice,
                                       int bands)
{
    TPSPP_REQUIRE(kernel_choice >= 0 && kernel_choice <= 9, "kernel_choice must be 0..9");
    // kernel_choice 7 (run-time-geometry kernel): bits 0-2 = workgroups per image, bit 3 = never an image pair; every other
    // choice: workgroups per image (pair) of the LDS-staged / mirror kernels, built and tested for 0..8
    // kernel_choice 8 (row bands with span staging): bits 0-5 = workgroups per image (0 = heuristic), bit 6 (64) = every
    // workgroup on its global-memory path, bit 7 (128) = the measured-span form instead of the windows requested at launch,
    // bits 8-15 = LDS budget per workgroup in KB (0 = 38)
    TPSPP_REQUIRE(bands >= 0 && bands <= (kernel_choice == 8 ? 0xffff : (kernel_choice == 7 || kernel_choice == 9) ? 15 : 8),
                  "bands must be in [0, 8] (kernel_choice 7 / 9: [0, 15], bit 3 = never an image pair; 8: see tpspp.h)");
    g_tune_kernel = kernel_choice % 10 == 3 ? 2 : kernel_choice;
    g_tune_mirror = kernel_choice == 3 ? 2 : 0;       // 3: LDS-staged kernel WITHOUT the mirror trick
    tpspp::geo_set_bands((kernel_choice == 7 || kernel_choice == 9) ? bands : 0);
    tpspp::span_set_tuning(kernel_choice == 8 ? (bands & 63) : 0, kernel_choice == 8 ? ((bands >> 6) & 1) : 0,
                           kernel_choice == 8 ? ((bands >> 8) & 255) : 0, kernel_choice == 8 ? ((bands >> 7) & 1) : 0);
    g_tune_bands = bands;
    TPSPP_REQUIRE(images_per_group >= 0 && images_per_group <= 64, "images_per_group out of range");
    TPSPP_REQUIRE(threads_per_group == 0 || (threads_per_group % 64 == 0 && threads_per_group >= 64 &&
                                             threads_per_group <= 256),
                  "threads_per_group must be 0 or a multiple of 64 in [64, 256]");
    g_tune_G = images_per_group;
    g_tune_tpb = threads_per_group;
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_table_mirror_symmetry(const float* p_hat_host, int p_hat_ld, int Ho, int Wo, int F)
{
    // HOST memory, classic (n, F+3) layout.  Bitwise check of the relations the mirror kernel uses.
    if (!p_hat_host || Ho <= 0 || Wo <= 0 || F <= 0 || (F & 1) || (Ho & 1) || (Wo & 1) || p_hat_ld < F + 3)
        return 0;
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto px = [&](int k) { return k < F / 2 ? F / 2 - 1 - k : F + F / 2 - 1 - k; };
    auto py = [&](int k) { return (k + F / 2) % F; };
    for (int r = 0; r < Ho / 2; ++r)
        for (int c = 0; c < Wo / 2; ++c) {
            const float* a = p_hat_host + (size_t)(r * Wo + c) * p_hat_ld;
            const float* bx = p_hat_host + (size_t)(r * Wo + (Wo - 1 - c)) * p_hat_ld;
            const float* by = p_hat_host + (size_t)((Ho - 1 - r) * Wo + c) * p_hat_ld;
            const float* bxy = p_hat_host + (size_t)((Ho - 1 - r) * Wo + (Wo - 1 - c)) * p_hat_ld;
            if (bits(bx[0]) != bits(a[0]) || bits(by[0]) != bits(a[0]) || bits(bxy[0]) != bits(a[0])) return 0;
            if (bits(bx[1]) != bits(-a[1]) || bits(by[1]) != bits(a[1]) || bits(bxy[1]) != bits(-a[1])) return 0;
            if (bits(bx[2]) != bits(a[2]) || bits(by[2]) != bits(-a[2]) || bits(bxy[2]) != bits(-a[2])) return 0;
            for (int k = 0; k < F; ++k) {
                if (bits(bx[3 + k]) != bits(a[3 + px(k)])) return 0;
                if (bits(by[3 + k]) != bits(a[3 + py(k)])) return 0;
                if (bits(bxy[3 + k]) != bits(a[3 + px(py(k))])) return 0;
            }
        }
    return 1;
}

TPSPP_EXPORT int tpspp_warp_set_trace(long long* device_buf)
{
    g_trace = device_buf;
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

TPSPP_EXPORT int tpspp_transpose_p_hat(const float* p_hat, int p_hat_ld, int n, int cols,
                                       float* p_hat_t, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(p_hat && p_hat_t, "tpspp_transpose_p_hat: null pointer");
    TPSPP_REQUIRE(n > 0 && cols > 0 && p_hat_ld >= cols, "tpspp_transpose_p_hat: bad sizes");
    TPSPP_REQUIRE((long)n * cols < (1L << 31), "tpspp_transpose_p_hat: table too large");
    const int total = n * cols;
    hipLaunchKernelGGL(transpose_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       tpspp::as_stream(stream), p_hat, p_hat_ld, n, cols, p_hat_t);
    return tpspp::check_launch("tpspp_transpose_p_hat");
}

namespace {
// Quadrant pixels per thread of the packed table: tpspp::geo_qp (tpspp_warp_geo.hip).  32x100 -> 1 (the image-pair
// kernel's layout), 32x128 -> 2, 48x160 -> 3; geometries with more than 13 wavefronts of quadrant pixels: up to 4.
int img_qp(int Ho, int Wo) { return tpspp::geo_qp(Ho, Wo); }
int img_nthr(int Ho, int Wo, int QP) { return tpspp::geo_nthr(Ho, Wo, QP); }
}  // namespace

TPSPP_EXPORT size_t tpspp_prepared_table_floats(int Ho, int Wo, int F)
{
    const int QP = img_qp(Ho, Wo);
    if (QP == 0 || F <= 0 || F + 3 > kMaxK) return 0;
    const int K = F + 3, KG = (K + 3) / 4;
    const int NW = (img_nthr(Ho, Wo, QP) + kWave - 1) / kWave;
    // + the span kernel's copy (one quadrant pixel per thread) for the large geometries (tpspp_warp_span.h)
    return (size_t)K * Ho * Wo + (size_t)NW * QP * KG * kWave * 4 + (size_t)tpspp::span_table_waves(Ho, Wo) * KG * kWave * 4;
}

TPSPP_EXPORT int tpspp_prepare_mirror_table(const float* p_hat, int p_hat_ld, int Ho, int Wo, int F,
                                            float* prepared, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(p_hat && prepared, "tpspp_prepare_mirror_table: null pointer");
    TPSPP_REQUIRE(tpspp_prepared_table_floats(Ho, Wo, F) != 0,
                  "tpspp_prepare_mirror_table: needs Ho %% 16 == 0, Wo %% 4 == 0 (whole 32-pixel blocks of quadrant "
                  "pixels), 0 < F <= %d", kMaxK - 3);
    TPSPP_REQUIRE(p_hat_ld >= F + 3, "tpspp_prepare_mirror_table: p_hat_ld too small (classic layout: F + 3 columns)");
    const int K = F + 3, KG = (K + 3) / 4, n = Ho * Wo;
    const int rc = tpspp_transpose_p_hat(p_hat, p_hat_ld, n, K, prepared, stream);
    if (rc != TPSPP_OK) return rc;
    const int QP = img_qp(Ho, Wo);
    const int BW = tpspp_img::img_block_w(Wo);
    const int CG = ((Wo / 2) + BW - 1) / BW, nthr = img_nthr(Ho, Wo, QP), NW = (nthr + kWave - 1) / kWave;
    const int total = NW * QP * KG * kWave * 4;
    hipLaunchKernelGGL(tpspp_img::pack_img_table_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       tpspp::as_stream(stream), p_hat, p_hat_ld, Wo, CG, QP, BW, nthr, K, prepared + (size_t)K * n);
    const int sw = tpspp::span_table_waves(Ho, Wo);
    if (sw > 0) {                                            // third section: QP = 1, every quadrant pixel
        const int nthr1 = img_nthr(Ho, Wo, 1), total1 = sw * KG * kWave * 4;
        hipLaunchKernelGGL(tpspp_img::pack_img_table_kernel, dim3((total1 + 255) / 256), dim3(256), 0,
                           tpspp::as_stream(stream), p_hat, p_hat_ld, Wo, CG, 1, BW, nthr1, K,
                           prepared + (size_t)K * n + (size_t)total);
    }
    return tpspp::check_launch("tpspp_prepare_mirror_table");
}

namespace {

template <int F, int C, int HC, int WC>
bool launch_lds_geo(const LdsParams& P, int ppt, int threads, size_t lds, hipStream_t st)
{
    const dim3 grid((unsigned)(((P.N + 1) / 2) * P.bands)), block(threads);
    // > 64 KB of dynamic LDS needs the opt-in, once per instantiation
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_lds_kernel<F, C, 1, HC, WC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_lds_kernel<F, C, 2, HC, WC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_lds_kernel<F, C, 3, HC, WC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_lds_kernel<F, C, 4, HC, WC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    switch (ppt) {
    case 1: hipLaunchKernelGGL((tps_warp_lds_kernel<F, C, 1, HC, WC>), grid, block, lds, st, P); return true;
    case 2: hipLaunchKernelGGL((tps_warp_lds_kernel<F, C, 2, HC, WC>), grid, block, lds, st, P); return true;
    case 3: hipLaunchKernelGGL((tps_warp_lds_kernel<F, C, 3, HC, WC>), grid, block, lds, st, P); return true;
    case 4: hipLaunchKernelGGL((tps_warp_lds_kernel<F, C, 4, HC, WC>), grid, block, lds, st, P); return true;
    default: return false;
    }
}

template <int F, int C, int HC, int WC, bool AUX>
void launch_lds_mirror_aux(const LdsParams& P, int threads, size_t lds, hipStream_t st)
{
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_lds_mirror_kernel<F, C, HC, WC, AUX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const dim3 grid((unsigned)(((P.N + 1) / 2) * P.bands)), block(threads);
    hipLaunchKernelGGL((tps_warp_lds_mirror_kernel<F, C, HC, WC, AUX>), grid, block, lds, st, P);
}

template <int F, int C, int HC, int WC>
void launch_lds_mirror_geo(const LdsParams& P, int threads, size_t lds, hipStream_t st)
{
    // the optional grid / tap-index outputs get their own instantiation: the common call wants neither
    if (P.grid || P.idx) launch_lds_mirror_aux<F, C, HC, WC, true>(P, threads, lds, st);
    else launch_lds_mirror_aux<F, C, HC, WC, false>(P, threads, lds, st);
}

template <int F, int C>
void launch_lds_mirror(const LdsParams& P, int threads, size_t lds, hipStream_t st)
{
    if (P.H == 32 && P.W == 100) launch_lds_mirror_geo<F, C, 32, 100>(P, threads, lds, st);
    else launch_lds_mirror_geo<F, C, 0, 0>(P, threads, lds, st);
}

template <int F, int C>
bool launch_lds(const LdsParams& P, int ppt, int threads, size_t lds, hipStream_t st)
{
    // the reference's own geometry (tps_preprocessor.py:39-43, crnn_tps.py:7-12) gets constants
    if (P.H == 32 && P.W == 100) return launch_lds_geo<F, C, 32, 100>(P, ppt, threads, lds, st);
    return launch_lds_geo<F, C, 0, 0>(P, ppt, threads, lds, st);
}


template <int C>
void launch_pair(const float* in, const float* ctrl, const float* inv_delta_c, const float* packed,
                 const float* p_hat, int p_hat_ld, int N, float* out, float* grid, int32_t* idx, hipStream_t st)
{
    using namespace tpspp_pair;
    PairParams P;
    P.in = in; P.ctrl = ctrl; P.inv_delta_c = inv_delta_c; P.packed = packed; P.N = N;
    P.p_hat = p_hat; P.p_hat_ld = p_hat_ld;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = g_trace;
    const size_t lds = pair_lds_bytes<20, C, 32, 100, 32, 100>(&P.zero_off, &P.out_off);
    const dim3 grid_dim((unsigned)((N + 1) / 2)), block((PairGeo<32, 100>::NW + kPairLoaders) * kWave);
    // > 64 KB of dynamic LDS needs the opt-in, once per device, for every instantiation that may be launched from here
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_pair_kernel<20, C, 32, 100, 32, 100, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_pair_kernel<20, C, 32, 100, 32, 100, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tps_warp_pair_kernel<20, C, 32, 100, 32, 100, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid_dim, block, lds, st, P); };
    if (g_trace) go(tps_warp_pair_kernel<20, C, 32, 100, 32, 100, false, true>);
    else if (grid || idx) go(tps_warp_pair_kernel<20, C, 32, 100, 32, 100, true, false>);
    else go(tps_warp_pair_kernel<20, C, 32, 100, 32, 100, false, false>);
}

// In-place kernel (tpspp_warp_img.h): the geometries it is instantiated for.  IMGS / QP / loaders per geometry:
// two images per workgroup where both fit the LDS beside each other, else one.
template <int C, int HH, int WW, int IMGS, int QP, int NLOAD, int WPC = 1>
void launch_img(const float* in, const float* ctrl, const float* inv_delta_c, const float* packed, int N,
                float* out, float* grid, int32_t* idx, hipStream_t st)
{
    using namespace tpspp_img;
    static_assert(ImgGeo<HH, WW, QP>::NW + NLOAD <= 16, "too many wavefronts");
    ImgParams P;
    P.in = in; P.ctrl = ctrl; P.inv_delta_c = inv_delta_c; P.packed = packed; P.N = N;
    P.out = out; P.grid = grid; P.idx = idx; P.late_from = 1 << 30; P.trace = g_trace;
    const size_t lds = ImgLds<20, C, HH, WW, HH, WW, IMGS>::bytes;
    static_assert(ImgLds<20, C, HH, WW, HH, WW, IMGS>::bytes <= 160 * 1024, "does not fit the LDS");
    const dim3 grid_dim((unsigned)((N + IMGS - 1) / IMGS)), block((ImgGeo<HH, WW, QP>::NW + NLOAD) * kWave);
    auto k_plain = tps_warp_img_kernel<20, C, HH, WW, HH, WW, IMGS, QP, NLOAD, WPC, false, false>;
    auto k_aux = tps_warp_img_kernel<20, C, HH, WW, HH, WW, IMGS, QP, NLOAD, WPC, true, false>;
    auto k_trace = tps_warp_img_kernel<20, C, HH, WW, HH, WW, IMGS, QP, NLOAD, WPC, false, true>;
    // > 64 KB of dynamic LDS needs the opt-in, once per instantiation and device
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_aux), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trace), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    if (g_trace) hipLaunchKernelGGL(k_trace, grid_dim, block, lds, st, P);
    else if (grid || idx) hipLaunchKernelGGL(k_aux, grid_dim, block, lds, st, P);
    else hipLaunchKernelGGL(k_plain, grid_dim, block, lds, st, P);
}

// true when (C, H, W) is one of the in-place kernel's geometries (and the launch was enqueued)
bool launch_img_geo(int C, int H, int W, const float* in, const float* ctrl, const float* inv_delta_c, const float* packed,
                    int N, float* out, float* grid, int32_t* idx, hipStream_t st)
{
#define TPSPP_IMG_GEO(CC, HH, WW, IMGS, QP, NLOAD, WPC) \
    if (C == CC && H == HH && W == WW) { launch_img<CC, HH, WW, IMGS, QP, NLOAD, WPC>(in, ctrl, inv_delta_c, packed, N, out, grid, idx, st); return true; }
    // (IMGS, QP, loaders, workgroups per CU): an image pair per workgroup where 13 compute wavefronts cover a quadrant with
    // one pixel per thread (32x100, 32x64); else one image per workgroup, two pixels per thread and two workgroups per CU
    // (32x128: the pair form would leave 8 compute wavefronts alone on a CU: 21 us per 512 images against 13)
    TPSPP_IMG_GEO(3, 32, 100, 2, 1, 3, 1)      // also the image-pair kernel's geometry (kernel_choice 6 selects this one)
    TPSPP_IMG_GEO(1, 32, 100, 2, 1, 3, 1)
    TPSPP_IMG_GEO(3, 32, 128, 1, 2, 1, 2)      // configs/textrecog/nrtr/nrtr_tps++.py:28-33
    TPSPP_IMG_GEO(1, 32, 128, 1, 2, 1, 2)
    TPSPP_IMG_GEO(3, 48, 160, 1, 3, 3, 1)
    TPSPP_IMG_GEO(1, 48, 160, 1, 3, 3, 1)
    // the reference's recog-config test shape (tests/test_models/test_recog_config.py:103-157): an image pair per
    // workgroup, two quadrant pixels per thread (12 + 3 wavefronts, 123 KB): 15.1 us per 512 images on one stream against
    // 17.9 with one image per workgroup and 18.7 with two such workgroups per CU (scripts/debug/bench_32x160_variants.py)
    TPSPP_IMG_GEO(3, 32, 160, 2, 2, 3, 1)
    TPSPP_IMG_GEO(1, 32, 160, 1, 2, 1, 2)
    TPSPP_IMG_GEO(3, 32, 64, 2, 1, 3, 1)
    TPSPP_IMG_GEO(1, 32, 64, 2, 1, 1, 1)
#undef TPSPP_IMG_GEO
    return false;
}

}  // namespace

TPSPP_EXPORT int tpspp_warp_fwd(const float* in0, int C0, int H0, int W0,
                                const float* in1, int C1, int H1, int W1,
                                const float* ctrl, const float* score,
                                const float* inv_delta_c, const float* p_hat, int p_hat_ld,
                                const float* p_xy, const float* p_hat_t, int table_flags,
                                int N, int F, int Ho, int Wo,
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
    hipStream_t st = tpspp::as_stream(stream);
    if (table_flags & TPSPP_IO_BF16) {
        // bf16 images in and out (the bf16 configuration): plane-streaming kernel only
        tpspp::StreamArgs A;
        A.in0 = in0; A.C0 = C0; A.H0 = H0; A.W0 = W0;
        A.in1 = in1; A.C1 = C1; A.H1 = H1; A.W1 = W1;
        A.ctrl = ctrl; A.score = score; A.inv_delta_c = inv_delta_c;
        A.score_t = (score && (table_flags & TPSPP_SCORE_TRANSPOSED)) ? 1 : 0;
        A.io_bf16 = 1;
        A.p_hat = p_hat; A.p_hat_ld = p_hat_ld; A.p_xy = p_xy; A.p_hat_t = p_hat_t;
        A.N = N; A.F = F; A.Ho = Ho; A.Wo = Wo;
        A.out0 = out0; A.out1 = out1; A.grid = grid_or_null; A.idx = idx_or_null;
        if (!tpspp::stream_kernel_applicable(A))
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: TPSPP_IO_BF16 needs a shape the plane-streaming kernel "
                               "takes (F = 20 or 32, <= 1024 output pixels, planes that fit the LDS ring)");
        return tpspp::launch_stream_kernel(A, g_trace, st);
    }

    // ---- prepared (packed) mirror-symmetric table: image-pair kernel for the reference's own 32x100 geometry, the
    // in-place kernel for the other geometries it is instantiated for ----
    {
        const bool packed_ok = (table_flags & TPSPP_TABLE_MIRROR4) && (table_flags & TPSPP_TABLE_PACKED) && p_hat_t &&
                               !in1 && !score && !p_xy && F == 20 && H0 == Ho && W0 == Wo &&
                               (reinterpret_cast<uintptr_t>(in0) % 16 == 0) && (reinterpret_cast<uintptr_t>(out0) % 16 == 0);
        const bool pair_ok = packed_ok && (C0 == 1 || C0 == 3) && Ho == 32 && Wo == 100;
        const float* packed = p_hat_t ? p_hat_t + (size_t)(F + 3) * Ho * Wo : nullptr;   // second part of the prepared table
        if (g_tune_kernel == 5 && !pair_ok)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape / table do not qualify for the image-pair kernel");
        if (pair_ok && (g_tune_kernel == 0 || g_tune_kernel == 5)) {
            if (C0 == 1) launch_pair<1>(in0, ctrl, inv_delta_c, packed, p_hat, p_hat_ld, N, out0, grid_or_null, idx_or_null, st);
            else         launch_pair<3>(in0, ctrl, inv_delta_c, packed, p_hat, p_hat_ld, N, out0, grid_or_null, idx_or_null, st);
            return tpspp::check_launch("tpspp_warp_fwd(pair)");
        }
        if (packed_ok && (g_tune_kernel == 0 || g_tune_kernel == 6) &&
            launch_img_geo(C0, Ho, Wo, in0, ctrl, inv_delta_c, packed, N, out0, grid_or_null, idx_or_null, st))
            return tpspp::check_launch("tpspp_warp_fwd(in-place)");
        if (g_tune_kernel == 6)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape / table do not qualify for the in-place kernel");
        // every other geometry: the in-place kernel with run-time geometry where ONE workgroup covers the image
        // (tpspp_warp_geo.h), else row bands with span staging (tpspp_warp_span.h; round 5 -- the in-place kernel's banded
        // form, every band staging the whole image, is kept for comparisons: kernel_choice 9)
        const bool want_geo = g_tune_kernel == 0 || g_tune_kernel == 7;
        if (packed_ok && (g_tune_kernel == 9 || (want_geo && tpspp::geo_kernel_single_workgroup(C0, Ho, Wo, F))) &&
            tpspp::launch_geo_kernel(C0, Ho, Wo, F, in0, ctrl, inv_delta_c, packed, N, out0, grid_or_null, idx_or_null, st))
            return tpspp::check_launch("tpspp_warp_fwd(in-place, run-time geometry)");
        if (g_tune_kernel == 9)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape / table do not qualify for the run-time-geometry in-place kernel");
        // (the third section exists only in buffers of the current layout: TPSPP_TABLE_SPAN, include/tpspp.h)
        if (packed_ok && (table_flags & TPSPP_TABLE_SPAN) && (want_geo || g_tune_kernel == 8) && tpspp::span_table_waves(Ho, Wo) > 0) {
            const int QPg = tpspp::geo_qp(Ho, Wo);
            const int KGs = (F + 3 + 3) / 4;
            const int NWg = (tpspp::geo_nthr(Ho, Wo, QPg) + kWave - 1) / kWave;
            const float* span_packed = packed + (size_t)NWg * QPg * KGs * kWave * 4;      // third section of the prepared table
            if (tpspp::launch_span_kernel(C0, Ho, Wo, F, in0, ctrl, inv_delta_c, span_packed, N, out0, grid_or_null, idx_or_null, st))
                return tpspp::check_launch("tpspp_warp_fwd(row bands, span staging)");
        }
        if (g_tune_kernel == 8)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape / table do not qualify for the span-staging kernel");
        // (a geometry the span kernel does not take but whose image fits the LDS: the banded in-place kernel)
        if (packed_ok && want_geo &&
            tpspp::launch_geo_kernel(C0, Ho, Wo, F, in0, ctrl, inv_delta_c, packed, N, out0, grid_or_null, idx_or_null, st))
            return tpspp::check_launch("tpspp_warp_fwd(in-place, run-time geometry, bands)");
        if (g_tune_kernel == 7)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape / table do not qualify for the run-time-geometry kernels");
    }

    // ---- LDS-staged kernel: single small input, classic layout, transposed table available ----
    {
        const int img_elems = C0 * H0 * W0;
        const int K = F + 3;
        const int n = Ho * Wo;
        const int npairs = (N + 1) / 2;
        // split the output pixels of a pair over `bands` workgroups while the pairs alone would
        // leave CUs with fewer than two resident groups (the batch-512 case: 256 pairs, 256 CUs)
        int bands = g_tune_bands > 0 ? g_tune_bands : (npairs < 512 ? 2 : 1);
        if (n < bands * 2 * kWave) bands = 1;
        const int band_px = (((n + bands - 1) / bands) + kWave - 1) / kWave * kWave;
        const int max_cw = kLdsMaxWaves - kLdsLoaders;
        const int ppt = (band_px + max_cw * kWave - 1) / (max_cw * kWave);
        const int cw = ppt > 0 ? (band_px + ppt * kWave - 1) / (ppt * kWave) : 0;   // compute wavefronts
        const int pieces = (2 * img_elems * 4 + 1023) / 1024;                      // 1-KB DMA pieces
        // LDS: T | whole pieces of the image pair + the 1-float over-read of the east tap
        const size_t lds = (size_t)(4 * K + ((K * K + 3) & ~3)) * sizeof(float) + (size_t)pieces * 1024 + 16;
        const bool shape_ok = p_hat_t && !in1 && !score && !p_xy && F == 20 && (C0 == 1 || C0 == 3) &&
                              (img_elems % 4 == 0) && lds <= 160 * 1024 && ppt >= 1 && ppt <= 4 &&
                              cw >= 2 && (reinterpret_cast<uintptr_t>(in0) % 16 == 0);
        if (g_tune_kernel == 2 && !shape_ok)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape does not qualify for the LDS kernel");
        if (shape_ok && g_tune_kernel != 1 && g_tune_kernel != 4) {
            LdsParams L;
            L.in = in0; L.C = C0; L.H = H0; L.W = W0;
            L.ctrl = ctrl; L.inv_delta_c = inv_delta_c; L.p_hat_t = p_hat_t;
            L.N = N; L.n = n; L.Ho = Ho; L.Wo = Wo;
            L.out = out0; L.grid = grid_or_null; L.idx = idx_or_null;
            L.compute_waves = cw; L.band_px = band_px; L.bands = bands; L.trace = g_trace;
            // mirror-symmetric table (verified by the caller): one thread per quadrant pixel
            if ((table_flags & TPSPP_TABLE_MIRROR4) && Ho % 2 == 0 && Wo % 2 == 0 && g_tune_mirror != 2) {
                const int nq = (Ho / 2) * (Wo / 2);
                int mb = g_tune_bands > 0 ? g_tune_bands : 1;
                if (nq < mb * 2 * kWave) mb = 1;
                int qpb = (((nq + mb - 1) / mb) + kWave - 1) / kWave * kWave;     // quadrant px per band
                while (qpb > max_cw * kWave) { ++mb; qpb = (((nq + mb - 1) / mb) + kWave - 1) / kWave * kWave; }
                const int mcw = qpb / kWave < 2 ? 2 : qpb / kWave;
                L.compute_waves = mcw; L.band_px = qpb; L.bands = mb;
                const int threads = (mcw + kLdsLoaders) * kWave;
                // zero words for out-of-image taps: one per channel plane, H*W apart, behind the DMA pieces
                L.zero_off = pieces * 256;
                const size_t mlds = (size_t)(4 * K + ((K * K + 3) & ~3)) * sizeof(float) + (size_t)pieces * 1024 +
                                    ((size_t)(C0 - 1) * H0 * W0 + 4) * sizeof(float);
                if (mlds > 160 * 1024) goto no_mirror;
                if (C0 == 1) launch_lds_mirror<20, 1>(L, threads, mlds, st);
                else         launch_lds_mirror<20, 3>(L, threads, mlds, st);
                return tpspp::check_launch("tpspp_warp_fwd(lds-mirror)");
            }
        no_mirror:
            L.compute_waves = cw; L.band_px = band_px; L.bands = bands; L.zero_off = 0;
            const int threads = (cw + kLdsLoaders) * kWave;
            const bool ok = (C0 == 1) ? launch_lds<20, 1>(L, ppt, threads, lds, st)
                                      : launch_lds<20, 3>(L, ppt, threads, lds, st);
            if (ok) return tpspp::check_launch("tpspp_warp_fwd(lds)");
        }
    }

    // ---- plane-streaming kernel: channel planes through an LDS ring (TPS_PP geometry) ----
    {
        tpspp::StreamArgs A;
        A.in0 = in0; A.C0 = C0; A.H0 = H0; A.W0 = W0;
        A.in1 = in1; A.C1 = C1; A.H1 = H1; A.W1 = W1;
        A.ctrl = ctrl; A.score = score; A.inv_delta_c = inv_delta_c;
        A.score_t = (score && (table_flags & TPSPP_SCORE_TRANSPOSED)) ? 1 : 0;
        A.io_bf16 = 0;
        A.p_hat = p_hat; A.p_hat_ld = p_hat_ld; A.p_xy = p_xy; A.p_hat_t = p_hat_t;
        A.N = N; A.F = F; A.Ho = Ho; A.Wo = Wo;
        A.out0 = out0; A.out1 = out1; A.grid = grid_or_null; A.idx = idx_or_null;
        const bool ok = tpspp::stream_kernel_applicable(A);
        if (g_tune_kernel == 4 && !ok)
            return tpspp::fail(TPSPP_EINVAL, "tpspp_warp_fwd: shape does not qualify for the streaming kernel");
        if (ok && g_tune_kernel != 1) return tpspp::launch_stream_kernel(A, g_trace, st);
    }

    WarpParams P;
    P.in0 = in0; P.C0 = C0; P.H0 = H0; P.W0 = W0;
    P.in1 = in1; P.C1 = in1 ? C1 : 0; P.H1 = in1 ? H1 : 1; P.W1 = in1 ? W1 : 1;
    P.ctrl = ctrl; P.score = score; P.inv_delta_c = inv_delta_c;
    P.score_t = (score && (table_flags & TPSPP_SCORE_TRANSPOSED)) ? 1 : 0;
    P.p_hat = p_hat; P.p_hat_ld = p_hat_ld; P.p_xy = p_xy; P.p_hat_t = p_hat_t;
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
    if (F == 20)      launch_warp<20>(P, grid, block, lds, st);
    else if (F == 32) launch_warp<32>(P, grid, block, lds, st);
    else              launch_warp<0>(P, grid, block, lds, st);
    return tpspp::check_launch("tpspp_warp_fwd");
}

// ---- prepared calls (round 6) -----------------------------------------------------------------------------------------------
// A serving loop that rectifies batch after batch into the same buffers pays, per launch, for handing 25 arguments through its
// language's foreign-function layer (ctypes: 1.6 us of a ~4.3 us launch call, against a device that needs ~8 us per 512-image
// batch -- and at the start of a burst the device waits for the host, not the other way round).  A plan keeps the arguments on
// this side: tpspp_warp_plan_run is tpspp_warp_fwd with the stored arguments (same checks, same dispatch, same kernel).
struct tpspp_warp_plan {
    const float* in0; int C0, H0, W0;
    const float* in1; int C1, H1, W1;
    const float* ctrl; const float* score; const float* inv_delta_c; const float* p_hat; int p_hat_ld;
    const float* p_xy; const float* p_hat_t; int table_flags, N, F, Ho, Wo;
    float* out0; float* out1; float* grid; int32_t* idx;
    tpspp_stream_t stream;
};

TPSPP_EXPORT int tpspp_warp_plan_create(const float* in0, int C0, int H0, int W0,
                                        const float* in1, int C1, int H1, int W1,
                                        const float* ctrl, const float* score,
                                        const float* inv_delta_c, const float* p_hat, int p_hat_ld,
                                        const float* p_xy, const float* p_hat_t, int table_flags,
                                        int N, int F, int Ho, int Wo,
                                        float* out0, float* out1, float* grid_or_null, int32_t* idx_or_null,
                                        tpspp_stream_t stream, tpspp_warp_plan_t** plan_out)
{
    TPSPP_REQUIRE(plan_out, "tpspp_warp_plan_create: plan_out is NULL");
    *plan_out = nullptr;
    TPSPP_REQUIRE(in0 && ctrl && inv_delta_c && p_hat && out0, "tpspp_warp_plan_create: null pointer");
    tpspp_warp_plan* p = new (std::nothrow) tpspp_warp_plan{in0, C0, H0, W0, in1, C1, H1, W1, ctrl, score, inv_delta_c, p_hat, p_hat_ld,
                                                             p_xy, p_hat_t, table_flags, N, F, Ho, Wo, out0, out1, grid_or_null,
                                                             idx_or_null, stream};
    TPSPP_REQUIRE(p, "tpspp_warp_plan_create: out of host memory");
    *plan_out = p;
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_warp_plan_run(const tpspp_warp_plan_t* p)
{
    TPSPP_REQUIRE(p, "tpspp_warp_plan_run: NULL plan");
    return tpspp_warp_fwd(p->in0, p->C0, p->H0, p->W0, p->in1, p->C1, p->H1, p->W1, p->ctrl, p->score, p->inv_delta_c, p->p_hat,
                          p->p_hat_ld, p->p_xy, p->p_hat_t, p->table_flags, p->N, p->F, p->Ho, p->Wo, p->out0, p->out1, p->grid, p->idx,
                          p->stream);
}

TPSPP_EXPORT int tpspp_warp_plan_run_on(const tpspp_warp_plan_t* p, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(p, "tpspp_warp_plan_run_on: NULL plan");
    return tpspp_warp_fwd(p->in0, p->C0, p->H0, p->W0, p->in1, p->C1, p->H1, p->W1, p->ctrl, p->score, p->inv_delta_c, p->p_hat,
                          p->p_hat_ld, p->p_xy, p->p_hat_t, p->table_flags, p->N, p->F, p->Ho, p->Wo, p->out0, p->out1, p->grid, p->idx,
                          stream);
}

TPSPP_EXPORT void tpspp_warp_plan_destroy(tpspp_warp_plan_t* p)
{
    delete p;
}
