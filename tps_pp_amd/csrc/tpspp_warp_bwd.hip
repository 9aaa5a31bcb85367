// Backward of the fused TPS warp: gradients w.r.t. the sampled inputs,
// the control points and the attention score, so the rectifier can sit inside a training graph.
//
// Forward (tpspp_warp.hip):  T = inv_delta_C [C'; 0];  row(p) = [1, P.x, P.y, rbf_k (0.5 s_k + 1)];
//                            grid(p) = row(p) T;  out_i = grid_sample(in_i, grid)  (bilinear, border,
//                            align_corners=True), i = 0 (feature map) and optionally 1 (image).
// Backward of the classic rectifier (one input of <= 3 channels, no score, 1024 < pixels <= 4096): ONE launch,
// warp_bwd_classic_kernel below (round 5; 85 -> 39 us per 512 images).  Everything else, two kernels:
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

#include <type_traits>

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

// ---- kernel A: sampler backward.  Thread = one output pixel x kCPT channels of one input ---------------
// grid = (pixel blocks, channel chunks of input 0 then input 1, images).  dL/d input goes out as float
// atomics (fire-and-forget), the coordinate gradient of the chunk is added to g_grid[b][p] (zeroed by
// the host side), so the parallelism is N * n * C / kCPT threads instead of N * n.  unsafeAtomicAdd = the
// hardware's return-less global_atomic_add_f32 (valid on ordinary device allocations; plain atomicAdd on a
// float compiles to a compare-and-swap loop here, 10x slower).
constexpr int kCPT = 8;
constexpr int kBwdMaxG = 8;      // sampling workgroups per (image, input) of kernel A'': bounds the workspace

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

// ---- kernel A'' (round 3): input-gradient planes accumulated in LDS as 64-bit FIXED POINT ------------------
// What bounds A' is one instruction: ds_add_f32 retires 0.33 lane-operations per clock and CU on gfx950 whatever the
// address pattern (scripts/ubench/lds_atomic_bench.hip: 204 G/s chip-wide; the TPS_PP geometry needs 268 M of them per
// 512 images = 1.3 ms), while the INTEGER LDS atomics run at 8 (ds_add_u64) to 11 (ds_add_u32) per clock and CU.  So a
// contribution w * g is added as round(w * g * 2^s) with ds_add_u64; s is chosen per pass from max |g| of the planes
// in flight so that a term has 50 significant bits and 4096 terms cannot overflow (2^62).  The sum of the rounded terms
// is exact and order-independent: the result is the correctly rounded fp32 of a sum that is closer to the true value
// than any fp32 accumulation order (each term is off by <= 2^-51 of the largest |g|), and it is bitwise reproducible
// from run to run, which float atomics are not.  (fp32 -> fixed point: one fp64 fma against 1.5 * 2^52 and a 64-bit
// subtraction.)
// Besides:
//   * a workgroup is 256 threads with <= 64 KB of planes: two share a CU in different phases;
//   * a thread owns PPT fixed output pixels: their taps (one offset, two fractions, two flags, the two border-clip
//     factors) are derived ONCE per workgroup, not once per channel chunk;
//   * its coordinate gradient is summed in registers over every channel the workgroup handles and leaves as ONE plain
//     store into the workgroup's own slice of the workspace (kernel B adds the slices): no LDS array, no global atomics,
//     no memset of g_grid;
//   * two channel planes per pass share the tap addresses and weights (40 loads per thread in flight).
// grid = (G groups of input 0 then G groups of input 1, images); workspace (N, 2 G, n, 2).
// NT threads: 256 (two workgroups per CU; up to 1024 output pixels) or 1024 (the classic 32x100 geometry: 3200 pixels)
// F64 (round 4, the default): the accumulators are doubles and a term is added with ds_add_f64 (3.1 lane-operations per
// clock and CU in scripts/ubench/lds_atomic_bench.hip: 9x ds_add_f32, 0.4-0.6x ds_add_u64).  No scale, so a plane with
// gradients of very different magnitudes keeps every term's fp32 bits (the fixed-point form drops what lies 2^-50 below
// the pass's largest |g|), and a non-finite incoming gradient poisons exactly the four taps it touches, as ATen's
// kernel does.  The sum of fp32 terms in fp64 is exact up to 2^-53 relative per addition: the fp32 result is the
// correctly rounded sum except for ties broken by the order of arrival.
template <int PPT, int NT, bool F64>
__global__ void __launch_bounds__(NT, NT == 256 ? 2 : 1)
warp_bwd_sample_lds2_kernel(const BwdParams P, int G, int cpt0, int cpt1, float* __restrict__ g_grid_part,
                            int* __restrict__ arrivals)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];   // [cpt][H*W] fixed point
    __shared__ float sMax[2][NT / 64];
    const int b = blockIdx.y;
    const int i = (int)blockIdx.x >= G ? 1 : 0;
    const int grp = (int)blockIdx.x - (i ? G : 0);
    const int cpt = i ? cpt1 : cpt0;
    const int H = P.H[i], W = P.W[i], C = P.C[i];
    const int plane = H * W;
    const int chunks = (C + cpt - 1) / cpt;
    const bool want = P.g_in[i] != nullptr;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) arrivals[b] = 0;      // the parameter kernel's arrival counter of this image (next launch)

    // this thread's pixels: taps once
    int o00[PPT]; float fw[PPT], fn[PPT], mx[PPT], my[PPT]; bool inx[PPT], iny[PPT], livep[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        livep[k] = p < P.n;
        const float2 g = reinterpret_cast<const float2*>(P.grid)[(size_t)b * P.n + (livep[k] ? p : 0)];
        float ix = ((g.x + 1.0f) * 0.5f) * (float)(W - 1);
        float iy = ((g.y + 1.0f) * 0.5f) * (float)(H - 1);
        mx[k] = (float)(W - 1) * 0.5f; my[k] = (float)(H - 1) * 0.5f;
        if (ix <= 0.0f) { ix = 0.0f; mx[k] = 0.0f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); mx[k] = 0.0f; }
        if (iy <= 0.0f) { iy = 0.0f; my[k] = 0.0f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); my[k] = 0.0f; }
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        fw[k] = ix - fx; fn[k] = iy - fy;
        inx[k] = (x0 + 1) < W; iny[k] = (y0 + 1) < H;
        o00[k] = y0 * W + x0;
    }
    float gx[PPT], gy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) gx[k] = gy[k] = 0.0f;

    const int n2 = plane;                                  // the two planes of a pass as 16-byte pieces (two fixed-point words each)
    ulonglong2* acc2 = reinterpret_cast<ulonglong2*>(acc);
    if (want) for (int e = tid; e < n2; e += NT) acc2[e] = make_ulonglong2(0ull, 0ull);
    __syncthreads();
    constexpr double kMagic = 6755399441055744.0;          // 1.5 * 2^52: x + kMagic has round(x) in its low mantissa bits
    for (int ch = grp; ch < chunks; ch += G) {
        const int c_lo = ch * cpt;
        const int nc = min(cpt, C - c_lo);
        const float* in = P.in[i] + ((size_t)b * C + c_lo) * plane;
        const float* go = P.g_out[i] + ((size_t)b * C + c_lo) * P.n;
        float* gi = want ? P.g_in[i] + ((size_t)b * C + c_lo) * plane : nullptr;
        for (int c2 = 0; c2 < nc; c2 += 2) {               // two planes per pass: same taps, same weights
            const bool two = c2 + 1 < nc;
            const float* pl0 = in + (size_t)c2 * plane;
            const float* pl1 = in + (size_t)(two ? c2 + 1 : c2) * plane;
            float gv[PPT][2], v[PPT][2][4];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {                // every load of the pass in flight together
                const int p = livep[k] ? tid + k * NT : 0;
                const int a01 = inx[k] ? o00[k] + 1 : o00[k], a10 = iny[k] ? o00[k] + W : o00[k];
                const int a11 = (inx[k] && iny[k]) ? o00[k] + W + 1 : o00[k];
                gv[k][0] = go[(size_t)c2 * P.n + p];
                gv[k][1] = go[(size_t)(two ? c2 + 1 : c2) * P.n + p];
                v[k][0][0] = pl0[o00[k]]; v[k][0][1] = pl0[a01]; v[k][0][2] = pl0[a10]; v[k][0][3] = pl0[a11];
                v[k][1][0] = pl1[o00[k]]; v[k][1][1] = pl1[a01]; v[k][1][2] = pl1[a10]; v[k][1][3] = pl1[a11];
            }
            // fixed-point scale of this pass: 2^(50 - e) with 2^e > max |g| over the two planes (a term then has 50
            // significant bits, 4096 terms stay below 2^62)
            double scale = 0.0, inv_scale = 0.0;
            bool fin[2] = {true, true};                    // a plane whose incoming gradient is not finite comes out as NaN
            if (want && !F64) {
                float m[2] = {0.0f, 0.0f};
                bool nan_[2] = {false, false};
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    if (livep[k]) {
                        m[0] = fmaxf(m[0], fabsf(gv[k][0])); nan_[0] = nan_[0] || (gv[k][0] != gv[k][0]);
                        if (two) { m[1] = fmaxf(m[1], fabsf(gv[k][1])); nan_[1] = nan_[1] || (gv[k][1] != gv[k][1]); }
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) m[q] = fmaxf(m[q], __shfl_xor(m[q], o, kWave));
                    if (__builtin_amdgcn_ballot_w64(nan_[q]) != 0) m[q] = __builtin_inff();
                    if ((tid & (kWave - 1)) == 0) sMax[q][tid >> 6] = m[q];
                }
                __syncthreads();
                float mm = 0.0f;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    m[q] = sMax[q][0];
#pragma unroll
                    for (int w_ = 1; w_ < NT / 64; ++w_) m[q] = fmaxf(m[q], sMax[q][w_]);
                    fin[q] = m[q] < 3.0e38f;
                    if (fin[q]) mm = fmaxf(mm, m[q]);
                }
                if (mm > 0.0f) {
                    int e2;
                    (void)frexpf(mm, &e2);                 // mm = f * 2^e2, f in [0.5, 1)
                    scale = ldexp(1.0, 50 - e2);
                    inv_scale = ldexp(1.0, e2 - 50);
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (!livep[k]) continue;
                const float w = fw[k], nn = fn[k], e = 1.0f - w, s = 1.0f - nn;
                const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
                const bool inxy = inx[k] && iny[k];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && !two) break;
                    const float g = gv[k][q];
                    const float a01 = inx[k] ? v[k][q][1] : 0.0f, a10 = iny[k] ? v[k][q][2] : 0.0f, a11 = inxy ? v[k][q][3] : 0.0f;
                    gx[k] += ((a01 - v[k][q][0]) * s + (a11 - a10) * nn) * g;
                    gy[k] += ((a10 - v[k][q][0]) * e + (a11 - a01) * w) * g;
                    if constexpr (F64) {
                        if (want) {
                            double* ap = reinterpret_cast<double*>(acc) + (size_t)q * plane + o00[k];
                            unsafeAtomicAdd(ap, (double)(nw * g));
                            if (inx[k]) unsafeAtomicAdd(ap + 1, (double)(ne * g));
                            if (iny[k]) unsafeAtomicAdd(ap + W, (double)(sw * g));
                            if (inxy) unsafeAtomicAdd(ap + W + 1, (double)(se * g));
                        }
                    } else if (want && scale != 0.0 && fin[q]) {
                        unsigned long long* ap = acc + (size_t)q * plane + o00[k];
                        auto fx64 = [&](float t) {
                            return (unsigned long long)(__double_as_longlong(fma((double)t, scale, kMagic)) - __double_as_longlong(kMagic));
                        };
                        atomicAdd(ap, fx64(nw * g));
                        if (inx[k]) atomicAdd(ap + 1, fx64(ne * g));
                        if (iny[k]) atomicAdd(ap + W, fx64(sw * g));
                        if (inxy) atomicAdd(ap + W + 1, fx64(se * g));
                    }
                }
            }
            if (want) {
                __syncthreads();
                // the pass's planes out (fixed point -> fp32, 16-byte stores) and zeroed for the next pass: the same
                // thread reads and clears a piece
                const int m4 = ((two ? 2 : 1) * plane) >> 2;
                float4* gi4 = reinterpret_cast<float4*>(gi + (size_t)c2 * plane);
                const float nanv = __builtin_nanf("");
                for (int e = tid; e < m4; e += NT) {
                    const ulonglong2 r0 = acc2[2 * e], r1 = acc2[2 * e + 1];
                    acc2[2 * e] = make_ulonglong2(0ull, 0ull);
                    acc2[2 * e + 1] = make_ulonglong2(0ull, 0ull);
                    float4 o;
                    if constexpr (F64) {
                        o.x = (float)__longlong_as_double((long long)r0.x); o.y = (float)__longlong_as_double((long long)r0.y);
                        o.z = (float)__longlong_as_double((long long)r1.x); o.w = (float)__longlong_as_double((long long)r1.y);
                    } else {
                    o.x = (float)((double)(long long)r0.x * inv_scale); o.y = (float)((double)(long long)r0.y * inv_scale);
                    o.z = (float)((double)(long long)r1.x * inv_scale); o.w = (float)((double)(long long)r1.y * inv_scale);
                    if (!fin[4 * e >= plane ? 1 : 0]) o = make_float4(nanv, nanv, nanv, nanv);   // (planes are whole float4s)
                    }
                    gi4[e] = o;
                }
                __syncthreads();
            }
        }
    }
    // (slice stride = the G * nin workgroups of this image: with one input no slice stays empty, nothing to zero)
    float2* part = reinterpret_cast<float2*>(g_grid_part) + ((size_t)b * gridDim.x + blockIdx.x) * P.n;
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (livep[k]) part[tid + k * NT] = make_float2(gx[k] * mx[k], gy[k] * my[k]);
}

// ---- kernel B: parameter gradients from g_grid.  B1: S workgroups per image, each over n / S pixels: dL/d score,
// dL/d grid (sum of the sampling workgroups' slices) and a partial dL/dT per workgroup; B2: one workgroup per image
// sums the S partials in a fixed order and applies inv_delta_C^T.  (Round 3 had one workgroup per image do all of it:
// 129 us of the 551 us backward at batch 512, latency-bound with 2 workgroups per CU.)
constexpr int kBwdMaxS = 8;

// TABLE: 0 = classic table [1, x, y, rbf] (K columns), 1 = TPS_PP (rbf only + p_xy);  TRANSPOSED: p_hat_t (cols, n) given;
// SCORE: 0 none, 1 (N, n, F), 2 transposed (N, F, n);  GSCORE: dL/d score wanted.
// Work split (round 4): a workgroup walks its pixels 256 at a time.  Phase 1, thread = pixel: dL/d grid = sum of the
// sampling workgroups' slices, stored and left in LDS.  Phase 2, wavefront w = table columns [w KPW, (w + 1) KPW),
// lane = pixel (4 x 64 per chunk): everything a (pixel, column group) needs is requested before the first use, the
// running dL/dT of the column group stays in 2 KPW registers.  (Round 3: thread = pixel over ALL columns -- a loop that
// waited for memory once per column, 129 us per 512 images against ~30 us of traffic; unrolled with the loads batched
// it needs 316 registers: one wavefront per SIMD.)  At the end a wavefront sums its own columns over its 64 lanes
// through LDS in fp64: dL/dC' = inv_delta_C^T dL/dT cancels heavily (|inv_delta_C| up to ~220), the order of an fp32
// reduction over thousands of pixels would show in the 4th digit of the result.
template <int KMAX, int TABLE, bool TRANSPOSED, int SCORE, bool GSCORE>
__global__ void __launch_bounds__(256)
warp_bwd_params_kernel(const BwdParams P, float* __restrict__ g_grid, const float* __restrict__ g_grid_part, int slots,
                       int S, double* __restrict__ gT_part, int* __restrict__ arrivals)
{
    constexpr int KPW = KMAX / 4;                              // table columns per wavefront
    __shared__ float sT[kMaxK * 2];
    __shared__ float2 sGG[256];
    __shared__ float sPart[4][2 * KPW][kWave + 1];
    const int b = blockIdx.x / S, sl_ = blockIdx.x - b * S;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
    const int K = P.F + 3, F = P.F, n = P.n;
    if (SCORE != 0) for (int i = tid; i < 2 * K; i += blockDim.x) sT[i] = P.T[(size_t)b * K * 2 + i];   // (only dL/d score reads T)

    float aT[KPW][2];
#pragma unroll
    for (int j = 0; j < KPW; ++j) aT[j][0] = aT[j][1] = 0.0f;

    const int per = (n + S - 1) / S;
    const int p_lo = sl_ * per, p_hi = min(n, p_lo + per);
    for (int p0 = p_lo; p0 < p_hi; p0 += 256) {
        __syncthreads();                                       // sT written / the previous chunk's sGG consumed
        {   // ---- phase 1: dL/d grid of pixel p0 + tid (the slices in ascending order) ----
            const int p = p0 + tid;
            float2 gg = make_float2(0.0f, 0.0f);
            if (p < p_hi) {
                if (slots > 0) {
                    float2 part[2 * kBwdMaxG];
                    const float2* ps = reinterpret_cast<const float2*>(g_grid_part) + (size_t)b * slots * n + p;
#pragma unroll
                    for (int sl = 0; sl < 2 * kBwdMaxG; ++sl) part[sl] = sl < slots ? ps[(size_t)sl * n] : make_float2(0.0f, 0.0f);
                    gg = part[0];
#pragma unroll
                    for (int sl = 1; sl < 2 * kBwdMaxG; ++sl) if (sl < slots) { gg.x += part[sl].x; gg.y += part[sl].y; }
                    reinterpret_cast<float2*>(g_grid)[(size_t)b * n + p] = gg;      // dL/d grid, as documented
                } else {
                    gg = reinterpret_cast<const float2*>(g_grid)[(size_t)b * n + p];
                }
            }
            sGG[tid] = gg;
        }
        __syncthreads();
        // ---- phase 2: this wavefront's columns of the chunk's pixels ----
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int p = p0 + it * kWave + lane;
            if (p0 + it * kWave >= p_hi) break;
            const bool livep = p < p_hi;
            const int pc = livep ? p : p_hi - 1;
            float v[KPW], sc[KPW];
            float2 pxy = make_float2(0.0f, 0.0f);
            if (TABLE == 1 && wv == 0) pxy = reinterpret_cast<const float2*>(P.p_xy)[pc];
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                const int k = wv * KPW + j;
                const int col = TABLE == 1 ? k - 3 : k;        // column of the table that holds row(p)[k]
                v[j] = 0.0f; sc[j] = 0.0f;
                if (k < K && col >= 0)
                    v[j] = TRANSPOSED ? P.p_hat_t[(size_t)col * n + pc] : P.p_hat[(size_t)pc * P.p_hat_ld + col];
                if (SCORE != 0 && k >= 3 && k < K)
                    sc[j] = P.score[SCORE == 2 ? ((size_t)b * F + (k - 3)) * n + pc : ((size_t)b * n + pc) * F + (k - 3)];
            }
            if (TABLE == 1 && wv == 0) { v[0] = 1.0f; v[1] = pxy.x; v[2] = pxy.y; }
            const float2 gg = livep ? sGG[it * kWave + lane] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                const int k = wv * KPW + j;
                if (k < K) {
                    float vv = v[j];
                    if (SCORE != 0 && k >= 3) {
                        if (GSCORE && livep) {
                            const size_t so = SCORE == 2 ? ((size_t)b * F + (k - 3)) * n + p : ((size_t)b * n + p) * F + (k - 3);
                            P.g_score[so] = 0.5f * vv * (gg.x * sT[2 * k] + gg.y * sT[2 * k + 1]);
                        }
                        vv = vv * (0.5f * sc[j] + 1.0f);
                    }
                    aT[j][0] = fmaf(vv, gg.x, aT[j][0]);
                    aT[j][1] = fmaf(vv, gg.y, aT[j][1]);
                }
            }
        }
    }
    // ---- the wavefront's columns: 64 partials each, added in lane order in fp64 ----
#pragma unroll
    for (int j = 0; j < KPW; ++j) { sPart[wv][2 * j][lane] = aT[j][0]; sPart[wv][2 * j + 1][lane] = aT[j][1]; }
    __syncthreads();
    if (lane < 2 * KPW) {
        const int k = wv * KPW + (lane >> 1);
        if (k < K) {
            double a = 0.0;
            for (int j = 0; j < kWave; ++j) a += (double)sPart[wv][lane][j];
            // write-through (sc0 sc1) store: the partial is read by another workgroup of this launch, maybe on another XCD
            __hip_atomic_store(gT_part + ((size_t)b * S + sl_) * (2 * kMaxK) + 2 * k + (lane & 1), a, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- B2 (round 5: was its own launch, warp_bwd_ctrl_kernel): the LAST of the image's S workgroups to arrive sums
    // the S partials in ascending order and applies inv_delta_C^T, in fp64, rounded once -- the same arithmetic in the
    // same order whoever is last.  Hand-off (MI355X_MICROARCH.md, "sc0 sc1 stores and loads both sides"): the 2 K partial
    // sums leave as write-through stores, drained (vmcnt(0)) before the barrier; one lane then takes a ticket with a
    // relaxed agent-scope atomic; the last arriver reads every partial with sc0 sc1 loads (they bypass its L1 and L2).
    // No release / acquire fence: a fence writes back / invalidates whole caches, which cost 15-20 us per backward when
    // every one of the 2048 workgroups issued one.  `arrivals` was zeroed by the launch before this one.
    __shared__ int sLast;
    __shared__ double sGT[kMaxK * 2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        sLast = S == 1 || __hip_atomic_fetch_add(arrivals + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
    __syncthreads();
    if (!sLast) return;
    for (int i = tid; i < 2 * K; i += blockDim.x) {
        double a = 0.0;
        for (int sl = 0; sl < S; ++sl)
            a += __hip_atomic_load(gT_part + ((size_t)b * S + sl) * (2 * kMaxK) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sGT[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < 2 * F; i += blockDim.x) {
        const int f = i >> 1, xy = i & 1;
        double a = 0.0;
        for (int k = 0; k < K; ++k) a = fma((double)P.inv_delta_c[(size_t)k * K + f], sGT[2 * k + xy], a);
        P.g_ctrl[(size_t)b * F * 2 + i] = (float)a;
    }
}


// ---- the classic rectifier's backward as ONE launch (round 5) --------------------------------------------------------------
// One input of C <= 3 channels, table [1, x, y, rbf] without a score, 1024 < n <= 4096 output pixels, all C input-gradient
// planes (fp64 accumulators) in <= 78 KB of LDS: the 3 x 32 x 100 geometry of BASELINE.json configs[1].  Rounds 3-5 ran it as
// kernel A'' (one 1024-thread workgroup per image, 128 registers -> ONE workgroup per CU, two rounds of a workgroup whose
// phases -- grid, gathers, LDS atomics, plane write-out, twice: two planes per pass -- wait for each other: 53 us per 512
// images) + kernel B (4 workgroups per image, partial dL/dT through memory, a ticket: 30 us).  Here
//   * the image's planes are staged in LDS (fp32, 38 KB) and the 12 taps of a pixel are LDS reads: gathering them from
//     global memory costs ~34 cycles of the texture-address unit per 64-lane instruction -- 17 us per 512 images; the
//     same space then becomes the fp64 accumulators of ALL planes (3 x 3200 x 8 B): phase A = dL/d grid, phase B = dL/d
//     input with the tap offsets and weights derived again from the 4 grid points the thread keeps (same expressions,
//     same bits);
//   * <= 64 registers, 78 KB of LDS: TWO 1024-thread workgroups per CU, one's atomics and write-out run under the other's
//     loads;
//   * dL/d grid stays in the workgroup (registers -> LDS, the planes' space after their write-out) and dL/dT is finished
//     here: wavefront w = table columns 3 (w & 7) .. + 2 over one half of the pixels (lane = pixel, an fp64 FMA chain per
//     lane in ascending pixel order), the 64 lanes and the two halves added in fp64 in a fixed order, then
//     dL/dC' = inv_delta_C^T dL/dT in fp64, rounded once (kernel B's arithmetic): no slices, no partials in memory, no
//     ticket, no second launch.
// The sampling arithmetic (taps, border-clip factors, weights, the order of the channels in the coordinate gradient, the
// fp64 LDS accumulation of dL/d input) is that of kernel A'', expression for expression: dL/d input and dL/d grid come out
// with the same bits; dL/dC' is the correctly rounded chain up to fp64 rounding (the two-kernel route keeps fp32 partial sums
// of ~12 terms per lane: ~1e-5 of the largest entry).
constexpr int kClassicNT = 1024;

template <int C>
__global__ void __launch_bounds__(kClassicNT, 8)
warp_bwd_classic_kernel(const BwdParams P, float* __restrict__ g_grid)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];   // see the phases below
    __shared__ double sPart[kClassicNT / kWave][6];
    __shared__ double sGT[24 * 2];
    constexpr int NT = kClassicNT, PPT = 4;
    const int b = blockIdx.x;
    const int H = P.H[0], W = P.W[0];
    const int plane = H * W, n = P.n;
    const bool want = P.g_in[0] != nullptr;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- requests: this thread's 4 grid points and incoming gradients, and the image's planes -> LDS ----
    const float* in = P.in[0] + (size_t)b * C * plane;
    const float* go = P.g_out[0] + (size_t)b * C * n;
    float* sin_ = reinterpret_cast<float*>(acc);               // phase A: [C][plane] fp32 (+ W + 1 words that masked taps may read)
    float2 g4[PPT];
    float gv[PPT][C];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        const int pc = p < n ? p : 0;
        g4[k] = reinterpret_cast<const float2*>(P.grid)[(size_t)b * n + pc];
#pragma unroll
        for (int c = 0; c < C; ++c)                            // (32-bit byte offsets from the image's uniform base)
            gv[k][c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(go) + 4u * (unsigned)(c * n + pc));
    }
    for (int e = tid; e < (C * plane) >> 2; e += NT)
        reinterpret_cast<float4*>(sin_)[e] = reinterpret_cast<const float4*>(in)[e];
    __syncthreads();

    // a pixel's taps: offset of the north-west tap + four flags in one word (east column / south row inside the image,
    // x / y clamped at the border), the two fractions: 3 registers per pixel across both phases
    struct Taps { int packed; float fw, fn; };
    auto taps = [&](float2 g) {
        Taps t;
        float ix = ((g.x + 1.0f) * 0.5f) * (float)(W - 1);
        float iy = ((g.y + 1.0f) * 0.5f) * (float)(H - 1);
        // (the border-clip factors: 0 where the coordinate was clamped, else (size - 1) / 2)
        bool clipx = false, clipy = false;
        if (ix <= 0.0f) { ix = 0.0f; clipx = true; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); clipx = true; }
        if (iy <= 0.0f) { iy = 0.0f; clipy = true; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); clipy = true; }
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        t.fw = ix - fx; t.fn = iy - fy;
        t.packed = (y0 * W + x0) | ((x0 + 1) < W ? 1 << 16 : 0) | ((y0 + 1) < H ? 1 << 17 : 0) | (clipx ? 1 << 18 : 0) | (clipy ? 1 << 19 : 0);
        return t;
    };

    // ---- phase A: dL/d grid from the staged taps ----
    Taps tp[PPT];                                              // (kept for phase B: offset + two fractions; the flags are lane masks)
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        tp[k] = taps(g4[k]);
        if (p < n) {
            const Taps t = tp[k];
            const float w = t.fw, nn = t.fn, e = 1.0f - w, s = 1.0f - nn;
            const int o00 = t.packed & 0xffff;
            const bool inx = t.packed & (1 << 16), iny = t.packed & (1 << 17), inxy = inx && iny;
            float gx = 0.0f, gy = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // (a masked tap is read anyway -- the word exists: next row, next plane or the pad -- and replaced by zero)
                const float* q = sin_ + c * plane + o00;
                const float v00 = q[0], r01 = q[1], r10 = q[W], r11 = q[W + 1];
                const float g = gv[k][c];
                const float a01 = inx ? r01 : 0.0f, a10 = iny ? r10 : 0.0f, a11 = inxy ? r11 : 0.0f;
                gx += ((a01 - v00) * s + (a11 - a10) * nn) * g;
                gy += ((a10 - v00) * e + (a11 - a01) * w) * g;
            }
            // the documented output; read back below by this same thread (8 registers less across phase B)
            reinterpret_cast<float2*>(g_grid)[(size_t)b * n + p] =
                make_float2(gx * ((t.packed & (1 << 18)) ? 0.0f : (float)(W - 1) * 0.5f), gy * ((t.packed & (1 << 19)) ? 0.0f : (float)(H - 1) * 0.5f));
        }
    }
    __syncthreads();                                           // every tap has been read: the space becomes the accumulators

    // ---- phase B: dL/d input, fp64 LDS accumulation ----
    ulonglong2* acc2 = reinterpret_cast<ulonglong2*>(acc);
    if (want) {
        for (int e = tid; e < (C * plane) >> 1; e += NT) acc2[e] = make_ulonglong2(0ull, 0ull);
        __syncthreads();
        // What bounds this kernel is the LDS pipeline (ds_add_f64 retires 3.1 lane-operations per clock and CU: 4 per tap set and
        // channel = 13 us per 512 images).  Neighbouring lanes are neighbouring output pixels, and for a smooth warp lane
        // l + 1's north-west tap IS lane l's north-east tap (south-west / south-east likewise): lane l hands its two east
        // contributions to lane l + 1 through DPP (wave_shr / wave_shl: no LDS traffic), which adds them to its own west
        // ones in fp64 -- the sum of two fp32 products, exact unless their exponents are > 29 apart -- and issues ONE atomic
        // per row: two per tap set instead of four wherever the pattern holds (every lane evaluates the same predicate
        // from both sides: neighbour alive, the giver's east column inside the image, offsets one apart).
        auto dpp_prev = [](int v) { return __builtin_amdgcn_update_dpp(-1, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); };
        auto dpp_next = [](int v) { return __builtin_amdgcn_update_dpp(-1, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); };
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = tid + k * NT;
            const bool livep = p < n;
            const Taps t = tp[k];
            const float w = t.fw, nn = t.fn, e = 1.0f - w, s = 1.0f - nn;
            const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
            const int pk = livep ? t.packed : -1;              // (a live pixel's word has bits 20.. clear: never -1)
            const int o00 = t.packed & 0xffff;
            const bool inx = t.packed & (1 << 16), iny = t.packed & (1 << 17), inxy = inx && iny;
            const int prev = dpp_prev(pk), next = dpp_next(pk);
            const bool recv = livep && prev != -1 && (prev & (1 << 16)) && (prev & 0xffff) + 1 == o00;
            const bool give = livep && next != -1 && inx && o00 + 1 == (next & 0xffff);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float g = gv[k][c];
                const float t00 = nw * g, t01 = ne * g, t10 = sw * g, t11 = se * g;
                const float p01 = __int_as_float(dpp_prev(__float_as_int(t01))), p11 = __int_as_float(dpp_prev(__float_as_int(t11)));
                double d00 = (double)t00, d10 = (double)t10;
                if (recv) { d00 += (double)p01; d10 += (double)p11; }     // (same row as the giver: its south row is inside iff ours is)
                if (livep) {
                    double* ap = reinterpret_cast<double*>(acc) + c * plane + o00;
                    unsafeAtomicAdd(ap, d00);
                    if (inx && !give) unsafeAtomicAdd(ap + 1, (double)t01);
                    if (iny) unsafeAtomicAdd(ap + W, d10);
                    if (inxy && !give) unsafeAtomicAdd(ap + W + 1, (double)t11);
                }
            }
        }
        __syncthreads();                                       // every contribution has landed
        // (one 16-byte LDS read = two accumulators per lane: consecutive lanes on consecutive banks; two reads per lane for a
        // 16-byte store put the lanes 32 bytes apart -- a two-way bank conflict on every read)
        float2* gi2 = reinterpret_cast<float2*>(P.g_in[0] + (size_t)b * C * plane);
        for (int e = tid; e < (C * plane) >> 1; e += NT) {
            const ulonglong2 r0 = acc2[e];
            gi2[e] = make_float2((float)__longlong_as_double((long long)r0.x), (float)__longlong_as_double((long long)r0.y));
        }
        __syncthreads();                                       // the planes' space is free
    }
    // dL/d grid into LDS for the column wavefronts below (each thread its own stores of phase A)
    // (x and y in planes of their own: the column wavefronts read 4 consecutive pixels per lane, 16 bytes from each plane)
    float* sgx = reinterpret_cast<float*>(acc);
    float* sgy = sgx + n;
    {
        unsigned long long raw[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = tid + k * NT;
            raw[k] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(g_grid) + (size_t)b * n + (p < n ? p : 0));
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = tid + k * NT;
            if (p < n) { sgx[p] = __uint_as_float((unsigned)raw[k]); sgy[p] = __uint_as_float((unsigned)(raw[k] >> 32)); }
        }
    }
    __syncthreads();

    // ---- dL/dT: wavefront = 3 columns of the transposed table x one half of the pixels; a lane takes 4 consecutive pixels
    // per step (16-byte table loads), an 8-term fp32 FMA chain per two loads, the chains added in fp64 (dL/dC' = inv_delta_C^T
    // dL/dT cancels heavily, |inv_delta_C| up to ~220: one 25-term fp32 chain per lane shows as 3e-5 of the largest entry;
    // fp64 from the first product on costs 22 us per 512 images on the half-rate fp64 vector ALU) ----
    const int K = P.F + 3, F = P.F;
    const int cgp = wv & 7, half = wv >> 3;
    constexpr int kStep = 4 * kWave, kIts = 8;                 // n <= 4096: a half is at most 8 steps of 256 pixels
    const int hn = (((n + 1) >> 1) + kStep - 1) / kStep * kStep;
    const int lo = half * hn, hi = min(n, lo + hn);           // (n is a multiple of 4: so is hi)
    const char* tabc = reinterpret_cast<const char*>(P.p_hat_t);
    double a[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) a[j][0] = a[j][1] = 0.0;
#pragma unroll
    for (int it = 0; it < kIts; it += 2) {                     // two steps per batch: 6 table loads in flight, dL/d grid read once
        if (lo + it * kStep >= hi) break;                      // (uniform)
        float4 tv[2][3], ga[2], gb[2];                         // (ga | gb: dL/d grid x | y of pixels p .. p + 3)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int p = lo + (it + h2) * kStep + 4 * lane;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = 3 * cgp + j;
                tv[h2][j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (p < hi && col < K) tv[h2][j] = *reinterpret_cast<const float4*>(tabc + 4u * ((unsigned)col * (unsigned)n + (unsigned)p));
            }
            ga[h2] = gb[h2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (p < hi) { ga[h2] = *reinterpret_cast<const float4*>(sgx + p); gb[h2] = *reinterpret_cast<const float4*>(sgy + p); }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {                          // one 8-term fp32 chain per column and batch, then fp64
            float cx = 0.0f, cy = 0.0f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const float4 t4 = tv[h2][j];
                cx = fmaf(t4.x, ga[h2].x, cx); cy = fmaf(t4.x, gb[h2].x, cy);
                cx = fmaf(t4.y, ga[h2].y, cx); cy = fmaf(t4.y, gb[h2].y, cy);
                cx = fmaf(t4.z, ga[h2].z, cx); cy = fmaf(t4.z, gb[h2].z, cy);
                cx = fmaf(t4.w, ga[h2].w, cx); cy = fmaf(t4.w, gb[h2].w, cy);
            }
            a[j][0] += (double)cx;
            a[j][1] += (double)cy;
        }
    }
    // the 64 lanes in fp64, a fixed order: inside the rows of 16 lanes on the DPP path (quad swaps, half-row and row mirrors:
    // every lane of a row ends with the row's sum), the four rows through v_readlane -- no LDS traffic (a __shfl_xor butterfly
    // of a double is 12 ds_bpermute: 72 per wavefront, half of this kernel's non-atomic LDS instructions)
    auto dpp_f64 = [](double v, auto ctrl) {
        constexpr int CT = decltype(ctrl)::value;
        const long long r = __double_as_longlong(v);
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)r, CT, 0xf, 0xf, false);
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(r >> 32), CT, 0xf, 0xf, false);
        return __longlong_as_double(((long long)hi_ << 32) | (unsigned)lo_);
    };
    auto lane_f64 = [](double v, int l) {
        const long long r = __double_as_longlong(v);
        const int lo_ = __builtin_amdgcn_readlane((int)r, l), hi_ = __builtin_amdgcn_readlane((int)(r >> 32), l);
        return __longlong_as_double(((long long)hi_ << 32) | (unsigned)lo_);
    };
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int xy = 0; xy < 2; ++xy) {
            double d = a[j][xy];
            d += dpp_f64(d, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
            d += dpp_f64(d, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
            d += dpp_f64(d, std::integral_constant<int, 0x141>{});     // row_half_mirror
            d += dpp_f64(d, std::integral_constant<int, 0x140>{});     // row_mirror
            const double tot = (lane_f64(d, 0) + lane_f64(d, 16)) + (lane_f64(d, 32) + lane_f64(d, 48));
            if (lane == 0) sPart[wv][2 * j + xy] = tot;
        }
    __syncthreads();
    // this thread's row of inv_delta_C^T (requested behind the barrier -- earlier, its 24 registers would sit on top of the column loop's -- and in flight while the halves are added)
    float idc[24];
    {
        const int f = (tid >> 1) < F ? (tid >> 1) : 0;
#pragma unroll
        for (int k = 0; k < 24; ++k) idc[k] = (tid < 2 * F && k < K) ? P.inv_delta_c[k * K + f] : 0.0f;
    }
    if (tid < 2 * K) {
        const int k = tid >> 1, xy = tid & 1;
        const int cg_ = k / 3, j = k - 3 * cg_;
        sGT[tid] = sPart[cg_][2 * j + xy] + sPart[8 + cg_][2 * j + xy];
    }
    __syncthreads();
    if (tid < 2 * F) {
        const int xy = tid & 1;
        double s_ = 0.0;
#pragma unroll
        for (int k = 0; k < 24; ++k) if (k < K) s_ = fma((double)idc[k], sGT[2 * k + xy], s_);
        P.g_ctrl[(size_t)b * F * 2 + tid] = (float)s_;
    }
}

}  // namespace

namespace { int g_bwd_fixed_point = 0; }

TPSPP_EXPORT int tpspp_warp_bwd_set_accumulator(int fixed_point)
{
    g_bwd_fixed_point = fixed_point ? 1 : 0;
    return TPSPP_OK;
}

TPSPP_EXPORT size_t tpspp_warp_bwd_workspace_floats(int N, int Ho, int Wo)
{
    if (N <= 0 || Ho <= 0 || Wo <= 0) return 0;
    return (size_t)N * Ho * Wo * 2 * (1 + 2 * kBwdMaxG)       // dL/d grid + the sampling workgroups' slices
         + (size_t)N * kBwdMaxS * 2 * kMaxK * 2 + 2            // + the parameter kernel's partial dL/dT (fp64) per workgroup
         + (size_t)N + 2;                                      // + its arrival counters (one int per image)
}

TPSPP_EXPORT int tpspp_warp_bwd(const float* g_out0, const float* in0, int C0, int H0, int W0,
                                const float* g_out1, const float* in1, int C1, int H1, int W1,
                                const float* grid, const float* T, const float* inv_delta_c,
                                const float* p_hat, int p_hat_ld, const float* p_xy, const float* score,
                                const float* p_hat_t_or_null, int table_flags, int N, int F, int Ho, int Wo,
                                float* g_in0, float* g_in1, float* g_ctrl, float* g_score,
                                float* g_grid_ws, size_t g_grid_ws_floats, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(g_grid_ws_floats >= tpspp_warp_bwd_workspace_floats(N, Ho, Wo),
                  "tpspp_warp_bwd: g_grid_ws too small (needs tpspp_warp_bwd_workspace_floats(N, Ho, Wo) floats)");
    TPSPP_REQUIRE(g_out0 && in0 && grid && inv_delta_c && p_hat && g_ctrl && g_grid_ws, "tpspp_warp_bwd: null pointer");
    TPSPP_REQUIRE(T || !score, "tpspp_warp_bwd: T is needed with a score (dL/d score = 0.5 rbf (T . dL/d grid))");
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
    TPSPP_REQUIRE(N <= 65535, "tpspp_warp_bwd: N > 65535");
    const dim3 block(256);
    const size_t plane0 = (size_t)H0 * W0 * sizeof(float), plane1 = in1 ? (size_t)H1 * W1 * sizeof(float) : 0;
    int slots = 0;
    float* part = g_grid_ws + (size_t)N * P.n * 2;          // behind dL/d grid: the sampling workgroups' slices
    // the parameter kernel's partial dL/dT (fp64, behind the slices, at an 8-byte boundary) and arrival counters
    double* gT_part = reinterpret_cast<double*>(
        (reinterpret_cast<uintptr_t>(g_grid_ws + (size_t)N * P.n * 2 * (1 + 2 * kBwdMaxG)) + 7) & ~(uintptr_t)7);
    int* arrivals = reinterpret_cast<int*>(gT_part + (size_t)N * kBwdMaxS * 2 * kMaxK);
    bool arrivals_zeroed = false;                           // (the LDS-accumulating sampling kernel zeroes them itself)
    // kernel A'': two fixed-point planes (8 bytes per element) of a pass in <= 64 KB of LDS, <= 4096 output pixels
    // (1024-thread workgroups above 1024: the classic 32x100 geometry), planes of whole 16-byte pieces
    const size_t kLds2 = 16 * 1024;
    // the classic geometry: sampling + parameter gradients in one launch (warp_bwd_classic_kernel)
    const bool fixed_point = g_bwd_fixed_point || (table_flags & TPSPP_BWD_FIXED_POINT);
    // (p_xy != NULL means the TPS_PP table layout -- p_hat is (n, F) without the [1, x, y] columns this kernel reads from the
    // transposed classic table --, so it excludes the one-launch form whether or not a score is given; the classic rectifier
    // never passes it: tps_pp_amd/tps_preprocessor.py.  The environment switch is read once; per call: TPSPP_BWD_TWO_KERNELS.)
    static const bool two_kernels_env = getenv("TPSPP_BWD_TWO_KERNELS") != nullptr;
    const bool two_kernels = two_kernels_env || (table_flags & TPSPP_BWD_TWO_KERNELS);
    if (!in1 && !score && !p_xy && !fixed_point && !two_kernels && C0 <= 3 && F + 3 <= 24 &&
        p_hat_t_or_null && reinterpret_cast<uintptr_t>(p_hat_t_or_null) % 16 == 0 && P.n % 4 == 0 &&
        P.n > 1024 && P.n <= 4096 && (size_t)C0 * plane0 * 2 <= 78 * 1024 && (size_t)C0 * plane0 * 2 >= (size_t)P.n * 8 &&
        (size_t)C0 * plane0 * 2 >= (size_t)C0 * plane0 + ((size_t)W0 + 1) * 4 &&
        plane0 % 16 == 0 && reinterpret_cast<uintptr_t>(in0) % 16 == 0 && (!g_in0 || reinterpret_cast<uintptr_t>(g_in0) % 16 == 0)) {
        const size_t lds = (size_t)C0 * plane0 * 2;
        auto go = [&](auto cc) {
            constexpr int CC = decltype(cc)::value;
            static bool attr_done[tpspp::kMaxDevices] = {};
            if (tpspp::first_use_on_device(attr_done)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_bwd_classic_kernel<CC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
                (void)hipGetLastError();
            }
            hipLaunchKernelGGL(warp_bwd_classic_kernel<CC>, dim3((unsigned)N), dim3(kClassicNT), lds, st, P, g_grid_ws);
        };
        if (C0 == 1) go(std::integral_constant<int, 1>{});
        else if (C0 == 2) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 3>{});
        return tpspp::check_launch("tpspp_warp_bwd(classic)");
    }
    if (P.n <= 4096 && plane0 <= kLds2 && plane1 <= kLds2 && plane0 % 16 == 0 && plane1 % 16 == 0 &&
        (!g_in0 || reinterpret_cast<uintptr_t>(g_in0) % 16 == 0) && (!g_in1 || reinterpret_cast<uintptr_t>(g_in1) % 16 == 0)) {
        // channels per chunk: a chunk is worked off two planes at a time, so the LDS only ever holds two
        const int cpt0 = 8, cpt1 = 8;
        const int chunks0 = (C0 + cpt0 - 1) / cpt0, chunks1 = in1 ? (C1 + cpt1 - 1) / cpt1 : 0;
        const int most = chunks0 > chunks1 ? chunks0 : chunks1;
        const bool big = P.n > 1024;                        // 1024-thread workgroups, 4 pixels per thread
        int G = ((big ? 1024 : 4096) + N * P.nin - 1) / (N * P.nin);      // ~16 (4) workgroups of 256 (1024) threads per CU
        G = G < 1 ? 1 : (G > most ? most : G);
        G = G > kBwdMaxG ? kBwdMaxG : G;
        slots = G * P.nin;                                  // every slice is written by its workgroup (round 5: was 2 G with a
                                                            // 26-MB memset of the unused half for single-input calls)
        const size_t accb = 2 * 2 * (plane0 > plane1 ? plane0 : plane1);      // two planes x 8 bytes per element
        const dim3 g2((unsigned)(G * P.nin), (unsigned)N);
        auto go = [&](auto f64) {
            constexpr bool F64 = decltype(f64)::value;
            if (P.n <= 256)       hipLaunchKernelGGL((warp_bwd_sample_lds2_kernel<1, 256, F64>), g2, block, accb, st, P, G, cpt0, cpt1, part, arrivals);
            else if (P.n <= 512)  hipLaunchKernelGGL((warp_bwd_sample_lds2_kernel<2, 256, F64>), g2, block, accb, st, P, G, cpt0, cpt1, part, arrivals);
            else if (P.n <= 1024) hipLaunchKernelGGL((warp_bwd_sample_lds2_kernel<4, 256, F64>), g2, block, accb, st, P, G, cpt0, cpt1, part, arrivals);
            else                  hipLaunchKernelGGL((warp_bwd_sample_lds2_kernel<4, 1024, F64>), g2, dim3(1024), accb, st, P, G, cpt0, cpt1, part, arrivals);
        };
        if (fixed_point) go(std::false_type{});
        else go(std::true_type{});
        arrivals_zeroed = true;
    } else {
    if (hipMemsetAsync(g_grid_ws, 0, (size_t)N * P.n * 2 * sizeof(float), st) != hipSuccess)
        return tpspp::check_launch("tpspp_warp_bwd(memset)");
    // LDS-accumulating sampler backward when whole planes fit (64 KB per workgroup), else global atomics
    const size_t kLdsBudget = 64 * 1024;
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
    }
    // parameter gradients: S workgroups per image (about eight workgroups per CU in all, at least 256 pixels each), then
    // the reduction over S
    int S = (2048 + N - 1) / N;
    S = S < 1 ? 1 : (S > kBwdMaxS ? kBwdMaxS : S);
    while (S > 1 && (P.n + S - 1) / S < 256) --S;
    if (S > 1 && !arrivals_zeroed && hipMemsetAsync(arrivals, 0, (size_t)N * sizeof(int), st) != hipSuccess)
        return tpspp::check_launch("tpspp_warp_bwd(memset)");
    const dim3 grid_dim((unsigned)(N * S));
    auto launch_b = [&](auto kmax, auto table, auto transposed, auto scorem, auto gscore) {
        auto kern = warp_bwd_params_kernel<decltype(kmax)::value, decltype(table)::value, decltype(transposed)::value,
                                           decltype(scorem)::value, decltype(gscore)::value>;
        hipLaunchKernelGGL(kern, grid_dim, block, 0, st, P, g_grid_ws, part, slots, S, gT_part, arrivals);
    };
    auto pick_score = [&](auto kmax, auto table, auto transposed) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if (!score) launch_b(kmax, table, transposed, I0{}, std::false_type{});
        else if (P.score_t) { if (g_score) launch_b(kmax, table, transposed, I2{}, std::true_type{}); else launch_b(kmax, table, transposed, I2{}, std::false_type{}); }
        else { if (g_score) launch_b(kmax, table, transposed, I1{}, std::true_type{}); else launch_b(kmax, table, transposed, I1{}, std::false_type{}); }
    };
    auto pick_table = [&](auto kmax) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        if (p_xy) { if (p_hat_t_or_null) pick_score(kmax, I1{}, std::true_type{}); else pick_score(kmax, I1{}, std::false_type{}); }
        else { if (p_hat_t_or_null) pick_score(kmax, I0{}, std::true_type{}); else pick_score(kmax, I0{}, std::false_type{}); }
    };
    if (F + 3 <= 24)      pick_table(std::integral_constant<int, 24>{});
    else if (F + 3 <= 36) pick_table(std::integral_constant<int, 36>{});
    else                  pick_table(std::integral_constant<int, 64>{});
    return tpspp::check_launch("tpspp_warp_bwd");
}
