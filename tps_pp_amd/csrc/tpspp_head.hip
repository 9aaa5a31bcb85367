// The recogniser head that consumes the rectified features: NRTR
// transformer encoder and greedy transformer decoder, fp32, one call per batch each.
//
// Replaces (reference, mmocr/models/): textrecog/encoders/nrtr_encoder.py:67-87,
// textrecog/decoders/nrtr_decoder.py:95-113,131-177, common/layers/transformer_layers.py:57-75,133-163,
// common/modules/transformer_module.py:24-33,75-99,119-125,155-163.
//
// Data layout: every activation is a CHANNEL-MAJOR matrix X[c][m] (row = feature, column = token,
// m = image*T + token for the encoder, m = image for one decoder step), i.e. exactly the (1, C, 1, M)
// "image" of a 1x1 convolution.  All projections therefore run on the library's fp32 MFMA
// convolution kernel (tpspp_conv.hip) with bias / GELU / residual fused in its epilogue and columns
// contiguous in every load and store; the encoder's NCHW input needs one re-layout on entry.  Where a
// consumer wants TOKEN-major rows (the decoder's q/k/v of one step, the cross-attention values) the
// same kernel is called with the operands swapped (the activation as its "weights", the prepared
// weight as its "image"): D[m][cout] instead of D[cout][m], no transpose kernel.
//
// The decoder is incremental: the reference re-runs the whole padded target (41 positions) through all
// layers at each of its 40 steps and keeps one row; with a causal mask that row depends only on earlier
// positions, so one position per step against cached self-attention keys/values gives the same numbers
// (up to summation order) with 1/41 of the arithmetic.  The encoder keys/values of the cross attention
// are projected once per layer.  The whole loop is enqueued without host synchronisation: the arg-max
// of step s is written to a device token buffer that step s+1's embedding kernel reads.
//
// Bounds: projections: MFMA (fp32 matrix rate); LayerNorm / attention kernels: HBM / L2 (one pass over
// their operands), LDS-broadcast reads in the encoder attention.
#include <cmath>

#include "tpspp_common.h"

#include <mutex>
#include "tpspp_tokgemm.h"

#include <cstdlib>

namespace {

constexpr int kWave = 64;
constexpr int kDK = 64;                 // head width (d_k = d_v = 64: every NRTR config of the reference)
constexpr int kKRow = kDK + 4;          // LDS row stride of the staged keys / values (16-B aligned rows)

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

// ---- re-layouts ---------------------------------------------------------------------------------
// (N, C, T) -> (C, N*T): out[c][b*T + t] = in[b][c][t]   (rows of T stay contiguous on both sides)
__global__ void __launch_bounds__(256)
nct_to_cm_kernel(const float* __restrict__ in, int N, int C, int T, float* __restrict__ out)
{
    const size_t total = (size_t)N * C * T;
    const size_t M = (size_t)N * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i / M, m = i - c * M;
        const size_t b = m / T, t = m - b * T;
        out[i] = in[(b * C + c) * T + t];
    }
}

// out (cols, rows) = in (rows, cols)^T through a padded 32x32 LDS tile
__global__ void __launch_bounds__(256)
transpose2d_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        tile[ty + k][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][ty + k];
    }
}

// ---- LayerNorm over the rows of a channel-major matrix ----------------------------------------------
// y[c][m] = (x[c][m] - mean_m) * rstd_m * gamma[c] + beta[c];  mean / biased variance over c.
// Workgroup = 16 columns x 16 interleaved channel slices; a thread keeps its C/16 values in registers
// (one pass over x, all loads in flight together: a decoder step has only a few hundred columns, so
// the kernel is latency-bound unless every load is issued up front).  C <= 16 * VPT.
// (round 6: compiled for three workgroups per CU -- VPT = 32 took 176 registers, two wavefronts per SIMD, in a kernel that only
// waits for memory)
template <int VPT>
__global__ void __launch_bounds__(256, VPT <= 32 ? 3 : 1)
layernorm_cm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                    int C, int M, float eps, float* __restrict__ y)
{
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int m = blockIdx.x * 16 + col;
    const bool ok = m < M;
    const int mm = ok ? m : M - 1;
    float v[VPT], g[VPT], be[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {                    // every load of the thread issued up front
        const int c = sl + 16 * i;
        const int cc = c < C ? c : C - 1;
        v[i] = x[(size_t)cc * M + mm];
        g[i] = gamma[cc];
        be[i] = beta[cc];
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) s += (sl + 16 * i) < C ? v[i] : 0.0f;
    red[sl][col] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][col];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const float d = (sl + 16 * i) < C ? v[i] - mean : 0.0f;
        q = fmaf(d, d, q);
    }
    red[sl][col] = q;
    __syncthreads();
    tot = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += red[j][col];
    const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = sl + 16 * i;
        if (c < C) y[(size_t)c * M + m] = (v[i] - mean) * rstd * g[i] + be[i];
    }
}

// ---- encoder self-attention -------------------------------------------------------------------------
// qkv (3C, M) channel-major, M = N*T; head h of image b: rows [64h, 64h+64) of each third, columns
// [bT, bT+T).  Workgroup = one (image, head, block of <=256 queries): the head's keys and values are
// staged in LDS ([key][feature], read back as wavefront-wide broadcasts), one query per thread with q
// and the output accumulator in registers.  Two passes over the keys (row maximum, then exp / sum /
// weighted values): scores are recomputed rather than stored, so T is bounded only by LDS (T <= 256).
// Keys >= valid_len[b] are masked (nrtr_encoder.py:51-65: the first ceil(T*valid_ratio) tokens count).
__global__ void __launch_bounds__(256)
attn_enc_kernel(const float* __restrict__ qkv, int C, int M, int T, const int* __restrict__ valid_len,
                float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                         // [T][kKRow]
    float* Vs = smem + (size_t)T * kKRow;
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t col0 = (size_t)b * T;
    // staging: 8 key rows + 8 value rows per wavefront in flight (a one-wavefront workgroup would
    // otherwise pay a full memory latency for each of its 128 rows)
    for (int j0 = 0; j0 < T; j0 += kWave) {
        const int j = j0 + lane;
        const int jj = j < T ? j : T - 1;
        for (int d0 = wv; d0 < kDK; d0 += 8 * nw) {
            float kk[8], vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * nw < kDK ? d0 + u * nw : wv;
                kk[u] = qkv[(size_t)(C + kDK * h + d) * M + col0 + jj];
                vv[u] = qkv[(size_t)(2 * C + kDK * h + d) * M + col0 + jj];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * nw;
                if (d < kDK && j < T) {
                    Ks[j * kKRow + d] = kk[u];
                    Vs[j * kKRow + d] = vv[u];
                }
            }
        }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    int nvalid = valid_len ? valid_len[b] : T;
    nvalid = nvalid < T ? nvalid : T;
    float q[kDK];
#pragma unroll
    for (int d = 0; d < kDK; ++d) q[d] = qkv[(size_t)(kDK * h + d) * M + col0 + i] * 0.125f;   // q / sqrt(d_k)

    auto score = [&](int j) {
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * kKRow);
        float s = 0.0f;
#pragma unroll
        for (int d4 = 0; d4 < kDK / 4; ++d4) {
            const float4 kv = kr[d4];
            s = fmaf(q[4 * d4 + 0], kv.x, s);
            s = fmaf(q[4 * d4 + 1], kv.y, s);
            s = fmaf(q[4 * d4 + 2], kv.z, s);
            s = fmaf(q[4 * d4 + 3], kv.w, s);
        }
        return s;
    };
    float mx = -INFINITY;
    for (int j = 0; j < nvalid; ++j) mx = fmaxf(mx, score(j));
    float acc[kDK];
#pragma unroll
    for (int d = 0; d < kDK; ++d) acc[d] = 0.0f;
    float l = 0.0f;
    for (int j = 0; j < nvalid; ++j) {
        const float p = expf(score(j) - mx);
        l += p;
        const float4* vr = reinterpret_cast<const float4*>(Vs + j * kKRow);
#pragma unroll
        for (int d4 = 0; d4 < kDK / 4; ++d4) {
            const float4 vv = vr[d4];
            acc[4 * d4 + 0] = fmaf(p, vv.x, acc[4 * d4 + 0]);
            acc[4 * d4 + 1] = fmaf(p, vv.y, acc[4 * d4 + 1]);
            acc[4 * d4 + 2] = fmaf(p, vv.z, acc[4 * d4 + 2]);
            acc[4 * d4 + 3] = fmaf(p, vv.w, acc[4 * d4 + 3]);
        }
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < kDK; ++d) out[(size_t)(kDK * h + d) * M + col0 + i] = acc[d] * inv;
}

// ---- encoder self-attention on the fp32 matrix cores (T <= 128) ----------------------------------------------------
// One wavefront = 32 queries of one (image, head).  Both products are v_mfma_f32_32x32x2_f32 with operands
// taken straight from the channel-major rows (a fragment = two 128-B row segments), nothing staged:
//   S^T tile (32 keys x 32 queries) = sum_d K[d][j] Q[d][i]   -> in the C/D layout a LANE is a QUERY and its
//     16 registers are 16 keys (the other 16 keys of the tile sit in lane^32), so the softmax is a
//     per-lane reduction plus one exchange with lane^32;
//   O^T tile (32 features x 32 queries) = sum_j V[d][j] P^T[j][i]: the probabilities are consumed as the B
//     operand IN PLACE -- register r of lane (i, half) holds key 8*(r>>2) + (r&3) + 4*half of its tile, and
//     those two keys (half = 0, 1) are taken as the k-pair of MFMA step r; the A operand reads V at the
//     same two keys.  No LDS, no barrier.
// NJ = ceil(T / 32) key tiles (template, <= 4).  Scores are scaled by 1/8 through q, keys >= valid_len[b]
// get probability 0.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// (round 6: compiled for three workgroups per CU -- NJ = 2, the 64-token case, took 172 registers = two wavefronts per SIMD,
// 168 is three; the kernel waits for memory)
template <int NJ>
__global__ void __launch_bounds__(256, NJ <= 2 ? 3 : 1)
attn_enc_mfma_kernel(const float* __restrict__ qkv, int C, int M, int T, const int* __restrict__ valid_len,
                     float* __restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l31 = lane & 31;
    const int nib = (T + 31) >> 5;                           // query blocks per (image, head)
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int H = C / kDK;
    const int ib = w % nib, bh = w / nib;
    if (bh >= (M / T) * H) return;                           // wave-uniform (M / T = images)
    const int h = bh % H, b = bh / H;
    const size_t col0 = (size_t)b * T;
    int nvalid = valid_len ? valid_len[b] : T;
    nvalid = nvalid < T ? nvalid : T;
    const int i = ib * 32 + l31;                             // this lane's query (B-operand column)
    const int ic = i < T ? i : T - 1;
    const float* qp = qkv + (size_t)(kDK * h) * M + col0;
    const float* kp = qkv + (size_t)(C + kDK * h) * M + col0;
    const float* vp = qkv + (size_t)(2 * C + kDK * h) * M + col0;

    // B fragments of Q for all 32 k-steps (k = feature pair 2*ks + half), scaled
    float qf[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) qf[ks] = qp[(size_t)(2 * ks + half) * M + ic] * 0.125f;

    f32x16_t sc[NJ];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[jb][r] = 0.0f;
        const int j = jb * 32 + l31;                         // A-operand row: key
        const int jc = j < T ? j : T - 1;
        float kf[32];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) kf[ks] = kp[(size_t)(2 * ks + half) * M + jc];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) sc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[ks], qf[ks], sc[jb], 0, 0, 0);
    }
    // softmax over the keys of query i: registers of this lane + the complementary keys in lane^32
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[jb][r] = j < nvalid ? sc[jb][r] : -INFINITY;
            mx = fmaxf(mx, sc[jb][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    float l = 0.0f;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = sc[jb][r] == -INFINITY ? 0.0f : expf(sc[jb][r] - mx);
            sc[jb][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32, kWave);
    const float inv = 1.0f / l;

    // O^T = V P^T, two feature tiles of 32
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        f32x16_t o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        const float* vrow = vp + (size_t)(dt * 32 + l31) * M;   // A-operand row: feature
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb) {
            float vf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                vf[r] = vrow[j < T ? j : T - 1];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], sc[jb][r], o, 0, 0, 0);
        }
        if (i < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(size_t)(kDK * h + d) * M + col0 + i] = o[r] * inv;
            }
        }
    }
}

// ---- decoder, one step: embedding + position table ----------------------------------------------------
// x[c][b] = emb[tokens[b][step]][c] + pos[step][c]      (nrtr_decoder.py:96-98, no scaling)
// tm: x is token-major (Nb, C) instead of channel-major (C, Nb) (the split-K step GEMM's layout)
__global__ void __launch_bounds__(256)
dec_embed_kernel(const float* __restrict__ emb, const float* __restrict__ pos, const int* __restrict__ tokens,
                 int Lt, int step, int C, int Nb, float* __restrict__ x, int tm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * Nb) return;
    const int c = tm ? i % C : i / Nb, b = tm ? i / C : i - c * Nb;
    const int tok = tokens[(size_t)b * Lt + step];
    x[i] = emb[(size_t)tok * C + c] + pos[(size_t)step * C + c];
}

// ---- decoder, one step: attention against the caches / the encoder, 16-byte loads -----------------------------------
// qkv_t (Nb, 3C) / q_t (Nb, C) TOKEN-major (this step's projections); one wavefront per (image, head).
// Self-attention (nrtr_decoder.py:100-102): key p is valid iff p <= step and tokens[b][p] != <PAD>; the new position is
// used from registers, so nothing written by the kernel is read back by it.  Cross-attention (nrtr_decoder.py:115-129):
// keys >= valid_len[b] are masked.  KV = float, or unsigned short: bf16 caches / encoder keys and values
// (TPSPP_HEAD_BF16: the encoder K/V are the only HBM-bound operand of a step; the position being decoded is used from its
// fp32 registers, earlier positions as they were rounded when written; scores, softmax, weighted sum in fp32).
// Rounds 1-2 gave a lane one 4-byte element per load (a key row per lane, a value row per wavefront): 128 load
// instructions of 256 bytes per (image, head), and the vector-memory unit takes 16 cycles per instruction whatever its
// width -- the cross-attention's 134 MB per layer-step moved at 4 TB/s.  Here keys AND values are token-major
// ((tokens, C) rows, a head's 64 features contiguous), a lane takes 16 bytes (EPL = 4 fp32 or 8 bf16 features) of a
// token, GS = 64 / EPL lanes share a token and a load instruction covers 64 / GS tokens x one contiguous head row:
// 4x (8x) fewer instructions for the same bytes.  Scores: per-lane partial dot products, summed inside the GS-lane
// group on the DPP path (quad swaps, half-row / row mirror); values: per-lane partial sums over the lane's tokens,
// summed across the groups with a few lane exchanges at the very end.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int GS> __device__ __forceinline__ float group_sum(float v)      // every lane of a GS-lane group: the group's sum
{
    v = dpp_add<0xB1>(v);                      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);                     // row_half_mirror: 8 lanes
    if (GS == 16) v = dpp_add<0x140>(v);       // row_mirror: 16 lanes
    return v;
}
template <int GS> __device__ __forceinline__ float across_groups_sum(float v)   // sum over the 64 / GS groups (same lane of each)
{
    if (GS == 8) v += __shfl_xor(v, 8, kWave);
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}
template <int GS> __device__ __forceinline__ float across_groups_max(float v)
{
    if (GS == 8) v = fmaxf(v, __shfl_xor(v, 8, kWave));
    v = fmaxf(v, __shfl_xor(v, 16, kWave));
    v = fmaxf(v, __shfl_xor(v, 32, kWave));
    return v;
}

// ---- system-scope (sc0 sc1) accesses for data exchanged between workgroups INSIDE one launch (the persistent decoder
// step, tpspp_head_persist.h): they bypass the CU's L1 and the XCD's L2, so a value written by another CU -- maybe on
// another XCD -- is never served from a stale line (MI355X_MICROARCH.md, "sc0 sc1 stores and loads both sides").  The loads
// are inline asm, invisible to the compiler's wait-count tracking: every use must sit behind a wait_sys() that names the
// loaded registers.
typedef float hf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hf32x4 ld16_sys(const float* p)
{
    hf32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st16_sys(float* p, hf32x4 v)
{
    // (s_nop: the VMEM store-data hazard is invisible to the compiler inside inline asm, as in tpspp_warp_pair.h)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// `plain` (the persistent step kernel, when the 16 workgroups of a cluster were verified to sit on ONE XCD): an ordinary store --
// the line stays in that XCD's L2, where the cluster's sc0 sc1 loads (which bypass L1 only) find it at the L2's latency; a
// write-through store drops the line and the same reader fetches it over the fabric (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void st16_x(float* p, hf32x4 v, bool plain)
{
    if (plain) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wait_sys(hf32x4& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void wait_sys(hf32x4& a, hf32x4& b, hf32x4& c)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
}
// the same load as two 8-byte system-scope atomic loads: VISIBLE to the compiler (it tracks their wait count itself and may
// schedule them among other loads) -- for small operands whose latency should overlap other requests
__device__ __forceinline__ hf32x4 ld16_sys_v(const float* p)
{
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return hf32x4{__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                  __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32))};
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct NoPre { __device__ __forceinline__ bool operator()() const { return true; } };   // (see self_attend's `pre`)

template <typename KV> struct Wide;
template <> struct Wide<float> {
    static constexpr int EPL = 4;
    typedef float4 raw;
    static __device__ __forceinline__ void unpack(const raw& r, float (&f)[4]) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
    static __device__ __forceinline__ raw pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Wide<unsigned short> {
    static constexpr int EPL = 8;
    typedef uint4 raw;
    static __device__ __forceinline__ void unpack(const raw& r, float (&f)[8])
    {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __builtin_bit_cast(float, w[i] << 16); f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ raw pack(const float (&f)[8])
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // round to nearest even, as kvc_store
            unsigned a = __builtin_bit_cast(unsigned, f[2 * i]), b = __builtin_bit_cast(unsigned, f[2 * i + 1]);
            a += 0x7fffu + ((a >> 16) & 1u); b += 0x7fffu + ((b >> 16) & 1u);
            w[i] = (a >> 16) | (b & 0xffff0000u);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// cross-attention of one (image b, head h) by one wavefront: q (this lane's EPL features, already scaled), Kx_t / Vx_t
// (Nb*T, C) token-major, out (Nb, C) token-major (or channel-major).
// MAXJJ: 64-token groups the code is built for (T <= 64 MAXJJ; 1 saves 48 registers of score storage)
template <typename KV, bool SYS = false, int MAXJJ = 4>
__device__ __forceinline__ void cross_attend(const float (&q)[Wide<KV>::EPL], const KV* __restrict__ Kx_t, const KV* __restrict__ Vx_t,
                                             int C, int Nb, int T, int nvalid, int b, int h, int lane, float* __restrict__ out,
                                             int out_cm, bool plain_st = false)
{
    typedef Wide<KV> Wd;
    constexpr int EPL = Wd::EPL, GS = kDK / EPL, TPI = kWave / GS, NP = kWave / TPI;   // tokens per instruction, pieces per 64 tokens
    const int grp = lane / GS, dl = lane % GS;             // this lane: tokens grp, grp + TPI, ...; features EPL dl ..
    const typename Wd::raw* kb = reinterpret_cast<const typename Wd::raw*>(Kx_t + ((size_t)b * T) * C + kDK * h + EPL * dl);
    const typename Wd::raw* vb = reinterpret_cast<const typename Wd::raw*>(Vx_t + ((size_t)b * T) * C + kDK * h + EPL * dl);
    const size_t rstride = (size_t)C / EPL;                // row pitch in raw pieces
    float sc[MAXJJ][NP];
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < MAXJJ; ++jj) {
#pragma unroll
        for (int i = 0; i < NP; ++i) sc[jj][i] = -INFINITY;
        if (jj * kWave < nvalid) {                         // wave-uniform
            typename Wd::raw kr[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {                 // every key piece of these 64 tokens in flight together
                const int t = jj * kWave + TPI * i + grp;
                kr[i] = kb[(size_t)(t < T ? t : T - 1) * rstride];
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float kf[EPL];
                Wd::unpack(kr[i], kf);
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) s = fmaf(q[e], kf[e], s);
                s = group_sum<GS>(s);
                const int t = jj * kWave + TPI * i + grp;
                sc[jj][i] = t < nvalid ? s : -INFINITY;
                mx = fmaxf(mx, sc[jj][i]);
            }
        }
    }
    mx = across_groups_max<GS>(mx);
    float l = 0.0f;
#pragma unroll
    for (int jj = 0; jj < MAXJJ; ++jj)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            sc[jj][i] = sc[jj][i] == -INFINITY ? 0.0f : expf(sc[jj][i] - mx);
            l += sc[jj][i];
        }
    l = across_groups_sum<GS>(l);
    const float inv = 1.0f / l;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int jj = 0; jj < MAXJJ; ++jj) {
        if (jj * kWave < nvalid) {
            typename Wd::raw vr[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int t = jj * kWave + TPI * i + grp;
                vr[i] = vb[(size_t)(t < T ? t : T - 1) * rstride];
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float vf[EPL];
                Wd::unpack(vr[i], vf);
                const float pw = sc[jj][i] * inv;          // 0 for masked tokens
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pw, vf[e], acc[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = across_groups_sum<GS>(acc[e]);
    if (grp == 0) {
        if (out_cm) {                                      // channel-major (C, Nb): the exact-fp32 step GEMMs' operand layout
#pragma unroll
            for (int e = 0; e < EPL; ++e) out[(size_t)(kDK * h + EPL * dl + e) * Nb + b] = acc[e];
        } else {
            float* o = out + (size_t)b * C + kDK * h + EPL * dl;
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                if (SYS) st16_x(o + e, hf32x4{acc[e], acc[e + 1], acc[e + 2], acc[e + 3]}, plain_st);
                else *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
            }
        }
    }
}

// cross-attention of image b, heads h and h + 1, by one wavefront, T <= 64 encoder tokens: cross_attend's arithmetic per head
// (same fragments, same reduction order), every stage run for BOTH heads before the next one starts -- the keys of both heads
// are in flight together, then the values of both (8 wavefronts per CU must keep as many requests outstanding as the 16 of
// the launch form; in the launch form itself two heads per wavefront measured 22-23 us per launch against 28.6 with one).
// SYS: q via compiler-visible system-scope loads, output system-scope (the persistent step kernel).
template <typename KV, bool SYS, typename Pre = NoPre>
__device__ __forceinline__ bool cross_attend2(const float* __restrict__ qrow, const KV* __restrict__ Kx_t, const KV* __restrict__ Vx_t,
                                              int C, int T, int nvalid, int b, int h, int lane, float* __restrict__ out, Pre pre = Pre(), bool plain_st = false)
{
    typedef Wide<KV> Wd;
    constexpr int EPL = Wd::EPL, GS = kDK / EPL, TPI = kWave / GS, NP = kWave / TPI, NH = 2;
    const int grp = lane / GS, dl = lane % GS;
    const size_t rstride = (size_t)C / EPL;                // row pitch in raw pieces
    typename Wd::raw kr[NH][NP], vr[NH][NP];
    float q[NH][EPL], sc[NH][NP], inv[NH];
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const typename Wd::raw* kb = reinterpret_cast<const typename Wd::raw*>(Kx_t + ((size_t)b * T) * C + kDK * (h + n) + EPL * dl);
#pragma unroll
        for (int i = 0; i < NP; ++i) {                     // every key piece of both heads in flight together
            const int t = TPI * i + grp;
            kr[n][i] = kb[(size_t)(t < T ? t : T - 1) * rstride];
        }
    }
    // (the keys do not depend on this step: they were requested in front of `pre`, the barrier behind which q is readable)
    if (!pre()) return false;
#pragma unroll
    for (int n = 0; n < NH; ++n) {
#pragma unroll
        for (int e = 0; e < EPL / 4; ++e) {
            const float* qp = qrow + kDK * (h + n) + EPL * dl + 4 * e;
            const hf32x4 rq = SYS ? ld16_sys_v(qp) : *reinterpret_cast<const hf32x4*>(qp);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[n][4 * e + i] = rq[i] * 0.125f;
        }
    }
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float kf[EPL];
            Wd::unpack(kr[n][i], kf);
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s = fmaf(q[n][e], kf[e], s);
            s = group_sum<GS>(s);
            const int t = TPI * i + grp;
            sc[n][i] = t < nvalid ? s : -INFINITY;
            mx = fmaxf(mx, sc[n][i]);
        }
        mx = across_groups_max<GS>(mx);
        float l = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            sc[n][i] = sc[n][i] == -INFINITY ? 0.0f : expf(sc[n][i] - mx);
            l += sc[n][i];
        }
        l = across_groups_sum<GS>(l);
        inv[n] = 1.0f / l;
    }
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const typename Wd::raw* vb = reinterpret_cast<const typename Wd::raw*>(Vx_t + ((size_t)b * T) * C + kDK * (h + n) + EPL * dl);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int t = TPI * i + grp;
            vr[n][i] = vb[(size_t)(t < T ? t : T - 1) * rstride];
        }
    }
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        float acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float vf[EPL];
            Wd::unpack(vr[n][i], vf);
            const float pw = sc[n][i] * inv[n];            // 0 for masked tokens
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pw, vf[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = across_groups_sum<GS>(acc[e]);
        if (grp == 0) {
            float* o = out + (size_t)b * C + kDK * (h + n) + EPL * dl;
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                if (SYS) st16_x(o + e, hf32x4{acc[e], acc[e + 1], acc[e + 2], acc[e + 3]}, plain_st);
                else *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
            }
        }
    }
    return true;
}


// cross-attention: q_t (Nb, C), Kx_t / Vx_t (Nb*T, C) token-major, out (Nb, C) token-major.  One wavefront per (image, head).
template <typename KV>
__global__ void __launch_bounds__(256)
attn_dec_cross_wide_kernel(const float* __restrict__ q_t, const KV* __restrict__ Kx_t, const KV* __restrict__ Vx_t,
                           int C, int Nb, int H, int T, const int* __restrict__ valid_len, float* __restrict__ out, int out_cm)
{
    constexpr int EPL = Wide<KV>::EPL, GS = kDK / EPL;
    const int lane = threadIdx.x & (kWave - 1);
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= Nb * H) return;
    const int b = pair / H, h = pair - b * H;
    const int dl = lane % GS;
    float q[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = q_t[(size_t)b * C + kDK * h + EPL * dl + e] * 0.125f;
    int nvalid = valid_len ? valid_len[b] : T;
    nvalid = nvalid < T ? nvalid : T;
    cross_attend<KV>(q, Kx_t, Vx_t, C, Nb, T, nvalid, b, h, lane, out, out_cm);
}

// two heads per wavefront (T <= 64, H even, token-major output): q_t rows at pitch `ldq`
template <typename KV>
__global__ void __launch_bounds__(256)
attn_dec_cross_wide2_kernel(const float* __restrict__ q_t, int ldq, const KV* __restrict__ Kx_t, const KV* __restrict__ Vx_t,
                            int C, int Nb, int H, int T, const int* __restrict__ valid_len, float* __restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int pair2 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // (image, head pair)
    const int H2 = H >> 1;
    if (pair2 >= Nb * H2) return;
    const int b = pair2 / H2, h = 2 * (pair2 - b * H2);
    int nvalid = valid_len ? valid_len[b] : T;
    nvalid = nvalid < T ? nvalid : T;
    cross_attend2<KV, false>(q_t + (size_t)b * ldq, Kx_t, Vx_t, C, T, nvalid, b, h, lane, out);
}


// masked self-attention against token-major caches Kc / Vc [image][head][position][64]; qkv_t (Nb, 3C); out (Nb, C).
// The new position's key / value are used from registers (fp32) and appended to the caches.
// NH consecutive heads h .. h + NH - 1 of image b by one wavefront (NH = 1: the launch-per-phase kernel; 2: the persistent
// step kernel, whose 8 wavefronts per CU need both heads' requests in flight together to keep the memory system as busy as
// the 16 wavefronts per CU of the launch form do -- every stage below runs over the NH heads before the next stage starts).
// SYS (the persistent step kernel): this step's q | k | v row and the output are exchanged with other workgroups of the same
// launch -- system-scope accesses; the caches belong to earlier / later launches and stay plain.
// `pre` (the persistent step kernel): called once the requests that do NOT depend on this step's projections are in flight --
// the cached keys / values and the token ids --; it is the cluster barrier behind which q | k | v become readable (false =
// barrier timeout: give up).
template <typename KV, bool SYS, int NH = 1, typename Pre = NoPre>
__device__ __forceinline__ bool self_attend(const float* __restrict__ qkv_t, int C, int Nb, int H, int step, int Lmax,
                                            KV* __restrict__ Kc, KV* __restrict__ Vc, const int* __restrict__ tokens, int Lt,
                                            int pad_idx, float* __restrict__ out, int out_cm, int b, int h, int lane, Pre pre = Pre(), bool plain_st = false)
{
    typedef Wide<KV> Wd;
    constexpr int EPL = Wd::EPL, GS = kDK / EPL, TPI = kWave / GS, NP = kWave / TPI;
    const int grp = lane / GS, dl = lane % GS;
    float q[NH][EPL], k[NH][EPL], v[NH][EPL];
    typename Wd::raw *kc[NH], *vc[NH];
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const size_t bh = (size_t)b * H + h + n;
        kc[n] = reinterpret_cast<typename Wd::raw*>(Kc + bh * Lmax * kDK + EPL * dl);
        vc[n] = reinterpret_cast<typename Wd::raw*>(Vc + bh * Lmax * kDK + EPL * dl);
    }
    constexpr int rstride = GS;                            // a cached row is GS pieces

    // Every load of the wavefront is requested here, before the first use of any of them: the cached keys AND values of the
    // earlier positions (the values do not depend on the scores) and the positions' token ids (the <PAD> mask).  Round 4: the
    // ids used to be read inside the score loop, one 4-byte load and one s_waitcnt vmcnt(0) per piece, and the values behind
    // the softmax -- eleven dependent round trips to memory in a kernel that moves 5-20 KB per wavefront (9-13 us per launch,
    // 240 launches per batch); now three (four with fp32 caches).
    // (fp32 caches: keys + values together are 128 registers of loads in flight, 140 in all -- three wavefronts per SIMD
    // instead of four, i.e. a second, nearly empty round of the 4096 wavefronts: 16.7 us against 13.4; their values stay
    // behind the softmax.  bf16 caches: 123 registers, 9.2 -> 8.9 us.)
    constexpr bool kValuesEarly = sizeof(KV) == 2;
    float sc[NH][NP];
    typename Wd::raw kr[NH][NP], vr[NH][NP];
    int tk[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = TPI * i + grp;
        tk[i] = pad_idx;
        if (TPI * i <= step) {                                 // (uniform tests: whole pieces beyond `step` are skipped)
            const int* tp = tokens + (size_t)b * Lt + (p <= step ? p : 0);
            // (SYS: the previous step of the same launch wrote the newest id)
            tk[i] = SYS ? __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *tp;
        }
        if (TPI * i < step) {
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                kr[n][i] = kc[n][(size_t)(p < step ? p : 0) * rstride];
                if (kValuesEarly) vr[n][i] = vc[n][(size_t)(p < step ? p : 0) * rstride];
            }
        }
    }
    if (!pre()) return false;
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const float* base = qkv_t + (size_t)b * 3 * C + kDK * (h + n) + EPL * dl;
        if (SYS) {
            // (compiler-visible system-scope loads: they stay in flight together with the cache rows requested above)
#pragma unroll
            for (int e = 0; e < EPL / 4; ++e) {
                const hf32x4 rq = ld16_sys_v(base + 4 * e), rk = ld16_sys_v(base + C + 4 * e), rv = ld16_sys_v(base + 2 * C + 4 * e);
#pragma unroll
                for (int i = 0; i < 4; ++i) { q[n][4 * e + i] = rq[i] * 0.125f; k[n][4 * e + i] = rk[i]; v[n][4 * e + i] = rv[i]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { q[n][e] = base[e] * 0.125f; k[n][e] = base[C + e]; v[n][e] = base[2 * C + e]; }
        }
    }
    // the new position's key / value join the caches (rows `step`: never among the rows read above)
    if (grp == 0) {
#pragma unroll
        for (int n = 0; n < NH; ++n) { kc[n][(size_t)step * rstride] = Wd::pack(k[n]); vc[n][(size_t)step * rstride] = Wd::pack(v[n]); }
    }
    float inv[NH];
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        float cur = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cur = fmaf(q[n][e], k[n][e], cur);
        cur = group_sum<GS>(cur);
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = TPI * i + grp;
            float s = -INFINITY;
            if (TPI * i <= step) {                             // uniform
                float dot = 0.0f;
                if (TPI * i < step) {
                    float kf[EPL];
                    Wd::unpack(kr[n][i], kf);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) dot = fmaf(q[n][e], kf[e], dot);
                    dot = group_sum<GS>(dot);
                }
                if (p == step) dot = cur;
                const bool valid = p <= step && tk[i] != pad_idx;
                s = valid ? dot : -INFINITY;
            }
            sc[n][i] = s;
            mx = fmaxf(mx, s);
        }
        mx = across_groups_max<GS>(mx);
        float l = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) { sc[n][i] = sc[n][i] == -INFINITY ? 0.0f : expf(sc[n][i] - mx); l += sc[n][i]; }
        l = across_groups_sum<GS>(l);
        inv[n] = 1.0f / l;
    }
    if (!kValuesEarly) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = TPI * i + grp;
            if (TPI * i < step) {
#pragma unroll
                for (int n = 0; n < NH; ++n) vr[n][i] = vc[n][(size_t)(p < step ? p : 0) * rstride];
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        float acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = TPI * i + grp;
            if (TPI * i <= step) {
                const float pw = sc[n][i] * inv[n];
                if (p == step) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pw, v[n][e], acc[e]);
                } else if (p < step) {
                    float vf[EPL];
                    Wd::unpack(vr[n][i], vf);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pw, vf[e], acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = across_groups_sum<GS>(acc[e]);
        if (grp == 0) {
            if (out_cm) {                                      // channel-major (C, Nb): the exact-fp32 step GEMMs' operand layout
#pragma unroll
                for (int e = 0; e < EPL; ++e) out[(size_t)(kDK * (h + n) + EPL * dl + e) * Nb + b] = acc[e];
            } else {
                float* o = out + (size_t)b * C + kDK * (h + n) + EPL * dl;
#pragma unroll
                for (int e = 0; e < EPL; e += 4) {
                    if (SYS) st16_x(o + e, hf32x4{acc[e], acc[e + 1], acc[e + 2], acc[e + 3]}, plain_st);
                    else *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
                }
            }
        }
    }
    return true;
}

template <typename KV>
__global__ void __launch_bounds__(256)
attn_dec_self_wide_kernel(const float* __restrict__ qkv_t, int C, int Nb, int H, int step, int Lmax,
                          KV* __restrict__ Kc, KV* __restrict__ Vc, const int* __restrict__ tokens, int Lt,
                          int pad_idx, float* __restrict__ out, int out_cm)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= Nb * H) return;
    const int b = pair / H, h = pair - b * H;
    self_attend<KV, false>(qkv_t, C, Nb, H, step, Lmax, Kc, Vc, tokens, Lt, pad_idx, out, out_cm, b, h, lane);
}

// ---- decoder, one step: classifier epilogue ------------------------------------------------------------------
// logits (Cc, Nb) channel-major.  Greedy: out[b][step][:] = softmax(logits[:, b]) and
// tokens[b][step+1] = arg-max (first maximum)  (nrtr_decoder.py:168-175).  Forced: out = raw logits.
__global__ void __launch_bounds__(256)
dec_classify_kernel(const float* __restrict__ logits, int Cc, int Nb, int step, int L, int greedy,
                    float* __restrict__ out, int* __restrict__ tokens, int Lt, int tm,
                    const float* __restrict__ emb = nullptr, const float* __restrict__ pos = nullptr,
                    float* __restrict__ x_next = nullptr, int C = 0)
{
    // one wavefront per image, lanes over classes
    const int lane = threadIdx.x & (kWave - 1);
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= Nb) return;
    float* o = out + ((size_t)b * L + step) * Cc;
    const size_t ls = tm ? 1 : (size_t)Nb;                 // stride between the classes of an image
    logits += tm ? (size_t)b * Cc : (size_t)b;
    // x_next (token-major step pipeline): the next step's embedding row of this image is written here as well, from the
    // token this step decided (or was given) -- one launch per step instead of two
    auto embed_next = [&](int tok) {
        if (x_next) {
            const float* er = emb + (size_t)tok * C;
            const float* pr = pos + (size_t)(step + 1) * C;
            for (int c = lane; c < C; c += kWave) x_next[(size_t)b * C + c] = er[c] + pr[c];
        }
    };
    if (!greedy) {
        for (int c = lane; c < Cc; c += kWave) o[c] = logits[(size_t)c * ls];
        embed_next(tokens[(size_t)b * Lt + step + 1]);
        return;
    }
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < Cc; c += kWave) {
        const float v = logits[(size_t)c * ls];
        if (v > mx) { mx = v; am = c; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(mx, off, kWave);
        const int oi = __shfl_xor(am, off, kWave);
        if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    float sum = 0.0f;
    for (int c = lane; c < Cc; c += kWave) sum += expf(logits[(size_t)c * ls] - mx);
    sum = wave_sum(sum);
    for (int c = lane; c < Cc; c += kWave) o[c] = expf(logits[(size_t)c * ls] - mx) / sum;
    if (lane == 0) tokens[(size_t)b * Lt + step + 1] = am;
    embed_next(am);
}

__global__ void __launch_bounds__(256)
dec_init_tokens_kernel(int* __restrict__ tokens, int Nb, int Lt, int start_idx, int pad_idx,
                       const int* __restrict__ forced, int Lf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb * Lt) return;
    const int b = i / Lt, p = i - b * Lt;
    if (forced) tokens[i] = p < Lf ? forced[(size_t)b * Lf + p] : pad_idx;
    else tokens[i] = p == 0 ? start_idx : pad_idx;
}

// ---- decoder, one step: the projections on the bf16 matrix cores with the three-term split ------------------------
// out (M, Co) = act(LN?(X) W + bias) [+ res], all TOKEN-major, X (M, K) fp32, M = images of the step.
// Why not the fp32 split-K kernel (conv1x1_skinny_f32_kernel) the exact-fp32 configuration keeps: its operands arrive
// as 4-byte loads (512 load instructions per 32x32 tile at 16 cycles each in the vector-memory unit: 3.7 us of an 8 us
// launch) and meet in 16 partial tiles of 64 KB.  Here
//   * the MFMA runs  D[co][token] = W^T[co][k] X^T[k][token]  so that a lane's B fragment is 8 consecutive k of ITS
//     token -- two 16-byte loads of the token-major row -- and its results are 4 consecutive outputs of its token
//     (16-byte stores);
//   * the weight arrives pre-split (hi = bf16(w), lo = bf16(w - hi)) and pre-arranged on the host in fragment order,
//     [Co/32][K/16][hi|lo][k half][32 outputs][8 k]: a fragment is one 16-byte load, a wavefront load 1 KB contiguous;
//   * x is split in registers (two v_cvt_pk_bf16_f32 and a subtraction per pair); a product is hi*hi + hi*lo + lo*hi
//     accumulated in fp32 (~5e-6 of a layer's scale, as in tpspp_conv2d_bf16_fwd's split3);
//   * four wavefronts split K (K/64 MFMA k-steps each, every load of a wavefront in flight together), four partial
//     tiles meet in 16 KB of LDS;
//   * LayerNorm folded as in tpspp_linear_ln_fwd: sum and sum of squares of the raw row ride on the same loads.
typedef __bf16 dbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned du32x4 __attribute__((ext_vector_type(4)));
typedef float df32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dbf16x2 __attribute__((ext_vector_type(2)));

struct DGemm {
    const float* X; const du32x4* Wp; const float* bias; const float* colsum; const float* res; float* out;
    int M, K, Co;            // Co: valid outputs (the arranged weight is padded to a multiple of 32)
    float eps; int act;      // act: 0 none, 2 GELU (erf)
    int remap = 0;           // dec_gemm_tile
};

__device__ __forceinline__ unsigned dpack2(float lo, float hi)
{
    df32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dbf16x2));
}

// Workgroup -> (token block, output tile) for the step GEMMs.  Workgroups go to the 8 XCDs round-robin in launch order and
// every XCD has its own L2, which each launch fills with the X rows (the previous launch's output) and the W tiles its
// workgroups touch.  In launch order an XCD gets 2 of the 16 token blocks and ALL output tiles; scripts/ubench/gemm_chain_bench
// (the projection's loads alone, 2000 dependent launches): launch order 5.03 us, 4 token blocks x half the tiles 4.77 us,
// 8 x a quarter 5.30 us, all 16 x an eighth 5.56 us -- X, freshly written elsewhere, costs more per byte than W.
// In the decoder (same box, interleaved): bf16 15.36 -> 15.19 ms, bf16x3 20.49 -> 20.42 ms, exact fp32 23.24 -> 23.30 (left
// in launch order).  Only the 16-token-block grids of batch 512 with an even number of tile rows are remapped
// (TPSPP_HEAD_PLAIN_ORDER=1: never).
__device__ __forceinline__ void dec_gemm_tile(int& bx, int& by, bool remap)
{
    bx = blockIdx.x; by = blockIdx.y;
    if (remap && gridDim.x == 16 && (gridDim.y & 1) == 0) {
        const int lin = blockIdx.x + 16 * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
        bx = (xcd & 3) * 4 + (slot & 3);
        by = (xcd >> 2) * (gridDim.y >> 1) + (slot >> 2);
    }
}

// NT: 32-output tiles per workgroup (1; 3 for the 1536-wide q|k|v projection: at one tile per workgroup its 768 workgroups
// ran as two rounds of latency-bound launches -- ~14 us against 6.5 us for the 512-wide projections -- while three tiles per
// workgroup are one round of 256 and read X once instead of three times; every tile is computed exactly as before).
template <int KSW, bool LN, int NT = 1>
__global__ void __launch_bounds__(256)
dec_gemm_x3_kernel(const DGemm P)
{
    __shared__ float sRed[4][16][kWave];
    __shared__ float sS1[8][32], sS2[8][32];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int bx, by;
    dec_gemm_tile(bx, by, P.remap != 0);
    const int m0 = bx * 32, ct0 = by * NT;
    const int KS = P.K >> 4;
    const int m = m0 + l31;
    const int mc = m < P.M ? m : P.M - 1;
    const int ntiles = (P.Co + 31) >> 5;
    const float4* xp = reinterpret_cast<const float4*>(P.X + (size_t)mc * P.K + 16 * (wv * KSW) + 8 * half);
    float4 xa[KSW][2];
    du32x4 ah[NT][KSW], al[NT][KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) { xa[j][0] = xp[4 * j]; xa[j][1] = xp[4 * j + 1]; }   // every load of the wavefront in flight together
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int ct = ct0 + tl < ntiles ? ct0 + tl : ntiles - 1;     // (a tile past the last one repeats it; never finished)
        const du32x4* wp = P.Wp + ((size_t)(ct * KS + wv * KSW) * 4 + half) * 32 + l31;
#pragma unroll
        for (int j = 0; j < KSW; ++j) { ah[tl][j] = wp[(size_t)j * 128]; al[tl][j] = wp[(size_t)j * 128 + 64]; }
    }
    // ... and the epilogue's operands with them (round 4): behind the reduction they were one more dependent round trip
    // to memory per launch, in a loop of ~2000 dependent launches
    bool mine[NT];
    size_t o[NT];
    float4 cs[NT], b4[NT], r4[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int c = (ct0 + tl) * 32 + 8 * wv + 4 * half;
        mine[tl] = m < P.M && c < P.Co;                    // (Co is a multiple of 4: a whole piece is in or out)
        o[tl] = (size_t)mc * P.Co + (c < P.Co ? c : 0);
        cs[tl] = make_float4(0.f, 0.f, 0.f, 0.f); b4[tl] = cs[tl]; r4[tl] = cs[tl];
        if (mine[tl]) {
            if (LN) cs[tl] = *reinterpret_cast<const float4*>(P.colsum + c);
            if (P.bias) b4[tl] = *reinterpret_cast<const float4*>(P.bias + c);
            if (P.res) r4[tl] = *reinterpret_cast<const float4*>(P.res + o[tl]);
        }
    }
    f32x16_t acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tl][i] = 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const float x[8] = {xa[j][0].x, xa[j][0].y, xa[j][0].z, xa[j][0].w, xa[j][1].x, xa[j][1].y, xa[j][1].z, xa[j][1].w};
        du32x4 bh, bl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned pk = dpack2(x[2 * q], x[2 * q + 1]);
            const float h0 = __builtin_bit_cast(float, pk << 16), h1 = __builtin_bit_cast(float, pk & 0xffff0000u);
            bh[q] = pk;
            bl[q] = dpack2(x[2 * q] - h0, x[2 * q + 1] - h1);
            if (LN) {
                s1 += x[2 * q] + x[2 * q + 1];
                s2 = fmaf(x[2 * q], x[2 * q], s2);
                s2 = fmaf(x[2 * q + 1], x[2 * q + 1], s2);
            }
        }
        const dbf16x8 Bh = __builtin_bit_cast(dbf16x8, bh), Bl = __builtin_bit_cast(dbf16x8, bl);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const dbf16x8 Ah = __builtin_bit_cast(dbf16x8, ah[tl][j]), Al = __builtin_bit_cast(dbf16x8, al[tl][j]);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[tl], 0, 0, 0);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[tl], 0, 0, 0);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[tl], 0, 0, 0);
        }
    }
    if (LN) { sS1[wv * 2 + half][l31] = s1; sS2[wv * 2 + half][l31] = s2; }
    float mean = 0.0f, rstd = 1.0f;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        if (tl > 0) __syncthreads();                       // the previous tile's partial sums have been read
#pragma unroll
        for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[tl][r];
        __syncthreads();
        // wavefront w finishes accumulator registers 4 w .. 4 w + 3: outputs 32 ct + 8 w + 4 half + (0 .. 3) of token l31
        if (LN && tl == 0) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { t1 += sS1[j][l31]; t2 += sS2[j][l31]; }
            mean = t1 / (float)P.K;
            rstd = 1.0f / sqrtf(fmaxf(t2 / (float)P.K - mean * mean, 0.0f) + P.eps);
        }
        if (!mine[tl]) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = (sRed[0][4 * wv + e][lane] + sRed[1][4 * wv + e][lane]) + (sRed[2][4 * wv + e][lane] + sRed[3][4 * wv + e][lane]);
        if (LN) {
            v[0] = rstd * (v[0] - mean * cs[tl].x); v[1] = rstd * (v[1] - mean * cs[tl].y);
            v[2] = rstd * (v[2] - mean * cs[tl].z); v[3] = rstd * (v[3] - mean * cs[tl].w);
        }
        if (P.bias) { v[0] += b4[tl].x; v[1] += b4[tl].y; v[2] += b4[tl].z; v[3] += b4[tl].w; }
        if (P.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        }
        if (P.res) { v[0] += r4[tl].x; v[1] += r4[tl].y; v[2] += r4[tl].z; v[3] += r4[tl].w; }
        *reinterpret_cast<float4*>(P.out + o[tl]) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- q projection + cross-attention of a layer-step in ONE launch (round 4; the review's item 3) ------------------------------
// A workgroup owns 16 images x one head: (A) their 64 q features = LN(y) Wq + b exactly as dec_gemm_x3_kernel<8, true>
// computes them -- the same loads, k split, products, reduction order and epilogue, for the head's two 32-output tiles, the
// 16 tokens in the lower half of the MFMA's columns -- left in LDS; (B) each of the four wavefronts then runs cross_attend for
// four of the sixteen (image, head) pairs.  Bit for bit the two launches it replaces.  Grid (Nb / 16, H) = 256 workgroups at
// batch 512: the 128 KB of the head's weight slice is read once per 16 images as before (33 MB of L2 reads per launch).
// MEASURED (batch 512, 40 steps, scripts/debug/dec_fusion_ab.py, interleaved runs on one box): the greedy decoder takes
// 24.9 ms with it against 21.8 ms with the two launches (bf16x3 head), 18.05 against 16.65 ms (bf16 head): +13 / +6 us per
// layer-step.  The separate attention launch runs 16 wavefronts per CU, one (image, head) each, and is bound by the keys' and
// values' bytes; here four wavefronts per CU walk four pairs each, one after the other, behind the projection's own load ->
// product -> reduce chain, and what the fusion removes -- one kernel boundary, ~1.5-2 us -- is less than what the lost
// parallelism costs.  Hence opt-in only (TPSPP_HEAD_QCROSS=1), kept as the measured form of the review's proposal.
template <typename KV>
__global__ void __launch_bounds__(256)
dec_q_cross_x3_kernel(const DGemm P, const KV* __restrict__ Kx_t, const KV* __restrict__ Vx_t, int H, int T,
                      const int* __restrict__ valid_len, float* __restrict__ out)
{
    constexpr int KSW = 8;
    __shared__ float sRed[4][2][16][kWave];
    __shared__ float sS1[8][32], sS2[8][32];
    __shared__ __attribute__((aligned(16))) float sQ[16][kDK];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31, l15 = lane & 15;
    const int m0 = blockIdx.x * 16, h = blockIdx.y;
    const int KS = P.K >> 4;
    const int m = m0 + l15;
    const int mc = m < P.M ? m : P.M - 1;
    const float4* xp = reinterpret_cast<const float4*>(P.X + (size_t)mc * P.K + 16 * (wv * KSW) + 8 * half);
    float4 xa[KSW][2];
    du32x4 ah[2][KSW], al[2][KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) { xa[j][0] = xp[4 * j]; xa[j][1] = xp[4 * j + 1]; }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const du32x4* wp = P.Wp + ((size_t)((2 * h + tl) * KS + wv * KSW) * 4 + half) * 32 + l31;
#pragma unroll
        for (int j = 0; j < KSW; ++j) { ah[tl][j] = wp[(size_t)j * 128]; al[tl][j] = wp[(size_t)j * 128 + 64]; }
    }
    const int c = 8 * wv + 4 * half;                       // + 32 tl: this lane's four q features of the head (epilogue)
    float4 cs[2], b4[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        cs[tl] = *reinterpret_cast<const float4*>(P.colsum + kDK * h + 32 * tl + c);
        b4[tl] = P.bias ? *reinterpret_cast<const float4*>(P.bias + kDK * h + 32 * tl + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x16_t acc[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tl][i] = 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const float x[8] = {xa[j][0].x, xa[j][0].y, xa[j][0].z, xa[j][0].w, xa[j][1].x, xa[j][1].y, xa[j][1].z, xa[j][1].w};
        du32x4 bh, bl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned pk = dpack2(x[2 * q], x[2 * q + 1]);
            const float h0 = __builtin_bit_cast(float, pk << 16), h1 = __builtin_bit_cast(float, pk & 0xffff0000u);
            bh[q] = pk;
            bl[q] = dpack2(x[2 * q] - h0, x[2 * q + 1] - h1);
            s1 += x[2 * q] + x[2 * q + 1];
            s2 = fmaf(x[2 * q], x[2 * q], s2);
            s2 = fmaf(x[2 * q + 1], x[2 * q + 1], s2);
        }
        const dbf16x8 Bh = __builtin_bit_cast(dbf16x8, bh), Bl = __builtin_bit_cast(dbf16x8, bl);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const dbf16x8 Ah = __builtin_bit_cast(dbf16x8, ah[tl][j]), Al = __builtin_bit_cast(dbf16x8, al[tl][j]);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[tl], 0, 0, 0);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[tl], 0, 0, 0);
            acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[tl], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) sRed[wv][tl][r][lane] = acc[tl][r];
    sS1[wv * 2 + half][l31] = s1; sS2[wv * 2 + half][l31] = s2;
    __syncthreads();
    if (l31 < 16) {
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { t1 += sS1[j][l31]; t2 += sS2[j][l31]; }
        const float mean = t1 / (float)P.K;
        const float rstd = 1.0f / sqrtf(fmaxf(t2 / (float)P.K - mean * mean, 0.0f) + P.eps);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = (sRed[0][tl][4 * wv + e][lane] + sRed[1][tl][4 * wv + e][lane]) +
                       (sRed[2][tl][4 * wv + e][lane] + sRed[3][tl][4 * wv + e][lane]);
            v[0] = rstd * (v[0] - mean * cs[tl].x); v[1] = rstd * (v[1] - mean * cs[tl].y);
            v[2] = rstd * (v[2] - mean * cs[tl].z); v[3] = rstd * (v[3] - mean * cs[tl].w);
            if (P.bias) { v[0] += b4[tl].x; v[1] += b4[tl].y; v[2] += b4[tl].z; v[3] += b4[tl].w; }
            *reinterpret_cast<float4*>(&sQ[l31][32 * tl + c]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    constexpr int EPL = Wide<KV>::EPL, GS = kDK / EPL;
    const int dl = lane % GS;
    for (int t = wv; t < 16; t += 4) {
        const int b = m0 + t;
        if (b >= P.M) break;                               // wave-uniform
        float q[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = sQ[t][EPL * dl + e] * 0.125f;
        int nvalid = valid_len ? valid_len[b] : T;
        nvalid = nvalid < T ? nvalid : T;
        cross_attend<KV>(q, Kx_t, Vx_t, P.Co, P.M, T, nvalid, b, h, lane, out, 0);
    }
}

#include "tpspp_head_persist.h"

// The same GEMM with exact fp32 products (v_mfma_f32_32x32x2_f32) for the exact-fp32 configuration: the weight arrives
// as fp32 in fragment order, [Co/32][K/8][k half][32 outputs][4 k] with k = 8 u + 4 half + e, x as 16-byte pieces of the
// token-major row (the lane's four k of the step), four MFMAs per 16-byte pair; eight wavefronts split K.
struct DGemmF {
    const float* X; const float4* Wp; const float* bias; const float* colsum; const float* res; float* out;
    int M, K, Co; float eps; int act;
    int remap = 0;
};
template <int KUW, bool LN, int NT = 1>
__global__ void __launch_bounds__(512)
dec_gemm_f32_kernel(const DGemmF P)
{
    __shared__ float sRed[8][16][kWave];
    __shared__ float sS1[16][32], sS2[16][32];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int bx, by;
    dec_gemm_tile(bx, by, P.remap != 0);
    const int m0 = bx * 32, ct0 = by * NT;
    const int KU = P.K >> 3;
    const int m = m0 + l31;
    const int mc = m < P.M ? m : P.M - 1;
    const int ntiles = (P.Co + 31) >> 5;
    const float4* xp = reinterpret_cast<const float4*>(P.X + (size_t)mc * P.K + 8 * (wv * KUW) + 4 * half);
    float4 xa[KUW], wa[NT][KUW];
#pragma unroll
    for (int j = 0; j < KUW; ++j) xa[j] = xp[2 * j];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int ct = ct0 + tl < ntiles ? ct0 + tl : ntiles - 1;
        const float4* wp = P.Wp + ((size_t)(ct * KU + wv * KUW) * 2 + half) * 32 + l31;
#pragma unroll
        for (int j = 0; j < KUW; ++j) wa[tl][j] = wp[(size_t)j * 64];
    }
    // the epilogue's operands ride with them (wavefronts 0 - 3 finish the tile; see dec_gemm_x3_kernel)
    bool mine[NT];
    size_t o[NT];
    float4 cs[NT], b4[NT], r4[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int c = (ct0 + tl) * 32 + 8 * (wv & 3) + 4 * half;
        mine[tl] = wv < 4 && m < P.M && c < P.Co;
        o[tl] = (size_t)mc * P.Co + (c < P.Co ? c : 0);
        cs[tl] = make_float4(0.f, 0.f, 0.f, 0.f); b4[tl] = cs[tl]; r4[tl] = cs[tl];
        if (mine[tl]) {
            if (LN) cs[tl] = *reinterpret_cast<const float4*>(P.colsum + c);
            if (P.bias) b4[tl] = *reinterpret_cast<const float4*>(P.bias + c);
            if (P.res) r4[tl] = *reinterpret_cast<const float4*>(P.res + o[tl]);
        }
    }
    f32x16_t acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tl][i] = 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < KUW; ++j) {
        const float x[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (LN) { s1 += x[e]; s2 = fmaf(x[e], x[e], s2); }
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) {
                const float w = e == 0 ? wa[tl][j].x : e == 1 ? wa[tl][j].y : e == 2 ? wa[tl][j].z : wa[tl][j].w;
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x[e], acc[tl], 0, 0, 0);
            }
        }
    }
    if (LN) { sS1[wv * 2 + half][l31] = s1; sS2[wv * 2 + half][l31] = s2; }
    float mean = 0.0f, rstd = 1.0f;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        if (tl > 0) __syncthreads();                       // the previous tile's partial sums have been read
#pragma unroll
        for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[tl][r];
        __syncthreads();
        // wavefronts 0 - 3 finish accumulator registers 4 w .. 4 w + 3: outputs 32 ct + 8 w + 4 half + (0 .. 3) of token l31
        if (LN && tl == 0) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { t1 += sS1[j][l31]; t2 += sS2[j][l31]; }
            mean = t1 / (float)P.K;
            rstd = 1.0f / sqrtf(fmaxf(t2 / (float)P.K - mean * mean, 0.0f) + P.eps);
        }
        if (!mine[tl]) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = 0.0f;
#pragma unroll
            for (int pz = 0; pz < 8; pz += 2) t += sRed[pz][4 * wv + e][lane] + sRed[pz + 1][4 * wv + e][lane];
            v[e] = t;
        }
        if (LN) {
            v[0] = rstd * (v[0] - mean * cs[tl].x); v[1] = rstd * (v[1] - mean * cs[tl].y);
            v[2] = rstd * (v[2] - mean * cs[tl].z); v[3] = rstd * (v[3] - mean * cs[tl].w);
        }
        if (P.bias) { v[0] += b4[tl].x; v[1] += b4[tl].y; v[2] += b4[tl].z; v[3] += b4[tl].w; }
        if (P.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        }
        if (P.res) { v[0] += r4[tl].x; v[1] += r4[tl].y; v[2] += r4[tl].z; v[3] += r4[tl].w; }
        *reinterpret_cast<float4*>(P.out + o[tl]) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// (TPSPP_HEAD_NARROW_QKV=1: one 32-output tile per workgroup for the q|k|v projection as well, as before round 4)
const bool g_head_narrow_qkv = getenv("TPSPP_HEAD_NARROW_QKV") != nullptr;
const bool g_head_plain_order = getenv("TPSPP_HEAD_PLAIN_ORDER") != nullptr;

// launches the step GEMM; false when the shape has no instantiation (K must be 256 or 512, Co a multiple of 4)
bool dec_gemm_x3(hipStream_t st, const float* X, const void* Wp, const float* bias, const float* colsum, float eps,
                 const float* res, int act, int M, int K, int Co, float* out, bool f32 = false)
{
    if ((K != 256 && K != 512) || (Co & 3) || M <= 0) return false;
    const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((Co + 31) / 32));
    // the q|k|v projection (1536 outputs, folded LayerNorm) of the three-term form: three tiles per workgroup, one round of
    // 256 workgroups instead of 768 in two (bf16x3 decoder 20.85 -> 20.45 ms, bf16 15.68 -> 15.33, bit-identical scores,
    // scripts/debug/dec_fusion_ab.py TPSPP_HEAD_NARROW_QKV).  Not the exact-fp32 form: its 96 fp32 matrix instructions per
    // wavefront are the launch's time either way (23.40 -> 23.72 ms with three tiles).
    if (!f32 && K == 512 && colsum && Co >= 1024 && !g_head_narrow_qkv) {
        const dim3 grid3((unsigned)((M + 31) / 32), (unsigned)(((Co + 31) / 32 + 2) / 3));
        DGemm P;
        P.X = X; P.Wp = reinterpret_cast<const du32x4*>(Wp); P.bias = bias; P.colsum = colsum; P.res = res; P.out = out;
        P.M = M; P.K = K; P.Co = Co; P.eps = eps; P.act = act; P.remap = g_head_plain_order ? 0 : 1;
        hipLaunchKernelGGL((dec_gemm_x3_kernel<8, true, 3>), grid3, dim3(256), 0, st, P);
        return true;
    }
    if (f32) {
        DGemmF P;
        P.X = X; P.Wp = reinterpret_cast<const float4*>(Wp); P.bias = bias; P.colsum = colsum; P.res = res; P.out = out;
        P.M = M; P.K = K; P.Co = Co; P.eps = eps; P.act = act; P.remap = 0;      // (measured: nothing for the exact-fp32 form)
        if (K == 512) {
            if (colsum) hipLaunchKernelGGL((dec_gemm_f32_kernel<8, true>), grid, dim3(512), 0, st, P);
            else        hipLaunchKernelGGL((dec_gemm_f32_kernel<8, false>), grid, dim3(512), 0, st, P);
        } else {
            if (colsum) hipLaunchKernelGGL((dec_gemm_f32_kernel<4, true>), grid, dim3(512), 0, st, P);
            else        hipLaunchKernelGGL((dec_gemm_f32_kernel<4, false>), grid, dim3(512), 0, st, P);
        }
        return true;
    }
    DGemm P;
    P.X = X; P.Wp = reinterpret_cast<const du32x4*>(Wp); P.bias = bias; P.colsum = colsum; P.res = res; P.out = out;
    P.M = M; P.K = K; P.Co = Co; P.eps = eps; P.act = act; P.remap = g_head_plain_order ? 0 : 1;
    if (K == 512) {
        if (colsum) hipLaunchKernelGGL((dec_gemm_x3_kernel<8, true>), grid, dim3(256), 0, st, P);
        else        hipLaunchKernelGGL((dec_gemm_x3_kernel<8, false>), grid, dim3(256), 0, st, P);
    } else {
        if (colsum) hipLaunchKernelGGL((dec_gemm_x3_kernel<4, true>), grid, dim3(256), 0, st, P);
        else        hipLaunchKernelGGL((dec_gemm_x3_kernel<4, false>), grid, dim3(256), 0, st, P);
    }
    return true;
}

// ---- host side ---------------------------------------------------------------------------------------------------
// (rows, cols) bf16 -> (cols, rows) bf16, 64x64 tiles through LDS
__global__ void __launch_bounds__(256)
transpose2d_b16_kernel(const unsigned short* __restrict__ in, int rows, int cols, unsigned short* __restrict__ out)
{
    __shared__ unsigned short tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][i];
    }
}

// lab switch (TPSPP_HEAD_NO_TOKGEMM=1 in the environment): the wide projections through the convolution kernel as before
const bool g_head_no_tokgemm = getenv("TPSPP_HEAD_NO_TOKGEMM") != nullptr;
// TPSPP_HEAD_QCROSS=1: the q projection and the cross-attention of a decoder layer-step as ONE launch (dec_q_cross_x3_kernel).
// Off by default: bit-identical and slower (see the kernel's header).
const bool g_head_qcross = getenv("TPSPP_HEAD_QCROSS") != nullptr;
// TPSPP_HEAD_NO_PERSIST=1: the reduced-precision step pipeline as ~50 launches per step (rounds 3-4) instead of ONE persistent
// launch per step (tpspp_head_persist.h); bit-identical scores -- for A/B runs and the bit-identity test
// TPSPP_HEAD_CROSS1=1: the cross-attention with one head per wavefront as before round 5 (A/B runs)
const bool g_head_cross1 = getenv("TPSPP_HEAD_CROSS1") != nullptr;
bool head_no_persist() { return getenv("TPSPP_HEAD_NO_PERSIST") != nullptr; }
long long* g_head_trace = nullptr;     // tpspp_head_set_trace

struct Gemm {
    hipStream_t st;
    int rc = 0;
    // the same product on the bf16 matrix cores: W16 = the weight arranged for tpspp_conv2d_bf16_fwd (1x1),
    // X / res fp32 (rounded to bf16 as they are staged), out fp32 or bf16
    // split3: the "bf16x3" three-term split on the fp32 operands (W16 then holds hi and lo slabs)
    void cm16(const void* W16, const float* bias, const float* X, int K, int Co, int M, void* out, int out_f32, int act,
              const float* res, int split3 = 0)
    {
        if (rc) return;
        // the token GEMM (tpspp_tokgemm.hip) where the shape qualifies, else the product as a 1x1 convolution
        tpspp::TokGemmArgs ta;
        ta.X = X; ta.W = W16; ta.bias = bias; ta.res = res; ta.out = out; ta.out_f32 = out_f32;
        ta.K = K; ta.Co = Co; ta.M = M; ta.act = act; ta.x3 = split3; ta.kgc = tpspp_conv_bf16_chunk_channels(1) / 8;
        if (!g_head_no_tokgemm && tpspp::tok_gemm_applicable(ta)) {
            tpspp::launch_tok_gemm(ta, st);
            rc = tpspp::check_launch("token GEMM");
            return;
        }
        const void* src[1] = {X};
        const int dims[6] = {K, 1, M, 1, 1, 1};
        rc = tpspp_conv2d_bf16_fwd(src, dims, 1, W16, bias, res, 1, nullptr, nullptr, res ? 1 : 0, act, 1, Co, 1, 1, 1, 1,
                                   out, out_f32, 1, M, split3, st);
    }
    // out (Co, M) = act(W^T X + bias) [+ res]      W (K, Co) k-major, X (K, M) channel-major
    void cm(const float* W, const float* bias, const float* X, int K, int Co, int M, float* out, int act,
            const float* res)
    {
        if (rc) return;
        const float* src[1] = {X};
        const int dims[5] = {K, 1, M, 1, 1};
        rc = tpspp_conv2d_fwd(src, dims, 1, W, (K % 32 == 0) ? W : nullptr, bias, res, nullptr, nullptr,
                              res ? 1 : 0, act, 1, Co, 1, 1, 1, 1, out, 1, M, st);
    }
    // out (M, Co) token-major = X^T W           (operands swapped: X plays the weights, W the image)
    void tm(const float* W, const float* X, int K, int Co, int M, float* out)
    {
        if (rc) return;
        const float* src[1] = {W};
        const int dims[5] = {K, 1, Co, 1, 1};
        rc = tpspp_conv2d_fwd(src, dims, 1, X, (K % 32 == 0) ? X : nullptr, nullptr, nullptr, nullptr, nullptr,
                              0, 0, 1, M, 1, 1, 1, 1, out, 1, Co, st);
    }
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {
    char* p;
    size_t left;
    bool ok = true;
    float* f(size_t n)
    {
        const size_t bytes = align256(n * sizeof(float));
        if (bytes > left) { ok = false; return nullptr; }
        float* r = reinterpret_cast<float*>(p);
        p += bytes; left -= bytes;
        return r;
    }
};

size_t enc_ws_bytes(int N, int C, int T, int Di)
{
    const size_t M = (size_t)N * T;
    return align256((size_t)C * M * 4) * 3 + align256((size_t)3 * C * M * 4) + align256((size_t)Di * M * 4);
}

size_t dec_ws_bytes(int N, int C, int T, int Di, int n_layers, int L, int Cc)
{
    const size_t MT = (size_t)N * T;
    size_t s = 0;
    s += (size_t)n_layers * 2 * align256((size_t)C * MT * 4);                // Kx, Vx_t per layer
    s += (size_t)n_layers * 2 * align256((size_t)N * C * (size_t)L * 4);      // self-attention caches
    s += 3 * align256((size_t)C * N * 4);                                    // x, y, a
    s += align256((size_t)3 * C * N * 4);                                    // qkv_t
    s += align256((size_t)Di * N * 4);                                       // hidden
    s += align256((size_t)Cc * N * 4);                                       // logits
    s += align256((size_t)N * (L + 1) * 4);                                  // tokens
    s += align256((size_t)C * MT * 4);                                       // keys before their transposition (token-major step pipeline)
    s += align256(((size_t)(N + 31) / 32 * 32 + 64) * 4);                    // the persistent step kernel's cluster counters (128 B apart) + error flag
    return s;
}


// ---- the persistent decode's two requirements (include/tpspp.h, tpspp_nrtr_decoder_fwd) -----------------------------------
// (1) Co-residency.  The 16 workgroups of a cluster spin on each other's counter, and blocks are dispatched in order: a launch
//     makes progress iff a whole GROUP of 8 clusters (128 consecutive blocks) can be resident.  persist_groups() = how many such
//     groups the device holds at once (CU count x the kernel's occupancy at 512 threads + sizeof(PShared) of LDS, / 128), capped
//     at 2 (512 images per launch); 0 -- a smaller part, a CU-masked or CPX partition, a failed attribute call -- sends the
//     decode down the launch-per-phase pipeline.
// (2) One persistent decode in flight per device.  Two of them on two streams are harmless by themselves (in-order dispatch:
//     one of them always owns a whole group), three can starve each other until the timeout; instead of reasoning about the
//     dispatcher, every persistent decode waits (hipStreamWaitEvent, no host synchronisation) for the event recorded behind
//     the previous one on this device, whatever its stream, and records its own.  Decodes of one process therefore never
//     overlap each other on a device; kernels of other streams may overlap them freely (they finish on their own, the barrier
//     timeout is wall-clock time).  Other PROCESSES on the same device are not seen by this guard: one process per GPU.
struct PersistDevice {
    hipEvent_t done = nullptr;
    bool recorded = false;
    int groups[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};      // per kernel instantiation; -1 = not queried yet
};
std::mutex g_persist_mu;
PersistDevice g_persist_dev[tpspp::kMaxDevices];

template <typename Kern>
int persist_groups(PersistDevice& D, int which, Kern kern, int dev)
{
    if (D.groups[which] < 0) {
        int groups = 0, cus = 0, per_cu = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PShared)) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, sizeof(PShared)) == hipSuccess) {
            groups = (int)(((long)cus * per_cu) / 128);
            if (groups > 2) groups = 2;
        }
        (void)hipGetLastError();
        D.groups[which] = groups;
    }
    int groups = D.groups[which];
    if (const char* f = getenv("TPSPP_HEAD_PERSIST_GROUPS")) {   // lab / test switch (read per call): pretend the device holds fewer groups
        const int v = atoi(f);
        if (v >= 0 && v < groups) groups = v;
    }
    return groups;
}

// lab / test kernel (tpspp_lab_occupy): `blocks` workgroups that hold `lds` bytes of LDS each and spin on the wall clock
__global__ void __launch_bounds__(64) lab_occupy_kernel(long long ticks, int* sink)
{
    extern __shared__ int occ_smem[];
    const long long t0 = (long long)wall_clock64();
    int n = 0;
    while ((long long)wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++n; }
    if (n < 0) { occ_smem[threadIdx.x] = n; sink[0] = occ_smem[0]; }       // (never: keeps the LDS allocation alive)
}

}  // namespace

TPSPP_EXPORT int tpspp_head_set_trace(long long* device_buf)
{
    g_head_trace = device_buf;
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_lab_occupy(int workgroups, int lds_bytes, int milliseconds, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(workgroups > 0 && workgroups <= 4096 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && milliseconds > 0 && milliseconds <= 2000,
                  "tpspp_lab_occupy: workgroups in [1, 4096], lds_bytes in [0, 160 KB], milliseconds in [1, 2000]");
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(lab_occupy_kernel, dim3((unsigned)workgroups), dim3(64), (size_t)lds_bytes, tpspp::as_stream(stream),
                       (long long)milliseconds * 100000, (int*)nullptr);
    return tpspp::check_launch("tpspp_lab_occupy");
}

TPSPP_EXPORT size_t tpspp_nrtr_encoder_workspace(int N, int C, int T, int d_inner)
{
    if (N <= 0 || C <= 0 || T <= 0 || d_inner <= 0) return 0;
    return enc_ws_bytes(N, C, T, d_inner);
}

TPSPP_EXPORT size_t tpspp_nrtr_decoder_workspace(int N, int C, int T, int d_inner, int n_layers, int max_seq_len,
                                                 int num_out)
{
    if (N <= 0 || C <= 0 || T <= 0 || d_inner <= 0 || n_layers <= 0 || max_seq_len <= 0 || num_out <= 0) return 0;
    return dec_ws_bytes(N, C, T, d_inner, n_layers, max_seq_len, num_out);
}

TPSPP_EXPORT int tpspp_transpose2d(const float* in, int rows, int cols, float* out, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && out && rows > 0 && cols > 0, "tpspp_transpose2d: bad argument");
    const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    TPSPP_REQUIRE(grid.y <= 65535, "tpspp_transpose2d: too many rows");
    hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, tpspp::as_stream(stream), in, rows, cols, out);
    return tpspp::check_launch("tpspp_transpose2d");
}

TPSPP_EXPORT int tpspp_layernorm_cm_fwd(const float* x, const float* gamma, const float* beta, int C, int M,
                                        float eps, float* y, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(x && gamma && beta && y && C > 0 && M > 0, "tpspp_layernorm_cm_fwd: bad argument");
    TPSPP_REQUIRE(C <= 1024, "tpspp_layernorm_cm_fwd: at most 1024 features");
    const dim3 grid((unsigned)((M + 15) / 16)), block(256);
    hipStream_t st = tpspp::as_stream(stream);
    if (C <= 128)      hipLaunchKernelGGL(layernorm_cm_kernel<8>, grid, block, 0, st, x, gamma, beta, C, M, eps, y);
    else if (C <= 512) hipLaunchKernelGGL(layernorm_cm_kernel<32>, grid, block, 0, st, x, gamma, beta, C, M, eps, y);
    else               hipLaunchKernelGGL(layernorm_cm_kernel<64>, grid, block, 0, st, x, gamma, beta, C, M, eps, y);
    return tpspp::check_launch("tpspp_layernorm_cm_fwd");
}

static int launch_attn_enc(const float* qkv, int N, int C, int T, const int* valid_len, float* out, hipStream_t st)
{
    const int H = C / kDK;
    const size_t M = (size_t)N * T;
    const size_t lds = (size_t)2 * T * kKRow * sizeof(float);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_enc_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    if (T <= 128) {                                          // matrix-core kernel: a wavefront per 32 queries
        const int nj = (T + 31) / 32;
        const long waves = (long)N * H * nj;
        const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
        switch (nj) {
        case 1: hipLaunchKernelGGL(attn_enc_mfma_kernel<1>, grid, block, 0, st, qkv, C, (int)M, T, valid_len, out); break;
        case 2: hipLaunchKernelGGL(attn_enc_mfma_kernel<2>, grid, block, 0, st, qkv, C, (int)M, T, valid_len, out); break;
        case 3: hipLaunchKernelGGL(attn_enc_mfma_kernel<3>, grid, block, 0, st, qkv, C, (int)M, T, valid_len, out); break;
        default: hipLaunchKernelGGL(attn_enc_mfma_kernel<4>, grid, block, 0, st, qkv, C, (int)M, T, valid_len, out); break;
        }
        return tpspp::check_launch("attn_enc_mfma_kernel");
    }
    const int threads = kWave * ((T + kWave - 1) / kWave < 4 ? (T + kWave - 1) / kWave : 4);
    const dim3 grid((unsigned)((T + threads - 1) / threads), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL(attn_enc_kernel, grid, dim3(threads), lds, st, qkv, C, (int)M, T, valid_len, out);
    return tpspp::check_launch("attn_enc_kernel");
}

TPSPP_EXPORT int tpspp_attn_enc_fwd(const float* qkv, int N, int C, int T, const int* valid_len, float* out,
                                    tpspp_stream_t stream)
{
    TPSPP_REQUIRE(qkv && out && N > 0 && T > 0 && C > 0 && C % kDK == 0, "tpspp_attn_enc_fwd: bad argument (C must be a multiple of 64)");
    TPSPP_REQUIRE(T <= 256, "tpspp_attn_enc_fwd: at most 256 tokens per image");
    TPSPP_REQUIRE(N <= 65535 && (size_t)N * T < (1u << 31), "tpspp_attn_enc_fwd: batch too large");
    return launch_attn_enc(qkv, N, C, T, valid_len, out, tpspp::as_stream(stream));
}

// layer pointer tables: see include/tpspp.h
enum { E_LN1G, E_LN1B, E_WQKV, E_BQKV, E_WFC, E_BFC, E_LN2G, E_LN2B, E_W1, E_B1, E_W2, E_B2, E_COUNT };
// decoder: the three LayerNorms are folded into the projections that follow them (tpspp_linear_ln_fwd)
enum { D_QKV_W, D_QKV_CS, D_QKV_B, D_WFC, D_BFC, D_Q_W, D_Q_CS, D_Q_B, D_WK, D_BK, D_WV, D_WFC2, D_BFC2, D_W1_W, D_W1_CS,
       D_W1_B, D_W2, D_B2,
       // the six per-step projections arranged for the step GEMM: split hi / lo bf16 (TPSPP_HEAD_BF16 / _BF16X3) or fp32
       D_QKV_X, D_WFC_X, D_Q_X, D_WFC2_X, D_W1_X, D_W2_X, D_COUNT };

TPSPP_EXPORT int tpspp_nrtr_encoder_fwd(const float* feat, int N, int C, int T, int d_inner, int n_layers,
                                        const float* const* layer_ptrs, const float* ln_g, const float* ln_b,
                                        const int* valid_len, void* workspace, size_t workspace_bytes,
                                        float* out_cm, float* out_ntc, int flags, tpspp_stream_t stream)
{
    const int x3 = (flags & TPSPP_HEAD_BF16X3) ? 1 : 0;
    const bool b16 = (flags & (TPSPP_HEAD_BF16 | TPSPP_HEAD_BF16X3)) != 0;
    TPSPP_REQUIRE(feat && layer_ptrs && ln_g && ln_b && workspace && (out_cm || out_ntc),
                  "tpspp_nrtr_encoder_fwd: null pointer");
    TPSPP_REQUIRE(N > 0 && T > 0 && n_layers > 0 && d_inner > 0 && C > 0 && C % kDK == 0,
                  "tpspp_nrtr_encoder_fwd: bad sizes (d_model must be a multiple of 64 = n_head * 64)");
    TPSPP_REQUIRE(T <= 256, "tpspp_nrtr_encoder_fwd: at most 256 tokens per image");
    TPSPP_REQUIRE(N <= 65535 && (size_t)N * T * 3 * C < ((size_t)1 << 40), "tpspp_nrtr_encoder_fwd: batch too large");
    TPSPP_REQUIRE(workspace_bytes >= enc_ws_bytes(N, C, T, d_inner),
                  "tpspp_nrtr_encoder_fwd: workspace too small (%zu < %zu)", workspace_bytes,
                  enc_ws_bytes(N, C, T, d_inner));
    for (int l = 0; l < n_layers; ++l) {
        const float* const* w = layer_ptrs + (size_t)l * E_COUNT;
        TPSPP_REQUIRE(w[E_LN1G] && w[E_LN1B] && w[E_WQKV] && w[E_WFC] && w[E_LN2G] && w[E_LN2B] && w[E_W1] && w[E_W2],
                      "tpspp_nrtr_encoder_fwd: layer %d has a null weight", l);
    }
    hipStream_t st = tpspp::as_stream(stream);
    const int M = N * T;
    Carver cv{reinterpret_cast<char*>(workspace), workspace_bytes};
    float* x = cv.f((size_t)C * M);
    float* y = cv.f((size_t)C * M);
    float* a = cv.f((size_t)C * M);
    float* qkv = cv.f((size_t)3 * C * M);
    float* hid = cv.f((size_t)d_inner * M);
    TPSPP_REQUIRE(cv.ok, "tpspp_nrtr_encoder_fwd: workspace carve failed");

    {
        const size_t total = (size_t)N * C * T;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
        hipLaunchKernelGGL(nct_to_cm_kernel, dim3(blocks), dim3(256), 0, st, feat, N, C, T, x);
    }
    Gemm g{st};
    int rc = 0;
    for (int l = 0; l < n_layers && !rc && !g.rc; ++l) {
        const float* const* w = layer_ptrs + (size_t)l * E_COUNT;
        // x = x + fc(attn(LN1(x)))                                   transformer_layers.py:67-70
        rc = tpspp_layernorm_cm_fwd(x, w[E_LN1G], w[E_LN1B], C, M, 1e-5f, y, stream);
        if (rc) break;
        if (b16) g.cm16(w[E_WQKV], w[E_BQKV], y, C, 3 * C, M, qkv, 1, 0, nullptr, x3);
        else g.cm(w[E_WQKV], w[E_BQKV], y, C, 3 * C, M, qkv, 0, nullptr);
        if (g.rc) break;
        rc = launch_attn_enc(qkv, N, C, T, valid_len, a, st);
        if (rc) break;
        if (b16) g.cm16(w[E_WFC], w[E_BFC], a, C, C, M, y, 1, 0, x, x3);
        else g.cm(w[E_WFC], w[E_BFC], a, C, C, M, y, 0, x);          // y = x + fc(a)
        // x = y + w2(gelu(w1(LN2(y))))                                transformer_layers.py:72-75
        rc = tpspp_layernorm_cm_fwd(y, w[E_LN2G], w[E_LN2B], C, M, 1e-5f, a, stream);
        if (rc) break;
        if (b16) {
            g.cm16(w[E_W1], w[E_B1], a, C, d_inner, M, hid, 1, 2, nullptr, x3);
            g.cm16(w[E_W2], w[E_B2], hid, d_inner, C, M, x, 1, 0, y, x3);
        } else {
            g.cm(w[E_W1], w[E_B1], a, C, d_inner, M, hid, 2, nullptr);
            g.cm(w[E_W2], w[E_B2], hid, d_inner, C, M, x, 0, y);
        }
    }
    if (rc) return rc;
    if (g.rc) return g.rc;
    float* fin = out_cm ? out_cm : y;
    rc = tpspp_layernorm_cm_fwd(x, ln_g, ln_b, C, M, 1e-5f, fin, stream);    // nrtr_encoder.py:85
    if (rc) return rc;
    if (out_ntc) return tpspp_transpose2d(fin, C, M, out_ntc, stream);        // (C, N*T) -> (N, T, C)
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_nrtr_decoder_fwd(const float* enc_cm, int N, int C, int T, int d_inner, int n_layers,
                                        const float* const* layer_ptrs, int layer_ptrs_len,
                                        const float* emb, const float* pos_table, int n_position,
                                        const float* w_cls, const float* cls_colsum, const float* b_cls, int num_out,
                                        int max_seq_len,
                                        int start_idx, int padding_idx, const int* valid_len,
                                        const int* forced_tokens, void* workspace, size_t workspace_bytes,
                                        float* out, int* tokens_out, int* status_out, int flags, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(n_layers > 0 && layer_ptrs_len == n_layers * D_COUNT + 1,
                  "tpspp_nrtr_decoder_fwd: layer_ptrs_len must be n_layers * %d + 1 (24 pointers per layer, then the "
                  "arranged classifier)", D_COUNT);
    const bool b16 = (flags & TPSPP_HEAD_BF16) != 0;
    const bool x3 = !b16 && (flags & TPSPP_HEAD_BF16X3) != 0;
    TPSPP_REQUIRE(enc_cm && layer_ptrs && emb && pos_table && w_cls && cls_colsum && workspace && out,
                  "tpspp_nrtr_decoder_fwd: null pointer");
    TPSPP_REQUIRE(N > 0 && T > 0 && n_layers > 0 && d_inner > 0 && C > 0 && C % kDK == 0 && num_out > 0,
                  "tpspp_nrtr_decoder_fwd: bad sizes (d_model must be a multiple of 64 = n_head * 64)");
    TPSPP_REQUIRE(T <= 256, "tpspp_nrtr_decoder_fwd: at most 256 encoder tokens per image");
    TPSPP_REQUIRE(max_seq_len >= 1 && max_seq_len <= kWave, "tpspp_nrtr_decoder_fwd: max_seq_len must be in [1, 64]");
    TPSPP_REQUIRE(n_position >= max_seq_len, "tpspp_nrtr_decoder_fwd: position table shorter than max_seq_len");
    TPSPP_REQUIRE(workspace_bytes >= dec_ws_bytes(N, C, T, d_inner, n_layers, max_seq_len, num_out),
                  "tpspp_nrtr_decoder_fwd: workspace too small");
    for (int l = 0; l < n_layers; ++l) {
        const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
        TPSPP_REQUIRE(w[D_QKV_W] && w[D_QKV_CS] && w[D_WFC] && w[D_Q_W] && w[D_Q_CS] && w[D_WK] && w[D_WV] && w[D_WFC2] &&
                          w[D_W1_W] && w[D_W1_CS] && w[D_W2],
                      "tpspp_nrtr_decoder_fwd: layer %d has a null weight", l);
    }
    hipStream_t st = tpspp::as_stream(stream);
    const int H = C / kDK, L = max_seq_len, Lt = L + 1;
    const int MT = N * T;
    Carver cv{reinterpret_cast<char*>(workspace), workspace_bytes};
    float *Kx[64], *Vx[64], *Kc[64], *Vc[64];
    TPSPP_REQUIRE(n_layers <= 64, "tpspp_nrtr_decoder_fwd: at most 64 layers");
    for (int l = 0; l < n_layers; ++l) {
        Kx[l] = cv.f((size_t)C * MT);
        Vx[l] = cv.f((size_t)C * MT);
        Kc[l] = cv.f((size_t)N * C * L);
        Vc[l] = cv.f((size_t)N * C * L);
    }
    float* x = cv.f((size_t)C * N);
    float* y = cv.f((size_t)C * N);
    float* a = cv.f((size_t)C * N);
    float* qkv = cv.f((size_t)3 * C * N);
    float* hid = cv.f((size_t)d_inner * N);
    float* logits = cv.f((size_t)num_out * N);
    int* tokens = reinterpret_cast<int*>(cv.f((size_t)N * Lt));
    float* ktmp = cv.f((size_t)C * MT);
    int* pcounters = reinterpret_cast<int*>(cv.f((size_t)(N + 31) / 32 * 32 + 64));
    TPSPP_REQUIRE(cv.ok, "tpspp_nrtr_decoder_fwd: workspace carve failed");

    // Reduced-precision head (TPSPP_HEAD_BF16 / _BF16X3) with arranged per-step weights in the table: every activation of
    // a step is TOKEN-major and the six projections + the classifier run on dec_gemm_x3_kernel (three-term split: inside
    // the fp32 tolerance, so the bf16 head takes it as well).  8 launches per layer-step as before, each about half as long.
    const void* cls_x = layer_ptrs[(size_t)n_layers * D_COUNT];       // the classifier, arranged (behind the layers), or NULL
    const bool gemm_f32 = !(b16 || x3);                               // exact fp32 products; else the three-term split
    bool fast = cls_x != nullptr && (C == 256 || C == 512) && (d_inner == 256 || d_inner == 512) && (num_out % 4) == 0;
    for (int l = 0; l < n_layers && fast; ++l) {
        const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
        fast = w[D_QKV_X] && w[D_WFC_X] && w[D_Q_X] && w[D_WFC2_X] && w[D_W1_X] && w[D_W2_X];
    }
    Gemm g{st};
    // encoder keys (channel-major; token-major for the step pipeline above) and values (token-major) of every layer, once
    for (int l = 0; l < n_layers; ++l) {
        const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
        if (b16) {
            // bf16 matrix cores, bf16 keys / values: K channel-major straight from the epilogue; V channel-major
            // into the upper half of its own (fp32-sized) slot, then transposed to token-major rows
            unsigned short* vt = reinterpret_cast<unsigned short*>(Vx[l]) + (size_t)C * MT;
            unsigned short* kt = reinterpret_cast<unsigned short*>(Kx[l]) + (size_t)C * MT;
            g.cm16(w[D_WK], w[D_BK], enc_cm, C, C, MT, kt, 0, 0, nullptr);
            g.cm16(w[D_WV], nullptr, enc_cm, C, C, MT, vt, 0, 0, nullptr);
            if (g.rc) return g.rc;
            hipLaunchKernelGGL(transpose2d_b16_kernel, dim3((unsigned)((MT + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256),
                               0, st, vt, C, MT, reinterpret_cast<unsigned short*>(Vx[l]));
            hipLaunchKernelGGL(transpose2d_b16_kernel, dim3((unsigned)((MT + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256),
                               0, st, kt, C, MT, reinterpret_cast<unsigned short*>(Kx[l]));
        } else {
            float* kdst = ktmp;
            if (x3) g.cm16(w[D_WK], w[D_BK], enc_cm, C, C, MT, kdst, 1, 0, nullptr, 1);   // fp32 keys, three-term split
            else g.cm(w[D_WK], w[D_BK], enc_cm, C, C, MT, kdst, 0, nullptr);
            {                                               // (C, N*T) -> (N*T, C): the wide-load cross-attention's layout
                if (g.rc) return g.rc;
                const int rc_t = tpspp_transpose2d(ktmp, C, MT, Kx[l], stream);
                if (rc_t) return rc_t;
            }
            g.tm(w[D_WV], enc_cm, C, C, MT, Vx[l]);        // (a value bias would be per channel = per column here: not supported)
        }
    }
    if (g.rc) return g.rc;
    hipLaunchKernelGGL(dec_init_tokens_kernel, dim3((unsigned)((N * Lt + 255) / 256)), dim3(256), 0, st, tokens, N,
                       Lt, start_idx, padding_idx, forced_tokens, L);
    const int greedy = forced_tokens ? 0 : 1;
    const unsigned pair_blocks = (unsigned)((N * H + 3) / 4);
    int rc = 0;
    // ---- the step as ONE persistent launch (tpspp_head_persist.h): every head configuration, d_model 512, 8 heads ----
    bool persist = fast && C == 512 && H == 8 && num_out <= 128 && n_layers <= kPMaxLayers && !g_head_qcross && !head_no_persist();
    const bool tbig = T > kWave;                              // more than 64 encoder tokens: the kernel's other instantiation
    auto kern = tbig ? (b16 ? (d_inner == 256 ? dec_step_persist_kernel<unsigned short, 2, false, true> : dec_step_persist_kernel<unsigned short, 4, false, true>)
                            : gemm_f32 ? (d_inner == 256 ? dec_step_persist_kernel<float, 2, true, true> : dec_step_persist_kernel<float, 4, true, true>)
                                       : (d_inner == 256 ? dec_step_persist_kernel<float, 2, false, true> : dec_step_persist_kernel<float, 4, false, true>))
                     : (b16 ? (d_inner == 256 ? dec_step_persist_kernel<unsigned short, 2, false> : dec_step_persist_kernel<unsigned short, 4, false>)
                            : gemm_f32 ? (d_inner == 256 ? dec_step_persist_kernel<float, 2, true> : dec_step_persist_kernel<float, 4, true>)
                                       : (d_inner == 256 ? dec_step_persist_kernel<float, 2, false> : dec_step_persist_kernel<float, 4, false>));
    int dev = 0, groups = 0;
    if (persist) {
        // requirement (1): a whole group of clusters resident, else the launch pipeline below; not under stream capture (the
        // in-flight guard's events would become graph nodes)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= tpspp::kMaxDevices ||
            hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            persist = false;
        } else {
            std::lock_guard<std::mutex> lk(g_persist_mu);
            groups = persist_groups(g_persist_dev[dev], (tbig ? 6 : 0) + (b16 ? 2 : gemm_f32 ? 4 : 0) + (d_inner == 256 ? 0 : 1), kern, dev);
            persist = groups > 0;
        }
    }
    if (status_out && !persist && hipMemsetAsync(status_out, 0, sizeof(int), st) != hipSuccess)
        return tpspp::check_launch("tpspp_nrtr_decoder_fwd(status)");
    if (persist) {
        PStep PS;
        for (int l = 0; l < n_layers; ++l) {
            const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
            PLayer& q = PS.L[l];
            q.qkv_x = w[D_QKV_X]; q.wfc_x = w[D_WFC_X]; q.q_x = w[D_Q_X]; q.wfc2_x = w[D_WFC2_X]; q.w1_x = w[D_W1_X]; q.w2_x = w[D_W2_X];
            q.qkv_cs = w[D_QKV_CS]; q.qkv_b = w[D_QKV_B]; q.bfc = w[D_BFC]; q.q_cs = w[D_Q_CS]; q.q_b = w[D_Q_B];
            q.bfc2 = w[D_BFC2]; q.w1_cs = w[D_W1_CS]; q.w1_b = w[D_W1_B]; q.b2 = w[D_B2];
            q.Kx = Kx[l]; q.Vx = Vx[l]; q.Kc = Kc[l]; q.Vc = Vc[l];
        }
        PS.n_layers = n_layers;
        PS.a = a; PS.qkv = qkv; PS.hid = hid; PS.logits = logits;
        PS.cls_x = cls_x; PS.cls_cs = cls_colsum; PS.cls_b = b_cls; PS.num_out = num_out;
        PS.emb = emb; PS.pos = pos_table; PS.tokens = tokens; PS.Lt = Lt; PS.out = out; PS.greedy = greedy; PS.pad_idx = padding_idx;
        PS.valid_len = valid_len; PS.N = N; PS.C = C; PS.T = T; PS.H = H; PS.d_inner = d_inner; PS.Lsteps = L; PS.Lmax = L;
        PS.err = pcounters + (size_t)(N + 31) / 32 * 32;
        PS.pairs = 2;
        PS.trace = g_head_trace;
        PS.no_plain = getenv("TPSPP_HEAD_WRITE_THROUGH") ? 1 : 0;
        // barrier timeout: wall-clock milliseconds (default 4000) -> units of 1024 ticks of the 100 MHz clock
        { const char* tv = getenv("TPSPP_HEAD_TIMEOUT_MS"); const long ms = tv ? atol(tv) : 4000; PS.timeout_k = (int)((ms < 1 ? 1 : ms > 60000 ? 60000 : ms) * 100000 / 1024); }
        { const char* tv = getenv("TPSPP_HEAD_TEST_STALL"); PS.test_stall_step = tv ? atoi(tv) : -1; }
        // odd clusters start 15 us late (TPSPP_HEAD_STAGGER_US overrides; 0 = together): bf16x3 20.0 -> 19.4 ms, bf16 15.4 -> 15.0
        { const char* sv = getenv("TPSPP_HEAD_STAGGER_US"); PS.stagger = (sv ? atoi(sv) : (N > 32 ? 15 : 0)) * 100; }
        // requirement (2): behind the previous persistent decode of this device, whatever stream it ran on
        std::lock_guard<std::mutex> lk(g_persist_mu);
        PersistDevice& PD = g_persist_dev[dev];
        if (!PD.done && hipEventCreateWithFlags(&PD.done, hipEventDisableTiming) != hipSuccess)
            return tpspp::check_launch("tpspp_nrtr_decoder_fwd(event)");
        if (PD.recorded && hipStreamWaitEvent(st, PD.done, 0) != hipSuccess)
            return tpspp::check_launch("tpspp_nrtr_decoder_fwd(wait for the previous persistent decode)");
        if (hipMemsetAsync(pcounters, 0, ((size_t)(N + 31) / 32 * 32 + 64) * sizeof(int), st) != hipSuccess)
            return tpspp::check_launch("tpspp_nrtr_decoder_fwd(memset)");
        hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)((C * N + 255) / 256)), dim3(256), 0, st, emb, pos_table, tokens, Lt, 0, C,
                           N, x, 1);
        const int per_step = 8 * n_layers + 2;                 // cluster barriers of one step
        // ONE launch per decode: the kernel loops over the steps, clusters run their 40 steps independently of each other.
        // (With the clusters spread over the XCDs -- the first version of this kernel -- one launch per step was faster:
        // 23.04 against 23.25 ms; with a cluster per XCD and ordinary stores the step loop inside wins: fp32 22.12 -> 21.58 ms,
        // bf16x3 19.67 -> 19.07, bf16 14.29 -> 13.83 at batch 512.)  TPSPP_HEAD_STEP_LAUNCHES=1: one launch per step -- same
        // scores (tests/test_gpu_head.py), for A/B runs.
        const int per_launch = getenv("TPSPP_HEAD_STEP_LAUNCHES") ? 1 : L;
        const int img_per_launch = 256 * groups;               // every cluster of a launch resident: `groups` x 8 clusters x 32 images
        for (int s = 0; s < L; s += per_launch) {
            // (+ one placement-check barrier per launch)
            PS.x = x; PS.y = y; PS.step = s; PS.nsteps = per_launch; PS.bar_base = s * per_step + s / per_launch;
            for (int n0 = 0; n0 < N; n0 += img_per_launch) {
                const int nimg = N - n0 < img_per_launch ? N - n0 : img_per_launch;
                PS.n0 = n0; PS.counters = pcounters + (n0 >> 5) * 32;
                PS.nclusters = (nimg + 31) / 32;
                // cluster c = 8 j + x is the 16 blocks 8 (16 j + ct) + x: one XCD per cluster (see the kernel)
                hipLaunchKernelGGL(kern, dim3((unsigned)((PS.nclusters + 7) / 8 * 128)), dim3(512), sizeof(PShared), st, PS);
            }
            if (per_launch == 1) {
                if (n_layers & 1) { float* t = x; x = y; y = t; }  // (the kernel swaps x / y once per layer)
                if (s + 1 < L) { float* t = x; x = y; y = t; }     // the next step's embedding went to y
            }
        }
        const hipError_t rec = hipEventRecord(PD.done, st);
        PD.recorded = PD.recorded || rec == hipSuccess;
        if (tokens_out)
            (void)hipMemcpyAsync(tokens_out, tokens, (size_t)N * Lt * sizeof(int), hipMemcpyDeviceToDevice, st);
        if (status_out)
            (void)hipMemcpyAsync(status_out, PS.err, sizeof(int), hipMemcpyDeviceToDevice, st);
        return tpspp::check_launch("tpspp_nrtr_decoder_fwd(persistent step)");
    }
    for (int s = 0; s < L && fast; ++s) {
        if (s == 0)
            hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)((C * N + 255) / 256)), dim3(256), 0, st, emb, pos_table,
                               tokens, Lt, s, C, N, x, 1);
        for (int l = 0; l < n_layers; ++l) {
            const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
            dec_gemm_x3(st, x, w[D_QKV_X], w[D_QKV_B], w[D_QKV_CS], 1e-5f, nullptr, 0, N, C, 3 * C, qkv, gemm_f32);
            if (b16)
                hipLaunchKernelGGL(attn_dec_self_wide_kernel<unsigned short>, dim3(pair_blocks), dim3(256), 0, st, qkv, C, N, H, s,
                                   L, reinterpret_cast<unsigned short*>(Kc[l]), reinterpret_cast<unsigned short*>(Vc[l]),
                                   tokens, Lt, padding_idx, a, 0);
            else
                hipLaunchKernelGGL(attn_dec_self_wide_kernel<float>, dim3(pair_blocks), dim3(256), 0, st, qkv, C, N, H, s, L, Kc[l],
                                   Vc[l], tokens, Lt, padding_idx, a, 0);
            dec_gemm_x3(st, a, w[D_WFC_X], w[D_BFC], nullptr, 0.0f, x, 0, N, C, C, y, gemm_f32);              // y = x + fc(a)
            if (!gemm_f32 && C == 512 && g_head_qcross) {
                // q projection + cross-attention in one launch (dec_q_cross_x3_kernel)
                DGemm Q;
                Q.X = y; Q.Wp = reinterpret_cast<const du32x4*>(w[D_Q_X]); Q.bias = w[D_Q_B]; Q.colsum = w[D_Q_CS]; Q.res = nullptr;
                Q.out = nullptr; Q.M = N; Q.K = C; Q.Co = C; Q.eps = 1e-5f; Q.act = 0;
                const dim3 qgrid((unsigned)((N + 15) / 16), (unsigned)H);
                if (b16)
                    hipLaunchKernelGGL(dec_q_cross_x3_kernel<unsigned short>, qgrid, dim3(256), 0, st, Q,
                                       reinterpret_cast<const unsigned short*>(Kx[l]),
                                       reinterpret_cast<const unsigned short*>(Vx[l]), H, T, valid_len, a);
                else
                    hipLaunchKernelGGL(dec_q_cross_x3_kernel<float>, qgrid, dim3(256), 0, st, Q, Kx[l], Vx[l], H, T, valid_len, a);
            } else {
            dec_gemm_x3(st, y, w[D_Q_X], w[D_Q_B], w[D_Q_CS], 1e-5f, nullptr, 0, N, C, C, qkv, gemm_f32);
            if (T <= kWave && (H & 1) == 0 && !g_head_cross1) {
                // two heads per wavefront (round 5): both heads' keys, then both heads' values in flight together
                const unsigned blocks2 = (unsigned)((N * (H / 2) + 3) / 4);
                if (b16)
                    hipLaunchKernelGGL(attn_dec_cross_wide2_kernel<unsigned short>, dim3(blocks2), dim3(256), 0, st, qkv, C,
                                       reinterpret_cast<const unsigned short*>(Kx[l]),
                                       reinterpret_cast<const unsigned short*>(Vx[l]), C, N, H, T, valid_len, a);
                else
                    hipLaunchKernelGGL(attn_dec_cross_wide2_kernel<float>, dim3(blocks2), dim3(256), 0, st, qkv, C, Kx[l], Vx[l], C, N,
                                       H, T, valid_len, a);
            } else if (b16)
                hipLaunchKernelGGL(attn_dec_cross_wide_kernel<unsigned short>, dim3(pair_blocks), dim3(256), 0, st, qkv,
                                   reinterpret_cast<const unsigned short*>(Kx[l]),
                                   reinterpret_cast<const unsigned short*>(Vx[l]), C, N, H, T, valid_len, a, 0);
            else
                hipLaunchKernelGGL(attn_dec_cross_wide_kernel<float>, dim3(pair_blocks), dim3(256), 0, st, qkv, Kx[l], Vx[l], C,
                                   N, H, T, valid_len, a, 0);
            }
            dec_gemm_x3(st, a, w[D_WFC2_X], w[D_BFC2], nullptr, 0.0f, y, 0, N, C, C, x, gemm_f32);            // x = y + fc(a)
            dec_gemm_x3(st, x, w[D_W1_X], w[D_W1_B], w[D_W1_CS], 1e-5f, nullptr, 2, N, C, d_inner, hid, gemm_f32);
            dec_gemm_x3(st, hid, w[D_W2_X], w[D_B2], nullptr, 0.0f, x, 0, N, d_inner, C, y, gemm_f32);        // y = x + w2(...)
            float* t = x; x = y; y = t;
        }
        dec_gemm_x3(st, x, cls_x, b_cls, cls_colsum, 1e-6f, nullptr, 0, N, C, num_out, logits, gemm_f32);
        // (x holds this step's final activations, read by the classifier launch above; the next step's embedding goes to y,
        //  which becomes x)
        const bool more = s + 1 < L;
        hipLaunchKernelGGL(dec_classify_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, logits, num_out,
                           N, s, L, greedy, out, tokens, Lt, 1, emb, pos_table, more ? y : nullptr, C);
        if (more) { float* t = x; x = y; y = t; }
    }
    for (int s = 0; s < L && !fast; ++s) {
        hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)((C * N + 255) / 256)), dim3(256), 0, st, emb, pos_table,
                           tokens, Lt, s, C, N, x, 0);
        for (int l = 0; l < n_layers; ++l) {
            const float* const* w = layer_ptrs + (size_t)l * D_COUNT;
            // x = x + fc(self_attn(LN1(x)))                          transformer_layers.py:150-154
            rc = tpspp_linear_ln_fwd(x, C, N, 1e-5f, w[D_QKV_W], w[D_QKV_CS], 3 * C, w[D_QKV_B], 0, nullptr, 1, qkv, stream);
            if (rc) return rc;
            if (b16)
                hipLaunchKernelGGL(attn_dec_self_wide_kernel<unsigned short>, dim3(pair_blocks), dim3(256), 0, st, qkv, C, N, H, s,
                                   L, reinterpret_cast<unsigned short*>(Kc[l]), reinterpret_cast<unsigned short*>(Vc[l]),
                                   tokens, Lt, padding_idx, a, 1);
            else
                hipLaunchKernelGGL(attn_dec_self_wide_kernel<float>, dim3(pair_blocks), dim3(256), 0, st, qkv, C, N, H, s, L, Kc[l],
                                   Vc[l], tokens, Lt, padding_idx, a, 1);
            g.cm(w[D_WFC], w[D_BFC], a, C, C, N, y, 0, x);            // y = x + fc(a)
            // x = y + fc(enc_attn(LN2(y), enc, enc))                   transformer_layers.py:156-159
            rc = tpspp_linear_ln_fwd(y, C, N, 1e-5f, w[D_Q_W], w[D_Q_CS], C, w[D_Q_B], 0, nullptr, 1, qkv, stream);
            if (rc) return rc;
            if (b16)
                hipLaunchKernelGGL(attn_dec_cross_wide_kernel<unsigned short>, dim3(pair_blocks), dim3(256), 0, st, qkv,
                                   reinterpret_cast<const unsigned short*>(Kx[l]),
                                   reinterpret_cast<const unsigned short*>(Vx[l]), C, N, H, T, valid_len, a, 1);
            else
                hipLaunchKernelGGL(attn_dec_cross_wide_kernel<float>, dim3(pair_blocks), dim3(256), 0, st, qkv, Kx[l], Vx[l], C,
                                   N, H, T, valid_len, a, 1);
            g.cm(w[D_WFC2], w[D_BFC2], a, C, C, N, x, 0, y);          // x = y + fc(a)
            // x = x + mlp(LN3(x))                                       transformer_layers.py:161-163
            rc = tpspp_linear_ln_fwd(x, C, N, 1e-5f, w[D_W1_W], w[D_W1_CS], d_inner, w[D_W1_B], 2, nullptr, 0, hid, stream);
            if (rc) return rc;
            g.cm(w[D_W2], w[D_B2], hid, d_inner, C, N, y, 0, x);      // y = x + w2(...)
            float* t = x; x = y; y = t;
            if (g.rc) return g.rc;
        }
        // final LayerNorm (eps 1e-6) folded into the classifier     nrtr_decoder.py:77,111 + :78
        rc = tpspp_linear_ln_fwd(x, C, N, 1e-6f, w_cls, cls_colsum, num_out, b_cls, 0, nullptr, 0, logits, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dec_classify_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, logits, num_out,
                           N, s, L, greedy, out, tokens, Lt, 0);
    }
    if (tokens_out)
        (void)hipMemcpyAsync(tokens_out, tokens, (size_t)N * Lt * sizeof(int), hipMemcpyDeviceToDevice, st);
    return tpspp::check_launch("tpspp_nrtr_decoder_fwd");
}

// ===== AttnConvertor.tensor2idx on the device (round 6) =================================================================
// scores (N, L, C): per position the maximum and its FIRST index (torch.max's tie rule; a NaN beats every number, as in ATen's
// reduction), then the reference's scan per image -- skip <PAD>, stop at the first <EOS> (convertors/attn.py:124-137) -- so that
// the host needs ONE copy of (N, L) indices + (N, L) scores instead of the score tensor's arg-max in fp64 and a Python scan.
// One wavefront per image: lanes over classes while reducing a position, lane = position while scanning.
namespace {
__global__ void __launch_bounds__(256)
attn_tensor2idx_kernel(const float* __restrict__ scores, int N, int L, int C, int end_idx, int pad_idx,
                       int* __restrict__ idx_out, float* __restrict__ val_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= N) return;
    bool ended = false;                                     // (wave-uniform)
    for (int l0 = 0; l0 < L; l0 += kWave) {
        const int nl = L - l0 < kWave ? L - l0 : kWave;
        float my_v = 0.0f;
        int my_i = -1;
        for (int j = 0; j < nl; ++j) {
            const float* row = scores + ((size_t)b * L + l0 + j) * C;
            float mx = -INFINITY;
            int am = 0x7fffffff;
            bool nan = false;
            for (int c = lane; c < C; c += kWave) {
                const float v = row[c];
                if (!nan && (v != v)) { nan = true; mx = v; am = c; }
                else if (!nan && (v > mx || am == 0x7fffffff)) { mx = v; am = c; }   // (first element always taken: a row of -inf has index 0)
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(mx, off, kWave);
                const int oi = __shfl_xor(am, off, kWave);
                const bool on = __shfl_xor((int)nan, off, kWave) != 0;
                const bool take = oi != 0x7fffffff &&
                                  (am == 0x7fffffff || (on && !nan) || (on == nan && (on ? oi < am : (ov > mx || (ov == mx && oi < am)))));
                if (take) { mx = ov; am = oi; nan = on; }
            }
            if (lane == j) { my_v = mx; my_i = am; }
        }
        const bool valid = lane < nl;
        const unsigned long long ends = __ballot(valid && my_i == end_idx);
        const int first_end = ended ? 0 : (ends ? __ffsll((long long)ends) - 1 : kWave);
        if (valid) {
            const bool keep = lane < first_end && my_i != pad_idx;
            idx_out[(size_t)b * L + l0 + lane] = keep ? my_i : -1;
            val_out[(size_t)b * L + l0 + lane] = my_v;
        }
        ended = ended || ends != 0;
    }
}
}  // namespace

TPSPP_EXPORT int tpspp_attn_tensor2idx_fwd(const float* scores, int N, int L, int C, int end_idx, int padding_idx,
                                           int* idx_out, float* val_out, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(scores && idx_out && val_out && N > 0 && L > 0 && C > 0, "tpspp_attn_tensor2idx_fwd: bad argument");
    TPSPP_REQUIRE((size_t)N * L * C < ((size_t)1 << 40) && N <= (1 << 30), "tpspp_attn_tensor2idx_fwd: batch too large");
    hipLaunchKernelGGL(attn_tensor2idx_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, tpspp::as_stream(stream), scores, N, L, C,
                       end_idx, padding_idx, idx_out, val_out);
    return tpspp::check_launch("tpspp_attn_tensor2idx_fwd");
}
