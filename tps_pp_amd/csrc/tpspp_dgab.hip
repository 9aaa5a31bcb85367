// DGAB (dynamic gated-attention block) of the TPS++ regressor for gfx950, two kernels:
//
//   dgab_gate_kernel   (VALU, HBM-bound)  one wavefront per (image, channel) plane of 16x64:
//       LayerNorm over the plane -> xn;  w = mlp_w([mean_H(xn), y]),  h = mlp_h([mean_W(xn), y]);
//       v_w = softmax(w[:-1]), v_h = softmax(h[:-1]);  A = v_h*xn*h[-1] + v_w*xn*w[-1]
//   dgab_chain_kernel  (fp32 MFMA)        persistent workgroups, 128 rows (8 planes) per tile:
//       x1 = x + proj(A);  out = x1 + fc2(gelu(fc1(LayerNorm(x1))))
//       with proj / fc1 / fc2 = nn.Linear along the LAST axis (W = 64 = dim, DGAB.py:36,52,71,76).
//
// The chain kernel never moves activations through LDS: a lane owns one tensor row (64 values split
// between the two half-wavefront lanes that share the row) in exactly the register pattern in which
// v_mfma_f32_32x32x2_f32 delivers its result -- lane (row, half), accumulator h2, register r holds
// feature f = 32*h2 + (r&3) + 8*(r>>2) + 4*half -- and the next GEMM consumes those registers
// directly as its B fragments, with the weight slabs pre-permuted on the host so that k-slot
// (h2*16 + r, half) multiplies feature f.  LDS holds only the weights (Wp, W1, W2: 144 KB, loaded
// once per persistent workgroup).  Bound: MFMA (38.6 GFLOP per 512 images at the fp32 matrix rate).
//
// Reference: mmocr/models/textrecog/backbones/tps_pp/DGAB.py:25-77 (DGAB_Block.forward, Mlp.forward,
// DGAB.forward), called from tps_pp.py:318-319.
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
constexpr int H = 16, W = 64, PT = 32, HID = 256;       // plane, points, MLP hidden width
constexpr float kEps = 1e-5f;                            // nn.LayerNorm default

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// v_cvt_pk_bf16_f32: two fp32 -> packed bf16, round to nearest even
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Wavefront-wide reductions on the DPP path of the vector ALU (quad swaps, half-row and row mirrors, row broadcasts,
// one v_readlane): 7 instructions and no LDS traffic, against 6 ds_bpermute round trips for the __shfl_xor butterfly --
// the gate kernel does 24 of them per plane.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(0.0f, v);                 // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(0.0f, v);                 // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(0.0f, v);                // row_half_mirror
    v += dpp_f<0x140>(0.0f, v);                // row_mirror: every lane holds its row's sum
    v += dpp_f<0x142, 0xa>(0.0f, v);           // row_bcast15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(0.0f, v);           // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return readlane_f(v, 63);
}

__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v, v));
    v = fmaxf(v, dpp_f<0x4E>(v, v));
    v = fmaxf(v, dpp_f<0x141>(v, v));
    v = fmaxf(v, dpp_f<0x140>(v, v));
    v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
    return readlane_f(v, 63);
}

// ------------------------------------------------------------------------------------------------
// gate kernel: lane = column, 16 registers = the rows of that column
struct GateParams {
    const float* x;        // (N, C, 16, 64)
    const float* y;        // (N, C, 32): en_feat viewed per channel (the reference's y^T)
    const float* g1; const float* b1;       // LayerNorm affine (16, 64)
    const float* mw_t;     // (96, 65): mlp_w weight transposed  [input][output]
    const float* mh_t;     // (48, 17): mlp_h weight transposed
    void* a;               // (N, C, 16, 64), fp32 or bf16 (A16)
    int planes;
};

// A16: the gated map is written as bf16 (the bf16 chain kernel reads it straight into MFMA operands)
template <bool A16>
__global__ void __launch_bounds__(256)
dgab_gate_kernel(const GateParams P)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int pl = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (pl >= P.planes) return;
    const float* xp = P.x + (size_t)pl * H * W;
    float v[H];
#pragma unroll
    for (int r = 0; r < H; ++r) v[r] = xp[r * W + lane];
    const float yv = lane < PT ? P.y[(size_t)pl * PT + lane] : 0.0f;

    // LayerNorm over the 16x64 plane (two-pass)
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < H; ++r) s += v[r];
    const float mean = wave_sum(s) * (1.0f / (H * W));
    float q = 0.0f;
#pragma unroll
    for (int r = 0; r < H; ++r) { const float d = v[r] - mean; q += d * d; }
    const float var = wave_sum(q) * (1.0f / (H * W));
    const float rstd = 1.0f / sqrtf(var + kEps);
    float colsum = 0.0f;
    float rowmean[H];
#pragma unroll
    for (int r = 0; r < H; ++r) {
        v[r] = (v[r] - mean) * rstd * P.g1[r * W + lane] + P.b1[r * W + lane];
        colsum += v[r];
        rowmean[r] = wave_sum(v[r]) * (1.0f / W);      // x.mean(3): over the columns
    }
    const float colmean = colsum * (1.0f / H);          // x.mean(2): over the rows

    // w = mlp_w(cat[colmean (64), y (32)]) -> 65 outputs: lane o < 64 owns w[o]; w[64] by every lane
    float wo = 0.0f, wlast = 0.0f;
    for (int i = 0; i < W; ++i) {
        const float in = readlane_f(colmean, i);
        wo = fmaf(P.mw_t[i * (W + 1) + lane], in, wo);
        wlast = fmaf(P.mw_t[i * (W + 1) + W], in, wlast);
    }
    for (int t = 0; t < PT; ++t) {
        const float in = readlane_f(yv, t);
        wo = fmaf(P.mw_t[(W + t) * (W + 1) + lane], in, wo);
        wlast = fmaf(P.mw_t[(W + t) * (W + 1) + W], in, wlast);
    }
    // h = mlp_h(cat[rowmean (16), y (32)]) -> 17 outputs: lane o < 17 owns h[o]
    float ho = 0.0f;
    {
        const int o = lane < H + 1 ? lane : H;
#pragma unroll
        for (int r = 0; r < H; ++r) ho = fmaf(P.mh_t[r * (H + 1) + o], rowmean[r], ho);
        for (int t = 0; t < PT; ++t) ho = fmaf(P.mh_t[(H + t) * (H + 1) + o], readlane_f(yv, t), ho);
    }
    // softmaxes over w[0:64] (all lanes) and h[0:16] (lanes 0..15)
    const float wmax = wave_max(wo);
    const float we = expf(wo - wmax);
    const float vw = we / wave_sum(we);
    const float hin = lane < H ? ho : -INFINITY;
    const float hmax = wave_max(hin);
    const float he = lane < H ? expf(ho - hmax) : 0.0f;
    const float vh = he / wave_sum(he);
    const float hlast = readlane_f(ho, H);

    // A = (v_h * xn) * h_last + (v_w * xn) * w_last      (op order of DGAB.py:50)
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const float vhr = readlane_f(vh, r);
        float t1 = vhr * v[r];
        t1 = t1 * hlast;
        float t2 = vw * v[r];
        t2 = t2 * wlast;
        const size_t o = (size_t)pl * H * W + r * W + lane;
        if (A16) reinterpret_cast<unsigned short*>(P.a)[o] = (unsigned short)(pack_bf16(t1 + t2, 0.0f) & 0xffffu);
        else reinterpret_cast<float*>(P.a)[o] = t1 + t2;
    }
}

// ------------------------------------------------------------------------------------------------
// chain kernel
struct ChainParams {
    const float* x;        // (rows, 64) residual input (the DGAB input)
    const float* a;        // (rows, 64) gated attention output (input of proj)
    const float* wp_s;     // [64 k-slots][64 out]           proj,  permuted slab
    const float* w1_s;     // [4 blocks][64 k-slots][64 out]  fc1 (out = hidden unit inside block)
    const float* w2_s;     // [4 blocks][64 k-slots][64 out]  fc2 (k-slot = hidden unit inside block)
    const float* bp; const float* b1; const float* b2;       // (64), (256), (64)
    const float* g2; const float* be2;                       // LayerNorm-2 affine (16, 64)
    float* out;            // (rows, 64)
    int wtiles;            // rows / 32: one 32-row tile per wavefront and loop trip
};

// feature owned by (k-slot index ks = h2*16 + r, half) in the MFMA C/D layout
__device__ __forceinline__ constexpr int feat(int ks, int half)
{
    return 32 * (ks >> 4) + (ks & 3) + 8 * ((ks & 15) >> 2) + 4 * half;
}

// D[i][row] += sum over the 32 k-slot pairs: A = slab[(2*ks + half)*64 + i], B = in[ks]
__device__ __forceinline__ void gemm64(const float* __restrict__ slab, const float (&in)[32], int half, int l31,
                                       f32x16& acc0, f32x16& acc1)
{
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        const float a0 = slab[(2 * ks + half) * 64 + l31];
        const float a1 = slab[(2 * ks + half) * 64 + 32 + l31];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, in[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, in[ks], acc1, 0, 0, 0);
    }
}

__device__ __forceinline__ float plane_sum(float v)
{
    // the 32 lanes that hold one 16-row plane: row bits 0..3 of the lane index and the half bit (5)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 32);
    return v;
}

// NWAVE wavefronts share one copy of the weights in LDS.  With 8 (two per SIMD, <= 256 registers each) one wavefront's
// LayerNorm / GELU vector work runs under the other's matrix instructions; each wavefront walks its own 32-row tiles,
// there is no barrier after the weight load.
template <int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64)
dgab_chain_kernel(const ChainParams P)
{
    constexpr int NT = NWAVE * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sWp = smem;                       // 64*64
    float* sW1 = sWp + 64 * 64;              // 4*64*64
    float* sW2 = sW1 + 4 * 64 * 64;          // 4*64*64
    float* sB = sW2 + 4 * 64 * 64;           // bp (64) | b1 (256) | b2 (64)
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64 / 4; i += NT) reinterpret_cast<float4*>(sWp)[i] = reinterpret_cast<const float4*>(P.wp_s)[i];
    for (int i = tid; i < 4 * 64 * 64 / 4; i += NT) {
        reinterpret_cast<float4*>(sW1)[i] = reinterpret_cast<const float4*>(P.w1_s)[i];
        reinterpret_cast<float4*>(sW2)[i] = reinterpret_cast<const float4*>(P.w2_s)[i];
    }
    for (int i = tid; i < 64; i += NT) { sB[i] = P.bp[i]; sB[64 + HID + i] = P.b2[i]; }
    for (int i = tid; i < HID; i += NT) sB[64 + i] = P.b1[i];
    __syncthreads();

    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int prow = l31 & (H - 1);          // row inside its 16-row plane (LayerNorm affine index)

    for (int wt = blockIdx.x * NWAVE + wv; wt < P.wtiles; wt += gridDim.x * NWAVE) {
        const size_t row = (size_t)wt * 32 + l31;
        // the weights do not change between tiles, and hoisting 64 + 64 loop-invariant LDS / parameter reads out of
        // this loop is what the optimiser does if it can see that -- at the price of spilling them; hide it
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* tWp = sWp + opaque;
        const float* tW1 = sW1 + opaque;
        const float* tW2 = sW2 + opaque;
        const float* tB = sB + opaque;
        const float* tg2 = P.g2 + opaque;
        const float* tbe2 = P.be2 + opaque;
        const float* ar = P.a + row * W;
        const float* xr = P.x + row * W;
        // ---- this lane's 32 features of its row: float4 groups at f = 32*h2 + 8*q + 4*half ----
        float in[32], xres[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {        // g = h2*4 + q  <->  k-slots 4g .. 4g+3
            const int f = 32 * (g >> 2) + 8 * (g & 3) + 4 * half;
            const float4 va = *reinterpret_cast<const float4*>(ar + f);
            const float4 vx = *reinterpret_cast<const float4*>(xr + f);
            in[4 * g + 0] = va.x; in[4 * g + 1] = va.y; in[4 * g + 2] = va.z; in[4 * g + 3] = va.w;
            xres[4 * g + 0] = vx.x; xres[4 * g + 1] = vx.y; xres[4 * g + 2] = vx.z; xres[4 * g + 3] = vx.w;
        }
        // ---- x1 = x + proj(A) ----
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.0f; c1[i] = 0.0f; }
        gemm64(tWp, in, half, l31, c0, c1);
        float x1[32];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const float p = (ks < 16 ? c0[ks & 15] : c1[ks & 15]) + tB[feat(ks, half)];
            x1[ks] = xres[ks] + p;
        }
        // ---- LayerNorm over each 16x64 plane (32 lanes x 32 registers), two-pass ----
        float s = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) s += x1[ks];
        const float mean = plane_sum(s) * (1.0f / (H * W));
        float q = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) { const float d = x1[ks] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(plane_sum(q) * (1.0f / (H * W)) + kEps);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int f = feat(ks, half);
            in[ks] = (x1[ks] - mean) * rstd * tg2[prow * W + f] + tbe2[prow * W + f];
        }
        // ---- out = x1 + fc2(gelu(fc1(xn))): hidden units in 4 blocks of 64, never leaving registers ----
        f32x16 o0, o1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] = 0.0f; o1[i] = 0.0f; }
#pragma unroll 1
        for (int hb = 0; hb < 4; ++hb) {
            f32x16 h0, h1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { h0[i] = 0.0f; h1[i] = 0.0f; }
            gemm64(tW1 + hb * 64 * 64, in, half, l31, h0, h1);
            float hid[32];
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const float z = (ks < 16 ? h0[ks & 15] : h1[ks & 15]) + tB[64 + hb * 64 + feat(ks, half)];
                hid[ks] = 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));     // exact GELU
            }
            gemm64(tW2 + hb * 64 * 64, hid, half, l31, o0, o1);
        }
        float* orow = P.out + row * W;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int f = 32 * (g >> 2) + 8 * (g & 3) + 4 * half;
            float4 v;
            float* pv = &v.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ks = 4 * g + e;
                pv[e] = x1[ks] + ((ks < 16 ? o0[ks & 15] : o1[ks & 15]) + tB[64 + HID + feat(ks, half)]);
            }
            *reinterpret_cast<float4*>(orow + f) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// chain kernel on the bf16 matrix cores (the bf16 configuration, BASELINE.json configs[2])
//
// Same structure -- a lane owns a tensor row in the C/D register pattern, activations never go through LDS --
// with v_mfma_f32_32x32x16_bf16: a lane supplies 8 consecutive k per step, and two consecutive result groups
// (channels 8g + 4h + {0..3}, 8(g+1) + 4h + {0..3}) are exactly such an operand once the next layer's weight
// slab is permuted on the host to that k-slot order (tpspp_front_bf16.hip uses the same chain).  Operands are
// rounded to bf16 once each (gated map, normalised x1, GELU output); x, x1, LayerNorm, GELU and both residual
// sums stay fp32.  The matrix work shrinks 16x (72 instead of 576 MFMAs per 32 rows), which leaves the vector
// ALU as the bound -- mostly the 128 GELUs per lane --, so erf is evaluated with Abramowitz-Stegun 7.1.26
// (|error| < 1.5e-7: invisible even before the bf16 rounding of its result) on v_rcp_f32 / v_exp_f32.
// LDS holds the bf16 weight slabs: 72 KB, two workgroups per CU.
struct ChainBParams {
    const float* x;              // (rows, 64) fp32 residual input
    const void* a;               // (rows, 64) gated map: bf16, or fp32 for the three-term split (X3)
    const u32x4* wp_s;           // [4 k-steps][2][64 out][8]            proj, natural k order
    const u32x4* w1_s;           // [4 blocks][4][2][64 hidden][8]       fc1, chain k order
    const u32x4* w2_s;           // [4 blocks][4][2][64 out][8]          fc2, chain k order (hidden units of the block)
    const float* bp; const float* b1; const float* b2;
    const float* g2; const float* be2;
    float* out;
    int wtiles;                  // rows / 32
};

__device__ __forceinline__ float gelu_as(float z)
{
    const float u = z * 0.70710678118654752440f;
    const float ax = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896340736f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    const float erf_u = __builtin_copysignf(erf_abs, u);
    return 0.5f * z * (1.0f + erf_u);
}

// two values at a time on the packed fp32 instructions (v_pk_mul / v_pk_fma / v_pk_add: the same IEEE operations in the
// same order, so the same bits); at this kernel's two wavefronts per SIMD the packed forms issue a third faster
// (scripts/ubench/valu_bench.hip)
__device__ __forceinline__ f32x2 gelu_as2(f32x2 z)
{
    const f32x2 c_rs2 = {0.70710678118654752440f, 0.70710678118654752440f};
    const f32x2 u = z * c_rs2;
    f32x2 ax; ax[0] = fabsf(u[0]); ax[1] = fabsf(u[1]);
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, ax, one);
    f32x2 t; t[0] = __builtin_amdgcn_rcpf(d[0]); t[1] = __builtin_amdgcn_rcpf(d[1]);
    f32x2 p = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    p = p * t;
    const f32x2 sq = -(ax * ax) * f32x2{1.44269504088896340736f, 1.44269504088896340736f};
    f32x2 e; e[0] = __builtin_amdgcn_exp2f(sq[0]); e[1] = __builtin_amdgcn_exp2f(sq[1]);
    const f32x2 erf_abs = __builtin_elementwise_fma(-p, e, one);
    f32x2 erf_u; erf_u[0] = __builtin_copysignf(erf_abs[0], u[0]); erf_u[1] = __builtin_copysignf(erf_abs[1], u[1]);
    const f32x2 halfv = {0.5f, 0.5f};
    return halfv * z * (one + erf_u);
}

// acc[t] += slab^T in over 4 k-steps (64 k), both 32-row tiles of the 64 outputs
__device__ __forceinline__ void gemm64b(const u32x4* __restrict__ slab, const u32x4 (&in)[4], int half, int l31,
                                        f32x16 (&acc)[2])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bb = __builtin_bit_cast(bf16x8, in[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1], 0, 0, 0);
    }
}

// three-term split ("bf16x3", tpspp_conv_bf16.hip): acc += hi*hi + hi*lo + lo*hi; slab holds hi at [0, 512) and lo at [512, 1024)
__device__ __forceinline__ void gemm64b3(const u32x4* __restrict__ slab, const u32x4 (&in)[4], const u32x4 (&inl)[4], int half,
                                         int l31, f32x16 (&acc)[2])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 a0l = __builtin_bit_cast(bf16x8, slab[512 + (2 * j + half) * 64 + l31]);
        const bf16x8 a1l = __builtin_bit_cast(bf16x8, slab[512 + (2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bb = __builtin_bit_cast(bf16x8, in[j]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, inl[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bb, acc[1], 0, 0, 0);
    }
}

// hi / lo halves of a pair: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(v0, v1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(v0 - h0, v1 - h1);
}

__device__ __forceinline__ void to_operands3(const float (&v)[32], u32x4 (&out)[4], u32x4 (&outl)[4])
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned h, l;
            split2(v[16 * t + 4 * g], v[16 * t + 4 * g + 1], h, l);
            out[2 * t + (g >> 1)][2 * (g & 1)] = h; outl[2 * t + (g >> 1)][2 * (g & 1)] = l;
            split2(v[16 * t + 4 * g + 2], v[16 * t + 4 * g + 3], h, l);
            out[2 * t + (g >> 1)][2 * (g & 1) + 1] = h; outl[2 * t + (g >> 1)][2 * (g & 1) + 1] = l;
        }
}

// 32 fp32 values in the C/D pattern (index 16t + 4g + e <-> feature 32t + 8g + 4h + e) -> chain-ordered operands
__device__ __forceinline__ void to_operands(const float (&v)[32], u32x4 (&out)[4])
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            out[2 * t + (g >> 1)][2 * (g & 1)] = pack_bf16(v[16 * t + 4 * g], v[16 * t + 4 * g + 1]);
            out[2 * t + (g >> 1)][2 * (g & 1) + 1] = pack_bf16(v[16 * t + 4 * g + 2], v[16 * t + 4 * g + 3]);
        }
}

// X3: fp32 gated map and the three-term bf16 split of every product (slabs hold hi and lo: [layer][hi|lo][512]);
// the parity-bound (1e-4) configuration.  Weights then take 144 KB of LDS: one workgroup per CU.
// NWAVE wavefronts per workgroup share the slabs; a wavefront walks its own 32-row tiles (no barrier after the load).
// X3: 8 wavefronts, one workgroup per CU (two per SIMD), weights re-read from LDS per tile (OPQ) so that 256 registers
// suffice: 272 -> 184 us per 512 images.  bf16: 4 wavefronts, two workgroups per CU.
template <bool X3, int NWAVE, bool OPQ = true>
__global__ void __launch_bounds__(NWAVE * 64, X3 ? 1 : 2)
dgab_chain_bf16_kernel(const ChainBParams P)
{
    constexpr int NT = NWAVE * 64;
    constexpr int SL = X3 ? 1024 : 512;                  // 16-B units per 64x64 slab (hi [+ lo])
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* sWp = reinterpret_cast<u32x4*>(smem);
    u32x4* sW1 = sWp + SL;                               // 4 slabs
    u32x4* sW2 = sW1 + 4 * SL;                           // 4 slabs
    float* sB = reinterpret_cast<float*>(sW2 + 4 * SL);  // bp (64) | b1 (256) | b2 (64)
    const int tid = threadIdx.x;
    for (int i = tid; i < SL; i += NT) sWp[i] = P.wp_s[i];
    for (int i = tid; i < 4 * SL; i += NT) { sW1[i] = P.w1_s[i]; sW2[i] = P.w2_s[i]; }
    for (int i = tid; i < 64; i += NT) { sB[i] = P.bp[i]; sB[64 + HID + i] = P.b2[i]; }
    for (int i = tid; i < HID; i += NT) sB[64 + i] = P.b1[i];
    __syncthreads();

    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int prow = l31 & (H - 1);

    for (int wt = blockIdx.x * NWAVE + wv; wt < P.wtiles; wt += gridDim.x * NWAVE) {
        const size_t row = (size_t)wt * 32 + l31;
        int opaque = 0;                                  // as in dgab_chain_kernel: keep the weight reads inside the loop
        if constexpr (OPQ) asm volatile("" : "+s"(opaque));
        const u32x4* tWp = sWp + opaque;
        const u32x4* tW1 = sW1 + opaque;
        const u32x4* tW2 = sW2 + opaque;
        const float* tB = sB + opaque;
        const float* tg2 = P.g2 + opaque;
        const float* tbe2 = P.be2 + opaque;
        const float* xr = P.x + row * W;
        // bf16 gated map: already the operand, 8 consecutive bf16 per k-step; fp32 (X3): split while loading
        u32x4 ab[4], abl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (X3) {
                const float* ap = reinterpret_cast<const float*>(P.a) + row * W + 16 * j + 8 * half;
                const float4 v0 = *reinterpret_cast<const float4*>(ap), v1 = *reinterpret_cast<const float4*>(ap + 4);
                unsigned h[4], l[4];
                split2(v0.x, v0.y, h[0], l[0]); split2(v0.z, v0.w, h[1], l[1]);
                split2(v1.x, v1.y, h[2], l[2]); split2(v1.z, v1.w, h[3], l[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { ab[j][e] = h[e]; abl[j][e] = l[e]; }
            } else {
                ab[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(P.a) + row * W + 16 * j + 8 * half);
            }
        }
        float x1[32];
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {                 // g8 = 4t + g
            const float4 vx = *reinterpret_cast<const float4*>(xr + 32 * (g8 >> 2) + 8 * (g8 & 3) + 4 * half);
            x1[4 * g8] = vx.x; x1[4 * g8 + 1] = vx.y; x1[4 * g8 + 2] = vx.z; x1[4 * g8 + 3] = vx.w;
        }
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        if constexpr (X3) gemm64b3(tWp, ab, abl, half, l31, acc);
        else gemm64b(tWp, ab, half, l31, acc);
        // ---- x1 = x + proj(A) ----
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int f = 32 * (i >> 4) + 8 * ((i & 15) >> 2) + (i & 3) + 4 * half;
            x1[i] = x1[i] + (acc[i >> 4][i & 15] + tB[f]);
        }
        // ---- LayerNorm over each 16x64 plane (32 lanes x 32 registers), two-pass, fp32 ----
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x1[i];
        const float mean = plane_sum(s) * (1.0f / (H * W));
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float d = x1[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(plane_sum(q) * (1.0f / (H * W)) + kEps);
        u32x4 xb[4], xbl[4];
        {
            float xn[32];
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) {
                const int f = 32 * (g8 >> 2) + 8 * (g8 & 3) + 4 * half;
                const float4 gg = *reinterpret_cast<const float4*>(tg2 + prow * W + f);
                const float4 bb = *reinterpret_cast<const float4*>(tbe2 + prow * W + f);
                xn[4 * g8] = (x1[4 * g8] - mean) * rstd * gg.x + bb.x;
                xn[4 * g8 + 1] = (x1[4 * g8 + 1] - mean) * rstd * gg.y + bb.y;
                xn[4 * g8 + 2] = (x1[4 * g8 + 2] - mean) * rstd * gg.z + bb.z;
                xn[4 * g8 + 3] = (x1[4 * g8 + 3] - mean) * rstd * gg.w + bb.w;
            }
            if constexpr (X3) to_operands3(xn, xb, xbl);
            else to_operands(xn, xb);
        }
        // ---- out = x1 + fc2(gelu(fc1(xn))): hidden units in 4 blocks of 64, never leaving registers ----
        f32x16 o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[t][i] = 0.0f;
#pragma unroll 1
        for (int hb = 0; hb < 4; ++hb) {
            f32x16 hh[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[t][i] = 0.0f;
            if constexpr (X3) gemm64b3(tW1 + hb * SL, xb, xbl, half, l31, hh);
            else gemm64b(tW1 + hb * SL, xb, half, l31, hh);
            float hid[32];
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const int f = 32 * (i >> 4) + 8 * ((i & 15) >> 2) + (i & 3) + 4 * half;      // f and f + 1: one 8-byte bias read
                const f32x2 bb2 = *reinterpret_cast<const f32x2*>(tB + 64 + hb * 64 + f);
                f32x2 zz; zz[0] = hh[i >> 4][i & 15]; zz[1] = hh[i >> 4][(i & 15) + 1];
                const f32x2 g2v = gelu_as2(zz + bb2);
                hid[i] = g2v[0]; hid[i + 1] = g2v[1];
            }
            u32x4 hb4[4], hb4l[4];
            if constexpr (X3) {
                to_operands3(hid, hb4, hb4l);
                gemm64b3(tW2 + hb * SL, hb4, hb4l, half, l31, o);
            } else {
                to_operands(hid, hb4);
                gemm64b(tW2 + hb * SL, hb4, half, l31, o);
            }
        }
        float* orow = P.out + row * W;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
            const int f = 32 * (g8 >> 2) + 8 * (g8 & 3) + 4 * half;
            float4 v;
            v.x = x1[4 * g8] + (o[g8 >> 2][4 * (g8 & 3)] + tB[64 + HID + f]);
            v.y = x1[4 * g8 + 1] + (o[g8 >> 2][4 * (g8 & 3) + 1] + tB[64 + HID + f + 1]);
            v.z = x1[4 * g8 + 2] + (o[g8 >> 2][4 * (g8 & 3) + 2] + tB[64 + HID + f + 2]);
            v.w = x1[4 * g8 + 3] + (o[g8 >> 2][4 * (g8 & 3) + 3] + tB[64 + HID + f + 3]);
            *reinterpret_cast<float4*>(orow + f) = v;
        }
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_dgab_fwd(const float* x, const float* y, const float* ln1_w, const float* ln1_b,
                                const float* mlp_w_t, const float* mlp_h_t, const float* proj_slab,
                                const float* proj_b, const float* ln2_w, const float* ln2_b,
                                const float* fc1_slab, const float* fc1_b, const float* fc2_slab,
                                const float* fc2_b, float* scratch, float* out, int N, int C,
                                tpspp_stream_t stream)
{
    TPSPP_REQUIRE(x && y && ln1_w && ln1_b && mlp_w_t && mlp_h_t && proj_slab && proj_b && ln2_w && ln2_b &&
                  fc1_slab && fc1_b && fc2_slab && fc2_b && scratch && out, "tpspp_dgab_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C > 0 && (C % 8) == 0, "tpspp_dgab_fwd: channels must be a multiple of 8");
    if (N == 0) return TPSPP_OK;
    hipStream_t st = tpspp::as_stream(stream);
    const int planes = N * C;
    GateParams G;
    G.x = x; G.y = y; G.g1 = ln1_w; G.b1 = ln1_b; G.mw_t = mlp_w_t; G.mh_t = mlp_h_t; G.a = scratch;
    G.planes = planes;
    hipLaunchKernelGGL(dgab_gate_kernel<false>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, st, G);
    int rc = tpspp::check_launch("tpspp_dgab_fwd(gate)");
    if (rc) return rc;
    ChainParams Q;
    Q.x = x; Q.a = scratch; Q.wp_s = proj_slab; Q.w1_s = fc1_slab; Q.w2_s = fc2_slab;
    Q.bp = proj_b; Q.b1 = fc1_b; Q.b2 = fc2_b; Q.g2 = ln2_w; Q.be2 = ln2_b; Q.out = out;
    Q.wtiles = planes / 2;                                          // 2 planes = 32 rows per wavefront tile
    constexpr int kChainWaves = 8;
    const size_t lds = (size_t)(64 * 64 + 2 * 4 * 64 * 64 + 64 + HID + 64) * sizeof(float);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgab_chain_kernel<kChainWaves>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const int wgs = (Q.wtiles + kChainWaves - 1) / kChainWaves;
    const int grid = wgs < 256 ? wgs : 256;                         // persistent: one workgroup per CU
    hipLaunchKernelGGL(dgab_chain_kernel<kChainWaves>, dim3((unsigned)grid), dim3(kChainWaves * 64), lds, st, Q);
    return tpspp::check_launch("tpspp_dgab_fwd(chain)");
}

TPSPP_EXPORT int tpspp_dgab_bf16_fwd(const float* x, const float* y, const float* ln1_w, const float* ln1_b,
                                     const float* mlp_w_t, const float* mlp_h_t, const void* proj_slab,
                                     const float* proj_b, const float* ln2_w, const float* ln2_b,
                                     const void* fc1_slab, const float* fc1_b, const void* fc2_slab,
                                     const float* fc2_b, void* scratch, float* out, int N, int C, int split3,
                                     tpspp_stream_t stream)
{
    TPSPP_REQUIRE(x && y && ln1_w && ln1_b && mlp_w_t && mlp_h_t && proj_slab && proj_b && ln2_w && ln2_b &&
                  fc1_slab && fc1_b && fc2_slab && fc2_b && scratch && out, "tpspp_dgab_bf16_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C > 0 && (C % 8) == 0, "tpspp_dgab_bf16_fwd: channels must be a multiple of 8");
    if (N == 0) return TPSPP_OK;
    hipStream_t st = tpspp::as_stream(stream);
    const int planes = N * C;
    GateParams G;
    G.x = x; G.y = y; G.g1 = ln1_w; G.b1 = ln1_b; G.mw_t = mlp_w_t; G.mh_t = mlp_h_t; G.a = scratch;
    G.planes = planes;
    if (split3) hipLaunchKernelGGL(dgab_gate_kernel<false>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, st, G);
    else hipLaunchKernelGGL(dgab_gate_kernel<true>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, st, G);
    int rc = tpspp::check_launch("tpspp_dgab_bf16_fwd(gate)");
    if (rc) return rc;
    ChainBParams Q;
    Q.x = x; Q.a = scratch;
    Q.wp_s = static_cast<const u32x4*>(proj_slab); Q.w1_s = static_cast<const u32x4*>(fc1_slab);
    Q.w2_s = static_cast<const u32x4*>(fc2_slab);
    Q.bp = proj_b; Q.b1 = fc1_b; Q.b2 = fc2_b; Q.g2 = ln2_w; Q.be2 = ln2_b; Q.out = out;
    Q.wtiles = planes / 2;
    constexpr int kW3 = 8, kW1 = 4;
    const size_t lds = (size_t)(9 * (split3 ? 1024 : 512)) * 16 + (size_t)(64 + HID + 64) * sizeof(float);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgab_chain_bf16_kernel<false, kW1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgab_chain_bf16_kernel<true, kW3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    if (split3) {
        const int wgs = (Q.wtiles + kW3 - 1) / kW3;
        const int grid = wgs < 256 ? wgs : 256;                     // persistent: one workgroup per CU
        hipLaunchKernelGGL((dgab_chain_bf16_kernel<true, kW3>), dim3((unsigned)grid), dim3(kW3 * 64), lds, st, Q);
    } else {
        // bf16: the 128 GELUs per lane bound this kernel on the vector ALU; more wavefronts (6 or 8 per workgroup,
        // with the weights re-read per tile so they fit) measured 244 / 220 us against 215 us for this form
        const int wgs = (Q.wtiles + kW1 - 1) / kW1;
        const int grid = wgs < 512 ? wgs : 512;                     // persistent: two workgroups per CU
        hipLaunchKernelGGL((dgab_chain_bf16_kernel<false, kW1, false>), dim3((unsigned)grid), dim3(kW1 * 64), lds, st, Q);
    }
    return tpspp::check_launch("tpspp_dgab_bf16_fwd(chain)");
}
