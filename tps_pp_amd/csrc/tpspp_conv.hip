// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// every product rounded once, k-ordered accumulation) for the TPS++ feature extractor and the
// recognizer's conv stem: 1x1 and 3x3 kernels, strides 1/2/(2,1), "same" padding, up to three
// channel-concatenated sources each with its own integer nearest-neighbour upsampling (so
// `cat(a1, a2, Upsample(a3))` and `Upsample -> conv` never materialise), fused bias + ReLU and an
// optional residual added before or after the activation.
//
// GEMM view:  D[cout][pixel] = W[cout][k] * X[k][pixel],  k = (ci*KH + ky)*KW + kx.
// The weights are the MFMA A operand and the im2col matrix the B operand, so that in the C/D layout
// (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) the 32 lanes of a half-wavefront hold 32
// CONSECUTIVE PIXELS of one output channel: stores are 128-B coalesced rows of the NCHW output.
//
// Workgroup = 256 threads = 4 wavefronts = 128 output pixels x 64 output channels; wavefront w owns
// pixels [32w, 32w+32) and both 32-channel halves (two 32x32 accumulators, the X fragment is shared).
// K is consumed in chunks of KC input channels (KC*KH*KW values of k): the weight slab (k-major,
// pre-transposed once on the host) and the im2col slab are staged in LDS k-major, so a fragment read
// is 32 consecutive floats per half-wavefront (conflict-free).  LDS per workgroup <= 55 KB: two
// workgroups per CU overlap one's staging (VALU address arithmetic + gathers through L1/L2, 9x
// reuse of every input element for a 3x3 kernel) with the other's MFMA phase.
//
// Replaces (reference, mmocr/models/textrecog/): mmcv ConvModule / nn.Conv2d call sites
// backbones/tps_pp/tps_pp.py:126-131,149-154,538-552,560-562; preprocessor/tps_preprocessor.py:101-128;
// backbones/resnet_v2_large.py:131-135 and layers/conv_layer.py:12-33 (BatchNorm folded on the host).
// Bound: MFMA (fp32 matrix rate = 157 TFLOP/s peak).
#include "tpspp_common.h"
#include <type_traits>

namespace {

constexpr int kWave = 64;
constexpr int BM = 128;       // output pixels per workgroup
constexpr int BN = 64;        // output channels per workgroup
constexpr int kThreads = 256;
#ifndef KC3
#define KC3 4
#endif
constexpr int kKC3 = KC3;       // input channels per K-chunk of the 3x3 kernels (layout of weight_tiled)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvSrc {
    const float* p;
    int C, H, W;              // stored size
    int uh, uw;               // nearest upsampling factors: logical size (H*uh, W*uw)
};

struct ConvParams {
    ConvSrc src[3];
    int nsrc;
    const float* wt;          // (K, Cout), k-major; K = Cin*KH*KW
    const float* bias;        // (Cout) or null
    const float* res;         // (N, Cout, Ho, Wo) or null
    const float* post_scale;  // (Cout) or null: out = act(...) * post_scale + post_shift  (BatchNorm AFTER
    const float* post_shift;  //                 the activation, nrtr_modality_transformer.py:42-48)
    float* out;               // (N, Cout, Ho, Wo)
    int N, Cin, Cout, Hi, Wi, Ho, Wo, sh, sw, ph, pw;
    int relu;                 // activation: 0 none, 1 ReLU, 2 GELU (exact erf form, nn.GELU default)
    int res_mode;             // 1: out = act(conv + bias) + res; 2: out = act(conv + bias + res)
};

template <int KH, int KW, int KC>
__global__ void __launch_bounds__(kThreads, 2)
conv_igemm_f32_kernel(const ConvParams P)
{
    constexpr int TAPS = KH * KW;
    constexpr int KCK = KC * TAPS;                 // k values per chunk (even)
    static_assert(KCK % 2 == 0, "chunk must hold an even number of k");
    __shared__ float sX[KCK][BM];                  // im2col slab,  k-major
    __shared__ float sW[KCK][BN];                  // weight slab,  k-major

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int n = blockIdx.z;
    const int co_base = blockIdx.y * BN;
    const int m0 = blockIdx.x * BM;
    const int HoWo = P.Ho * P.Wo;

    // this thread's pixel for the staging pass (fixed: BM = 128, 256 threads -> two k rows per pass)
    const int sm = tid & (BM - 1);
    const int skk0 = tid >> 7;                     // 0 or 1
    const int spix = m0 + sm;
    const bool spix_ok = spix < HoWo;
    const int soy = spix_ok ? spix / P.Wo : 0;
    const int sox = spix_ok ? spix - soy * P.Wo : 0;
    const int iy0 = soy * P.sh - P.ph, ix0 = sox * P.sw - P.pw;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }

    const int half = lane >> 5, l31 = lane & 31;
    int cbase = 0;                                 // first channel of the current source
    int s = 0;
    ConvSrc cur = P.src[0];
    for (int c0 = 0; c0 < P.Cin; c0 += KC) {
        while (c0 >= cbase + cur.C) { cbase += cur.C; ++s; cur = P.src[s]; }
        const float* sp = cur.p + ((size_t)n * cur.C + (c0 - cbase)) * cur.H * cur.W;
        const int plane = cur.H * cur.W;
        // ---- stage the im2col slab: sX[kk][m] = in[c0 + kk/TAPS][iy][ix] (0 outside) ----
#pragma unroll
        for (int kk = skk0; kk < KCK; kk += 2) {
            const int ci = kk / TAPS, t = kk - ci * TAPS;
            const int ky = t / KW, kx = t - ky * KW;
            const int iy = iy0 + ky, ix = ix0 + kx;
            float v = 0.0f;
            if (spix_ok && iy >= 0 && iy < P.Hi && ix >= 0 && ix < P.Wi && (c0 + ci) < P.Cin)
                v = sp[(size_t)ci * plane + (iy / cur.uh) * cur.W + (ix / cur.uw)];
            sX[kk][sm] = v;
        }
        // ---- stage the weight slab: sW[kk][co] = wt[(c0*TAPS + kk)][co_base + co] ----
        {
            const float* wp = P.wt + (size_t)c0 * TAPS * P.Cout + co_base;
            const int kmax = min(KCK, (P.Cin - c0) * TAPS);
            for (int e = tid; e < KCK * BN; e += kThreads) {
                const int kk = e / BN, co = e - kk * BN;
                sW[kk][co] = (kk < kmax && co_base + co < P.Cout) ? wp[(size_t)kk * P.Cout + co] : 0.0f;
            }
        }
        __syncthreads();
        // ---- MFMA: 2 k per instruction; lanes 0-31 take k, lanes 32-63 take k+1 ----
#pragma unroll 4
        for (int k2 = 0; k2 < KCK; k2 += 2) {
            const float b = sX[k2 + half][wv * 32 + l31];
            const float a0 = sW[k2 + half][l31];
            const float a1 = sW[k2 + half][32 + l31];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: bias, residual, ReLU; half-wavefronts store 128-B rows ----
    const int pix = m0 + wv * 32 + l31;
    if (pix < HoWo) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + 32 * h2 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < P.Cout) {
                    float v = h2 ? acc1[r] : acc0[r];
                    if (P.bias) v = v + P.bias[co];
                    const size_t o = ((size_t)n * P.Cout + co) * HoWo + pix;
                    if (P.res_mode == 2) v = v + P.res[o];
                    if (P.relu == 1) v = v > 0.0f ? v : 0.0f;
                    else if (P.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (P.res_mode == 1) v = v + P.res[o];
                    if (P.post_scale) v = v * P.post_scale[co] + P.post_shift[co];
                    P.out[o] = v;
                }
            }
        }
    }
}

// ---- tiled kernel: the input PATCH of a rectangular output tile is staged once per channel chunk ---
// (coalesced row copies, every input element fetched once instead of KH*KW times) and the im2col
// matrix is never materialised: with k ordered (tap, channel) inside a chunk, the B fragment of lane
// (pixel (ty,tx), half h) for tap (ky,kx), channel pair c2 is
//     patch[2*c2 + h][ty*SH + ky][tx*SW + kx]  =  lane_base + compile-time offset,
// i.e. one ds_read_b32 with an immediate and no address arithmetic in the MFMA loop.  The next
// chunk's patch and weight slab are prefetched into registers while the current chunk is multiplied.
// Requires weights pre-arranged as (chunk, tap, channel-in-chunk, cout) -- tpspp_conv_arrange_weight.
template <int KH, int SH, int SW, int TH, int TW, int KC, int NI = 1>
struct TileCfg {
    static constexpr int KW = KH;
    static constexpr int TAPS = KH * KW;
    // (a strided 1x1 layer stages ONLY the pixels it uses -- patch = tile, source pixel (SH y, SW x): round 6)
    static constexpr int PH = KH == 1 ? TH : (TH - 1) * SH + KH;      // patch rows
    static constexpr int PW = KH == 1 ? TW : (TW - 1) * SW + KW;      // patch cols
    static constexpr int PS = PH * PW;                 // floats per channel of the patch
    static constexpr int KCK = KC * TAPS;
    static constexpr int PATCH1 = KC * PS;             // one image's patch
    static constexpr int PATCH = NI * PATCH1;          // NI images per tile (small maps: 4x16 outputs)
    static constexpr int NP = (PATCH + kThreads - 1) / kThreads;          // patch floats per thread
    static constexpr int NW4 = (KCK * BN / 4 + kThreads - 1) / kThreads;  // weight float4 per thread
};

// NB (round 6): accumulators per wavefront -- 2 = the 64-channel tile; 1 for layers with at most 32 output channels (the
// backbone's first stage, its stem): the second accumulator multiplied zeros there, half of the layer's matrix time.
// (round 6: the stride-2 3x3 tiles are compiled for THREE workgroups per CU -- with two as the bound the compiler spent 169 - 175
// registers on them, i.e. two waves per SIMD; 168 is three.  512 -> 512 stride 2: 1622 -> 1417 us.  The stride-1 tiles use
// 116 - 124 registers (four workgroups per CU) whatever the bound says, and measured 1.3 % slower when compiled for three.
// The 1x1 tiles are compiled for four: they sat at 128 registers until an unrelated edit made them 133.)
template <int KH, int SH, int SW, int TH, int TW, int KC, int NI = 1, int NB = 2>
__global__ void __launch_bounds__(kThreads, KH == 1 ? 4 : (KH == 3 && SH == 2 && SW == 2) ? 3 : 2)
conv_tiled_f32_kernel(const ConvParams P)
{
    using Cfg = TileCfg<KH, SH, SW, TH, TW, KC, NI>;
    constexpr int KW = Cfg::KW, TAPS = Cfg::TAPS, PH = Cfg::PH, PW = Cfg::PW, PS = Cfg::PS;
    constexpr int KCK = Cfg::KCK, PATCH = Cfg::PATCH, PATCH1 = Cfg::PATCH1, NP = Cfg::NP, NW4 = Cfg::NW4;
    static_assert(NI * TH * TW == BM, "tile must hold 128 pixels");
    static_assert(KC % 2 == 0, "channel pairs");
    __shared__ __attribute__((aligned(16))) float sP[PATCH];
    __shared__ __attribute__((aligned(16))) float sW[KCK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int ctiles = (P.Cout + BN - 1) / BN;
    const int ngrp = blockIdx.z / ctiles;
    const int n0 = ngrp * NI;                                   // first image of this tile
    const int co_base = (blockIdx.z - ngrp * ctiles) * BN;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int iy_base = oy0 * SH - P.ph, ix_base = ox0 * SW - P.pw;
    const int HoWo = P.Ho * P.Wo;

    // this lane's pixel inside the tile and its B-fragment base address in the patch
    const int tp = wv * 32 + l31;
    const int timg = tp / (TH * TW), tpi = tp - timg * (TH * TW);
    const int ty = tpi / TW, tx = tpi - ty * TW;
    const int n = n0 + timg;                                    // the image this lane's pixel belongs to
    const int lane_base = timg * PATCH1 + half * PS + (KH == 1 ? ty * PW + tx : ty * SH * PW + tx * SW);

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }

    float rp[NP];
    float4 rw[NW4];
    int cbase = 0, s = 0;
    ConvSrc cur = P.src[0];
    const int nchunks = (P.Cin + KC - 1) / KC;

    // staging: this thread's patch elements are the same in every chunk -- logical input row (or -1: padding / outside
    // the tile / beyond the batch), column, channel inside the chunk and image are worked out once.  (Decoding them, two
    // integer divisions by the upsampling factors and a predicated load per element and chunk made the staging 8.7 vector
    // instructions per MFMA -- 5 k per wavefront in a 64 -> 64 layer, `SQ_INSTS_VALU` -- and held the matrix pipe at 60 %.)
    // LEAN (round 6, the stride-2 3x3 tiles: 2.5x the patch elements per thread): only the element's offset and one validity bit
    // are kept -- row, column, channel and image are worked out again when a new source is entered (and per chunk on the
    // upsampling paths, which no stride-2 layer of the model takes) -- so that these tiles fit the 168 registers of three
    // workgroups per CU (see __launch_bounds__ above) without spilling.
    constexpr bool LEAN = KH == 3 && SH == 2 && SW == 2;
    static_assert(NP <= 32, "one validity bit per staged element");
    int t_iy[LEAN ? 1 : NP], t_ix[LEAN ? 1 : NP], t_ci[LEAN ? 1 : NP], t_im[LEAN ? 1 : NP];
    unsigned t_off[NP];
    unsigned vmask = 0u;
    int t_src = -1;                                                   // the source t_off was computed for
    auto decode = [&](int i, int& iy, int& ix, int& ci, int& im) -> bool {
        const int e = tid + i * kThreads;
        im = NI > 1 ? e / PATCH1 : 0;
        const int e1 = e - im * PATCH1;
        ci = e1 / PS;
        const int r = e1 - ci * PS;
        const int py = r / PW, px = r - py * PW;
        iy = KH == 1 ? (oy0 + py) * SH : iy_base + py;
        ix = KH == 1 ? (ox0 + px) * SW : ix_base + px;
        return e < PATCH && iy >= 0 && iy < P.Hi && ix >= 0 && ix < P.Wi && (n0 + im) < P.N;
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int iy, ix, ci, im;
        const bool ok = decode(i, iy, ix, ci, im);
        if constexpr (LEAN) { if (ok) vmask |= 1u << i; }
        else { t_iy[i] = ok ? iy : -1; t_ix[i] = ix; t_ci[i] = ci; t_im[i] = im; }
    }
    // element i: valid?  row / column / channel / image (LEAN: worked out again)
    auto el_ok = [&](int i) -> bool { if constexpr (LEAN) return (vmask >> i) & 1u; else return t_iy[i] >= 0; };
    auto el = [&](int i, int& iy, int& ix, int& ci, int& im) {
        if constexpr (LEAN) { (void)decode(i, iy, ix, ci, im); }
        else { iy = t_iy[i]; ix = t_ix[i]; ci = t_ci[i]; im = t_im[i]; }
    };

    auto prefetch = [&](int chunk) {
        const int c0 = chunk * KC;
        while (c0 >= cbase + cur.C) { cbase += cur.C; ++s; cur = P.src[s]; }
        const int plane = cur.H * cur.W;
        const float* sp = cur.p + ((size_t)n0 * cur.C + (c0 - cbase)) * plane;       // uniform
        const int img_stride = cur.C * plane;
        const int cleft = P.Cin - c0;                                 // channels of the concatenation left from c0 on
        const bool pow2 = ((cur.uh & (cur.uh - 1)) | (cur.uw & (cur.uw - 1))) == 0;   // uniform
        // every load is unconditional (a predicate per load puts each one in its own basic block behind an s_waitcnt):
        // invalid elements read the chunk's first element and are zeroed by a select
        if (cur.uh == 1 && cur.uw == 1 && cleft >= KC) {
            // the common case -- a source at full resolution, a whole chunk: the element's offset from the chunk's first
            // channel does not depend on the chunk (t_off, worked out when the source was entered), so an element is one
            // load and one select
            if constexpr (KH == 1) {
                // (the 1x1 tiles keep the plain form: through the el() / el_ok() helpers below they compiled to 133 registers
                // instead of 128 -- three wavefronts per SIMD instead of four, 94 against 81 us on the 64 -> 64 layers)
                if (t_src != s) {
                    t_src = s;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        t_off[i] = t_iy[i] >= 0 ? (unsigned)(t_im[i] * img_stride + t_ci[i] * plane + t_iy[i] * cur.W + t_ix[i]) : 0u;
                }
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const float v = sp[t_off[i]];
                    rp[i] = t_iy[i] >= 0 ? v : 0.0f;
                }
            } else {
            if (t_src != s) {
                t_src = s;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    int iy, ix, ci, im;
                    el(i, iy, ix, ci, im);
                    t_off[i] = el_ok(i) ? (unsigned)(im * img_stride + ci * plane + iy * cur.W + ix) : 0u;
                }
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float v = sp[t_off[i]];
                rp[i] = el_ok(i) ? v : 0.0f;
            }
            }
        } else if (pow2) {                                            // nearest upsampling by 1 / 2 / 4: shifts
            const int lh = 31 - __builtin_clz(cur.uh), lw = 31 - __builtin_clz(cur.uw);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int iy, ix, ci, im;
                el(i, iy, ix, ci, im);
                const bool ok = el_ok(i) && ci < cleft;
                const unsigned off = ok ? (unsigned)(im * img_stride + ci * plane + (iy >> lh) * cur.W + (ix >> lw)) : 0u;
                const float v = sp[off];
                rp[i] = ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int iy, ix, ci, im;
                el(i, iy, ix, ci, im);
                const bool ok = el_ok(i) && ci < cleft;
                const unsigned off = ok ? (unsigned)(im * img_stride + ci * plane + (iy / cur.uh) * cur.W + ix / cur.uw) : 0u;
                const float v = sp[off];
                rp[i] = ok ? v : 0.0f;
            }
        }
        // weights: (chunk, tap, ci, cout) -> this chunk's slab is KCK rows of Cout floats
        const float* wp = P.wt + (size_t)chunk * KCK * P.Cout + co_base;
#pragma unroll
        for (int i = 0; i < NW4; ++i) {
            const int e = tid + i * kThreads;                     // float4 index inside the slab
            const int kk = e / (BN / 4), c4 = e - kk * (BN / 4);
            const bool ok = kk < KCK && co_base + 4 * c4 + 3 < P.Cout;
            const float4 v = *reinterpret_cast<const float4*>(wp + (ok ? (size_t)kk * P.Cout + 4 * c4 : 0));
            rw[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {                  // (only a last, partial round of elements is predicated)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + i * kThreads;
            if ((i + 1) * kThreads <= PATCH || e < PATCH) sP[e] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < NW4; ++i) {
            const int e = tid + i * kThreads;
            if ((i + 1) * kThreads <= KCK * BN / 4 || e < KCK * BN / 4) reinterpret_cast<float4*>(sW)[e] = rw[i];
        }
    };

    prefetch(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        commit();
        __syncthreads();
        if (chunk + 1 < nchunks) prefetch(chunk + 1);     // in flight during the MFMA phase
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
            for (int c2 = 0; c2 < KC / 2; ++c2) {
                const float b = sP[lane_base + 2 * c2 * PS + ky * PW + kx];
                const float a0 = sW[(tap * KC + 2 * c2 + half) * BN + l31];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
                if constexpr (NB == 2) {
                    const float a1 = sW[(tap * KC + 2 * c2 + half) * BN + 32 + l31];
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----
    // Kept off the vector ALU where it can be (the lesson of the bf16 kernel, tpspp_conv_bf16.hip): wave-uniform base
    // (image group, channel tile: scalar arithmetic) + one 32-bit lane offset, the bias as float4 quads, and a common
    // path (no residual, no affine, no GELU) without per-element branches.
    const int oy = oy0 + ty, ox = ox0 + tx;
    const bool valid = oy < P.Ho && ox < P.Wo && n < P.N;
    const bool full_c = co_base + BN <= P.Cout;                  // uniform
    const bool simple = P.res_mode == 0 && P.post_scale == nullptr && P.relu != 2;
    const unsigned lo = valid ? (unsigned)((timg * P.Cout + 4 * half) * HoWo + oy * P.Wo + ox) : 0u;
    float* obase = P.out + ((size_t)n0 * P.Cout + co_base) * HoWo;     // uniform
    // two loops, not one test inside the unrolled loop: the general form (residual, inlined erff, affine) would otherwise
    // sit 32 times between the few instructions of the plain bias + ReLU layers (tpspp_conv_bf16_impl.h has the numbers)
    const bool relu1 = P.relu == 1;
    auto epilogue = [&](auto simple_c) {
        constexpr bool SIMPLE = decltype(simple_c)::value;
        // every residual value of the lane's 32 outputs is requested BEFORE the first result is stored (round 6): `out` may
        // alias `res` for all the compiler knows, so a load behind a store waits for it -- 32 memory latencies per workgroup,
        // 65 us of the 410 us of a 64 -> 64 layer on 16x64 maps (the lesson of tpspp_tokgemm.hip / tpspp_conv3_wide.hip)
        float rres[SIMPLE ? 1 : 32];
        if constexpr (!SIMPLE) {
    #pragma unroll
            for (int i = 0; i < 16 * NB; ++i) {
                const int h2 = i >> 4, g = (i >> 2) & 3, e = i & 3;
                const int cu = 32 * h2 + 8 * g;
                const bool ok = valid && (full_c || co_base + cu + 4 * half + e < P.Cout);
                rres[i] = (P.res_mode && ok) ? (P.res + ((size_t)n0 * P.Cout + co_base + cu + e) * HoWo)[lo] : 0.0f;
            }
        }
    #pragma unroll
        for (int h2 = 0; h2 < NB; ++h2) {
    #pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cu = 32 * h2 + 8 * g;                       // + 4*half (in `lo`) + e
                const int co4 = co_base + cu + 4 * half;
                float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (P.bias) {
                    if (full_c) {
                        const float4 b4 = *reinterpret_cast<const float4*>(P.bias + co4);
                        b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) b[e] = co4 + e < P.Cout ? P.bias[co4 + e] : 0.0f;
                    }
                }
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = (h2 ? acc1[4 * g + e] : acc0[4 * g + e]) + b[e];
                    float* op = obase + (size_t)(cu + e) * HoWo;       // uniform
                    const bool ok = valid && (full_c || co4 + e < P.Cout);
                    if constexpr (SIMPLE) {
                        if (relu1) v = v > 0.0f ? v : 0.0f;
                    } else {
                        const int co = co4 + e;
                        const float rv = rres[16 * h2 + 4 * g + e];
                        if (P.res_mode == 2) v = v + rv;
                        if (P.relu == 1) v = v > 0.0f ? v : 0.0f;
                        else if (P.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                        if (P.res_mode == 1) v = v + rv;
                        if (P.post_scale && co < P.Cout) v = v * P.post_scale[co] + P.post_shift[co];
                    }
                    if (ok) op[lo] = v;
                }
            }
        }
    };
    if (simple) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// ---- wide 1x1 kernel: large GEMMs (transformer projections over N*T tokens, 1x1 convs on big maps) -----------
// The 128x64 tile moves 24 KB from L2 per 0.5 MFLOP (21 FLOP/B): at the fp32 matrix rate that is more
// than the L2 can feed, and the 1x1 convolutions sit at ~60 TFLOP/s.  Here a workgroup owns 128 pixels x
// 128 output channels (wavefront w: pixels [32w, 32w+32) x all 128 channels, four accumulators sharing
// one B fragment): 32 KB per 1 MFLOP, 1.25 LDS reads per MFMA instead of 1.5.  Both slabs are k-major
// rows of 128 floats, staged with float4 loads (512-B rows); next chunk prefetched into registers during
// the multiply.  Needs HoWo % 4 == 0 and Cout % 4 == 0 (vector loads); same k order as the tiled kernel.
constexpr int WBN = 128;

__global__ void __launch_bounds__(kThreads, 2)
conv1x1_wide_f32_kernel(const ConvParams P)
{
    constexpr int KC = 32;
    __shared__ __attribute__((aligned(16))) float sX[KC * BM];
    __shared__ __attribute__((aligned(16))) float sW[KC * WBN];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int n = blockIdx.z;
    const int HoWo = P.Ho * P.Wo;
    const int m0 = blockIdx.x * BM, co_base = blockIdx.y * WBN;
    const float* xp = P.src[0].p + (size_t)n * P.Cin * HoWo;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;

    // staging: thread t loads float4 #(t % 32) of rows t/32 + 8*i  (i = 0..3) of both slabs
    const int srow = tid >> 5, sc4 = tid & 31;
    const bool x_ok = m0 + 4 * sc4 + 3 < HoWo, w_ok = co_base + 4 * sc4 + 3 < P.Cout;
    float4 rx[4], rw[4];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = c0 + srow + 8 * i;
            const bool k_ok = k < P.Cin;
            // unconditional loads (a predicated load sits in its own basic block behind an s_waitcnt); out-of-range
            // rows / columns re-read the slab's first float4 and are zeroed by the select
            const float4 vx = *reinterpret_cast<const float4*>(xp + ((k_ok && x_ok) ? (size_t)k * HoWo + m0 + 4 * sc4 : 0));
            const float4 vw = *reinterpret_cast<const float4*>(P.wt + ((k_ok && w_ok) ? (size_t)k * P.Cout + co_base + 4 * sc4 : 0));
            rx[i] = (k_ok && x_ok) ? vx : make_float4(0.f, 0.f, 0.f, 0.f);
            rw[i] = (k_ok && w_ok) ? vw : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            reinterpret_cast<float4*>(sX)[(srow + 8 * i) * (BM / 4) + sc4] = rx[i];
            reinterpret_cast<float4*>(sW)[(srow + 8 * i) * (WBN / 4) + sc4] = rw[i];
        }
    };
    prefetch(0);
    for (int c0 = 0; c0 < P.Cin; c0 += KC) {
        commit();
        __syncthreads();
        if (c0 + KC < P.Cin) prefetch(c0 + KC);
#pragma unroll
        for (int k2 = 0; k2 < KC; k2 += 2) {
            const float b = sX[(k2 + half) * BM + wv * 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = sW[(k2 + half) * WBN + 32 * j + l31];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int pix = m0 + wv * 32 + l31;
    // plain layers (bias, optional ReLU, every channel of the tile present) and the general form are two loops (see
    // conv_tiled_f32_kernel); the bias of a register quad is one float4
    const bool simple = P.res_mode == 0 && P.post_scale == nullptr && P.relu != 2 && co_base + WBN <= P.Cout && P.bias != nullptr;
    if (pix < HoWo && simple) {
        const bool relu1 = P.relu == 1;
        float* ob = P.out + ((size_t)n * P.Cout + co_base) * HoWo + pix;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 32 * j + 8 * g + 4 * half;
                const float4 b4 = *reinterpret_cast<const float4*>(P.bias + co_base + cl);
                const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][4 * g + e] + bq[e];
                    if (relu1) v = v > 0.0f ? v : 0.0f;
                    ob[(size_t)(cl + e) * HoWo] = v;
                }
            }
        }
    } else if (pix < HoWo) {
        // the residual of all 64 outputs of the lane before the first store (see conv_tiled_f32_kernel's epilogue): the encoder's
        // and decoder's projections with a residual in the exact-fp32 head take this path
        // (one 32-channel accumulator at a time: all 64 values in flight together push the kernel past 256 registers)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float rres[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
                rres[r] = (P.res_mode && co < P.Cout) ? P.res[((size_t)n * P.Cout + co) * HoWo + pix] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < P.Cout) {
                    float v = acc[j][r];
                    if (P.bias) v = v + P.bias[co];
                    const size_t o = ((size_t)n * P.Cout + co) * HoWo + pix;
                    if (P.res_mode == 2) v = v + rres[r];
                    if (P.relu == 1) v = v > 0.0f ? v : 0.0f;
                    else if (P.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (P.res_mode == 1) v = v + rres[r];
                    if (P.post_scale) v = v * P.post_scale[co] + P.post_shift[co];
                    P.out[o] = v;
                }
            }
        }
    }
}

bool wide_applies(const ConvParams& P, int KH)
{
    if (KH != 1 || P.nsrc != 1 || P.sh != 1 || P.sw != 1) return false;
    if (P.src[0].uh != 1 || P.src[0].uw != 1) return false;
    const int HoWo = P.Ho * P.Wo;
    if ((HoWo & 3) || (P.Cout & 3) || HoWo < BM || P.Cout < WBN) return false;
    const long wgs = (long)((HoWo + BM - 1) / BM) * ((P.Cout + WBN - 1) / WBN) * P.N;
    return wgs >= 512 && P.N <= 65535 && (P.Cout + WBN - 1) / WBN <= 65535;
}

void launch_wide(const ConvParams& P, hipStream_t st)
{
    const dim3 grid((unsigned)((P.Ho * P.Wo + BM - 1) / BM), (unsigned)((P.Cout + WBN - 1) / WBN), (unsigned)P.N);
    hipLaunchKernelGGL(conv1x1_wide_f32_kernel, grid, dim3(kThreads), 0, st, P);
}

// ---- skinny 1x1 kernel: few output pixels (one decoder step, a batch of feature vectors) -----------------
// With M = N*Ho*Wo of a few hundred the 128x64 tiles above leave most CUs idle and every K-chunk pays
// a full memory latency.  Here a workgroup owns a 32-pixel x 32-channel tile and its four wavefronts
// SPLIT K: each wavefront streams its quarter of the weight and activation rows straight from L2 into
// MFMA operands (both are k-major, so a fragment is two 128-B rows per load instruction; no LDS
// staging, all loads of a wavefront in flight together), the four partial tiles are summed through
// LDS and every wavefront finishes a quarter of the tile (bias / activation / residual as above).
// NW wavefronts split K (4, or 16 for K >= 256: the kernel is one dependent chain per wavefront -- operand rows from
// L2, then K/NW/2 dependent 64-cycle MFMAs --, so a 4x shorter chain is worth the larger reduction).
template <int NW>
__global__ void __launch_bounds__(NW * kWave)
conv1x1_skinny_f32_kernel(const ConvParams P)
{
    __shared__ float sRed[NW][16][kWave];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int half = lane >> 5, l31 = lane & 31;
    const int n = blockIdx.z;
    const int HoWo = P.Ho * P.Wo;
    const int m0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int K = P.Cin;
    const int Kw = ((K + 2 * NW - 1) / (2 * NW)) * 2;  // k values per wavefront (even)
    const int k_lo = wv * Kw;
    const int k_hi = min(K, k_lo + Kw);
    const int pix = m0 + l31, co = co0 + l31;
    const bool pix_ok = pix < HoWo, co_ok = co < P.Cout;
    const float* xp = P.src[0].p + (size_t)n * K * HoWo + (pix_ok ? pix : 0);
    const float* wp = P.wt + (co_ok ? co : 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    constexpr int UB = 16;                             // MFMA steps per batch: 2*UB row loads in flight
    const int nit = (k_hi - k_lo + 1) >> 1;            // wave-uniform trip count
    if (nit == UB && k_lo + 2 * UB <= K && co0 + 32 <= P.Cout && m0 + 32 <= HoWo) {
        // the decoder's shapes: one full batch, nothing to guard; with 4 wavefronts per SIMD every instruction of this
        // prologue is paid four times over (a trace put 1.9 k cycles before the last load's issue), so the row addresses
        // are two running 32-bit offsets instead of a 64-bit product per load
        float a[UB], b[UB];
        unsigned oa = (unsigned)(k_lo + half) * (unsigned)P.Cout, ob = (unsigned)(k_lo + half) * (unsigned)HoWo;
        const unsigned sa = 2u * (unsigned)P.Cout, sb = 2u * (unsigned)HoWo;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            a[u] = wp[oa]; b[u] = xp[ob];
            oa += sa; ob += sb;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    } else
    for (int it0 = 0; it0 < nit; it0 += UB) {
        float a[UB], b[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k_lo + 2 * (it0 + u) + half;
            const bool k_ok = k < k_hi;
            const int kk = k_ok ? k : k_lo;
            const float av = wp[(size_t)kk * P.Cout];
            const float bv = xp[(size_t)kk * HoWo];
            a[u] = (k_ok && co_ok) ? av : 0.0f;
            b[u] = (k_ok && pix_ok) ? bv : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);             // keep every load above the first MFMA
#pragma unroll
        for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[r];
    __syncthreads();
    // every wavefront finishes 16 / min(NW, 16) accumulator registers (NW = 16: one output per lane, so that the bias /
    // residual latencies of the sixteen registers overlap instead of following each other in four wavefronts);
    // register r holds channel co0 + (r & 3) + 8 (r >> 2) + 4 half
    constexpr int RPW = NW >= 16 ? 1 : 16 / NW;
    if (!pix_ok || wv * RPW >= 16) return;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = RPW * wv + q;
        const int c = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < P.Cout) {
            float v = (sRed[0][r][lane] + sRed[1][r][lane]) + (sRed[2][r][lane] + sRed[3][r][lane]);
#pragma unroll
            for (int j = 4; j < NW; j += 4)
                v += (sRed[j][r][lane] + sRed[j + 1][r][lane]) + (sRed[j + 2][r][lane] + sRed[j + 3][r][lane]);
            if (P.bias) v = v + P.bias[c];
            const size_t o = ((size_t)n * P.Cout + c) * HoWo + pix;
            if (P.res_mode == 2) v = v + P.res[o];
            if (P.relu == 1) v = v > 0.0f ? v : 0.0f;
            else if (P.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            if (P.res_mode == 1) v = v + P.res[o];
            if (P.post_scale) v = v * P.post_scale[c] + P.post_shift[c];
            P.out[o] = v;
        }
    }
}

// ---- LayerNorm fused into the skinny projection --------------------------------------------------------------------
// out = act(W^T LN(x) + bias) [+ res] for a channel-major x (K, M) with few columns (a decoder step).
// With gamma folded into the weight rows on the host (Wg = diag(gamma) W) the normalisation commutes with
// the product:   W^T LN(x)[:, m] = rstd_m (Wg^T x[:, m] - mean_m colsum(Wg)) + W^T beta,
// so the MFMAs run on the RAW activation while the same loads feed sum / sum-of-squares per column
// (two VALU FMAs per value); mean and rstd meet the accumulators in the epilogue.  The split-K workgroup
// touches all K values of its 32 columns exactly once (4 wavefronts x 2 k-parities), so the statistics
// cost one extra LDS row per wavefront in the reduction that is there anyway: no separate LayerNorm
// launch, no normalised copy of x in HBM.  var = E[x^2] - mean^2 in fp32 (inputs are O(1) activations).
// LN_ON_A = false: x is the B operand, output (Cout, M) channel-major;
// LN_ON_A = true : operands swapped, x is the A operand, output (M, Cout') token-major.
struct LnArgs {
    const float* colsum;       // (Cout): column sums of Wg
    float eps;
};

template <bool LN_ON_A, int NW>
__global__ void __launch_bounds__(NW * kWave)
conv1x1_skinny_ln_f32_kernel(const ConvParams P, const LnArgs L)
{
    __shared__ float sRed[NW][16][kWave];
    __shared__ float sSum[2 * NW][32], sSq[2 * NW][32];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int half = lane >> 5, l31 = lane & 31;
    const int n = blockIdx.z;
    const int HoWo = P.Ho * P.Wo;
    const int m0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int K = P.Cin;
    const int Kw = ((K + 2 * NW - 1) / (2 * NW)) * 2;
    const int k_lo = wv * Kw;
    const int k_hi = min(K, k_lo + Kw);
    const int pix = m0 + l31, co = co0 + l31;
    const bool pix_ok = pix < HoWo, co_ok = co < P.Cout;
    const float* xp = P.src[0].p + (size_t)n * K * HoWo + (pix_ok ? pix : 0);
    const float* wp = P.wt + (co_ok ? co : 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    constexpr int UB = 16;
    const int nit = (k_hi - k_lo + 1) >> 1;
    if (nit == UB && k_lo + 2 * UB <= K && co0 + 32 <= P.Cout && m0 + 32 <= HoWo) {      // as in conv1x1_skinny_f32_kernel
        float a[UB], b[UB];
        unsigned oa = (unsigned)(k_lo + half) * (unsigned)P.Cout, ob = (unsigned)(k_lo + half) * (unsigned)HoWo;
        const unsigned sa = 2u * (unsigned)P.Cout, sb = 2u * (unsigned)HoWo;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            a[u] = wp[oa]; b[u] = xp[ob];
            oa += sa; ob += sb;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float xv = LN_ON_A ? a[u] : b[u];
            s1 += xv;
            s2 = fmaf(xv, xv, s2);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
    } else
    for (int it0 = 0; it0 < nit; it0 += UB) {
        float a[UB], b[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k_lo + 2 * (it0 + u) + half;
            const bool k_ok = k < k_hi;
            const int kk = k_ok ? k : k_lo;
            const float av = wp[(size_t)kk * P.Cout];
            const float bv = xp[(size_t)kk * HoWo];
            a[u] = (k_ok && co_ok) ? av : 0.0f;
            b[u] = (k_ok && pix_ok) ? bv : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float xv = LN_ON_A ? a[u] : b[u];
            s1 += xv;
            s2 = fmaf(xv, xv, s2);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[r];
    sSum[wv * 2 + half][l31] = s1;
    sSq[wv * 2 + half][l31] = s2;
    __syncthreads();
    constexpr int RPW = NW >= 16 ? 1 : 16 / NW;          // accumulator registers finished per wavefront (see the plain kernel)
    if (!pix_ok || wv * RPW >= 16) return;
    float mean_l = 0.0f, rstd_l = 0.0f;
    if (!LN_ON_A) {                                        // statistics of this lane's pixel column
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 2 * NW; ++j) { t1 += sSum[j][l31]; t2 += sSq[j][l31]; }
        mean_l = t1 / (float)K;
        rstd_l = 1.0f / sqrtf(fmaxf(t2 / (float)K - mean_l * mean_l, 0.0f) + L.eps);
    }
#pragma unroll
    for (int qd = 0; qd < RPW; ++qd) {
        const int r = RPW * wv + qd;
        const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;    // row inside the tile
        const int c = co0 + cl;
        if (c < P.Cout) {
            float mean = mean_l, rstd = rstd_l, csum;
            if (LN_ON_A) {                                 // statistics belong to the ROW (a column of x)
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int j = 0; j < 2 * NW; ++j) { t1 += sSum[j][cl]; t2 += sSq[j][cl]; }
                mean = t1 / (float)K;
                rstd = 1.0f / sqrtf(fmaxf(t2 / (float)K - mean * mean, 0.0f) + L.eps);
                csum = L.colsum[pix];
            } else {
                csum = L.colsum[c];
            }
            float val = (sRed[0][r][lane] + sRed[1][r][lane]) + (sRed[2][r][lane] + sRed[3][r][lane]);
#pragma unroll
            for (int j = 4; j < NW; j += 4)
                val += (sRed[j][r][lane] + sRed[j + 1][r][lane]) + (sRed[j + 2][r][lane] + sRed[j + 3][r][lane]);
            val = rstd * (val - mean * csum);
            if (P.bias) val = val + P.bias[LN_ON_A ? pix : c];
            const size_t o = ((size_t)n * P.Cout + c) * HoWo + pix;
            if (P.res_mode == 2) val = val + P.res[o];
            if (P.relu == 1) val = val > 0.0f ? val : 0.0f;
            else if (P.relu == 2) val = 0.5f * val * (1.0f + erff(val * 0.70710678118654752440f));
            if (P.res_mode == 1) val = val + P.res[o];
            P.out[o] = val;
        }
    }
}

// true when the 128x64 tiling would start fewer workgroups than the chip has CUs
bool skinny_applies(const ConvParams& P, int KH)
{
    if (KH != 1 || P.nsrc != 1 || P.sh != 1 || P.sw != 1) return false;
    if (P.src[0].uh != 1 || P.src[0].uw != 1) return false;
    const long big = (long)((P.Ho * P.Wo + BM - 1) / BM) * ((P.Cout + BN - 1) / BN) * P.N;
    return big < 256 && P.N <= 65535 && (P.Cout + 31) / 32 <= 65535;
}

void launch_skinny(const ConvParams& P, hipStream_t st)
{
    const dim3 grid((unsigned)((P.Ho * P.Wo + 31) / 32), (unsigned)((P.Cout + 31) / 32), (unsigned)P.N);
    if (P.Cin >= 256) hipLaunchKernelGGL(conv1x1_skinny_f32_kernel<16>, grid, dim3(16 * kWave), 0, st, P);
    else              hipLaunchKernelGGL(conv1x1_skinny_f32_kernel<4>, grid, dim3(4 * kWave), 0, st, P);
}

template <int KH, int SH, int SW, int TH, int TW, int KC, int NI = 1>
void launch_tiled(const ConvParams& P, hipStream_t st)
{
    const int ctiles = (P.Cout + BN - 1) / BN;
    const dim3 grid((unsigned)((P.Wo + TW - 1) / TW), (unsigned)((P.Ho + TH - 1) / TH),
                    (unsigned)(((P.N + NI - 1) / NI) * ctiles));
    if (P.Cout <= 32) hipLaunchKernelGGL((conv_tiled_f32_kernel<KH, SH, SW, TH, TW, KC, NI, 1>), grid, dim3(kThreads), 0, st, P);
    else hipLaunchKernelGGL((conv_tiled_f32_kernel<KH, SH, SW, TH, TW, KC, NI>), grid, dim3(kThreads), 0, st, P);
}

// picks a tile for the output width; returns false when no tiled instantiation fits
bool launch_tiled_any(const ConvParams& P, int KH, hipStream_t st)
{
    if ((P.Cout % 4) != 0) return false;                          // float4 weight rows
    if ((long)P.N * ((P.Cout + BN - 1) / BN) > 65535) return false;
#define TPSPP_TILE(KHv, SHv, SWv, THv, TWv, KCv) { launch_tiled<KHv, SHv, SWv, THv, TWv, KCv>(P, st); return true; }
#define TPSPP_TILE2(KHv, SHv, SWv, THv, TWv, KCv) { launch_tiled<KHv, SHv, SWv, THv, TWv, KCv, 2>(P, st); return true; }
    // small maps (the last backbone stage: 4x16 outputs): a 128-pixel tile spans TWO images instead of
    // leaving half of its rows empty
    const bool small_map = P.Ho <= 4 && P.Wo <= 16 && P.N >= 2;
    if (KH == 1) {
        // strided 1x1 (the backbone's downsample branches; round 6): the tiled kernel, staging only the pixels the layer uses
        // (TileCfg), instead of the im2col kernel -- 256 -> 512 on 8x32 -> 4x16 maps ran at 27 TFLOP/s there
        if (P.sh == 2 && P.sw == 2) {
            if (small_map) TPSPP_TILE2(1, 2, 2, 4, 16, 32)
            if (P.Wo >= 64) TPSPP_TILE(1, 2, 2, 2, 64, 32)
            if (P.Wo >= 32) TPSPP_TILE(1, 2, 2, 4, 32, 32)
            TPSPP_TILE(1, 2, 2, 8, 16, 32)
        }
        if (P.sh != 1 || P.sw != 1) return false;
        if (small_map) TPSPP_TILE2(1, 1, 1, 4, 16, 32)
        if (P.Wo >= 128) TPSPP_TILE(1, 1, 1, 1, 128, 32)
        if (P.Wo >= 64) TPSPP_TILE(1, 1, 1, 2, 64, 32)
        if (P.Wo >= 32) TPSPP_TILE(1, 1, 1, 4, 32, 32)
        TPSPP_TILE(1, 1, 1, 8, 16, 32)
    }
    if (small_map && P.sh == 1 && P.sw == 1) TPSPP_TILE2(3, 1, 1, 4, 16, kKC3)
    if (small_map && P.sh == 2 && P.sw == 2) TPSPP_TILE2(3, 2, 2, 4, 16, kKC3)
    if (P.sh == 1 && P.sw == 1) {
        if (P.Wo >= 64) TPSPP_TILE(3, 1, 1, 2, 64, kKC3)
        if (P.Wo >= 32) TPSPP_TILE(3, 1, 1, 4, 32, kKC3)
        TPSPP_TILE(3, 1, 1, 8, 16, kKC3)
    }
    if (P.sh == 2 && P.sw == 2) {
        if (P.Wo >= 64) TPSPP_TILE(3, 2, 2, 2, 64, kKC3)
        if (P.Wo >= 32) TPSPP_TILE(3, 2, 2, 4, 32, kKC3)
        TPSPP_TILE(3, 2, 2, 8, 16, kKC3)
    }
    if (P.sh == 2 && P.sw == 1) TPSPP_TILE(3, 2, 1, 8, 16, kKC3)
#undef TPSPP_TILE
#undef TPSPP_TILE2
    return false;
}

template <int KH, int KW, int KC>
void launch_conv(const ConvParams& P, hipStream_t st)
{
    const dim3 grid((unsigned)((P.Ho * P.Wo + BM - 1) / BM), (unsigned)((P.Cout + BN - 1) / BN), (unsigned)P.N);
    hipLaunchKernelGGL((conv_igemm_f32_kernel<KH, KW, KC>), grid, dim3(kThreads), 0, st, P);
}

int g_conv_force_generic = 0;

}  // namespace

namespace tpspp { int g_conv_bf16_no_persist = 0; int g_conv_bf16_no_wide = 0; }      // read by tpspp_conv_bf16.hip

TPSPP_EXPORT int tpspp_conv_set_tuning(int flags)
{
    g_conv_force_generic = (flags & 1) ? 1 : 0;
    tpspp::g_conv_bf16_no_persist = (flags & 2) ? 1 : 0;
    tpspp::g_conv_bf16_no_wide = (flags & 4) ? 1 : 0;
    return TPSPP_OK;
}

TPSPP_EXPORT int tpspp_conv_chunk_channels(int kernel_size)
{
    return kernel_size == 1 ? 32 : kKC3;
}

TPSPP_EXPORT int tpspp_conv2d_fwd(const float* const* src_ptrs, const int* src_dims, int nsrc,
                                  const float* weight_t, const float* weight_tiled, const float* bias,
                                  const float* residual, const float* post_scale, const float* post_shift,
                                  int res_mode, int relu, int N, int Cout, int KH, int KW, int sh, int sw,
                                  float* out, int Ho, int Wo, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(src_ptrs && src_dims && (weight_t || weight_tiled) && out, "tpspp_conv2d_fwd: null pointer");
    TPSPP_REQUIRE(nsrc >= 1 && nsrc <= 3, "tpspp_conv2d_fwd: 1..3 sources");
    TPSPP_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3), "tpspp_conv2d_fwd: kernel must be 1x1 or 3x3");
    TPSPP_REQUIRE(N >= 0 && Cout > 0 && sh >= 1 && sw >= 1 && Ho > 0 && Wo > 0, "tpspp_conv2d_fwd: bad sizes");
    TPSPP_REQUIRE(res_mode >= 0 && res_mode <= 2 && (res_mode == 0) == (residual == nullptr),
                  "tpspp_conv2d_fwd: residual / res_mode mismatch");
    TPSPP_REQUIRE(N <= 65535 && (Cout + BN - 1) / BN <= 65535, "tpspp_conv2d_fwd: grid too large");
    ConvParams P;
    P.nsrc = nsrc;
    int cin = 0, Hi = -1, Wi = -1;
    const int KC = (KH == 1) ? 32 : kKC3;
    for (int i = 0; i < nsrc; ++i) {
        const int* d = src_dims + 5 * i;                      // C, H, W, uh, uw
        TPSPP_REQUIRE(src_ptrs[i] && d[0] > 0 && d[1] > 0 && d[2] > 0 && d[3] >= 1 && d[4] >= 1,
                      "tpspp_conv2d_fwd: bad source %d", i);
        TPSPP_REQUIRE(nsrc == 1 || d[0] % KC == 0,
                      "tpspp_conv2d_fwd: concatenated sources need channel counts that are multiples of %d", KC);
        P.src[i].p = src_ptrs[i];
        P.src[i].C = d[0]; P.src[i].H = d[1]; P.src[i].W = d[2]; P.src[i].uh = d[3]; P.src[i].uw = d[4];
        const int lh = d[1] * d[3], lw = d[2] * d[4];
        TPSPP_REQUIRE(Hi < 0 || (Hi == lh && Wi == lw), "tpspp_conv2d_fwd: sources disagree on the logical size");
        Hi = lh; Wi = lw;
        cin += d[0];
    }
    for (int i = nsrc; i < 3; ++i) P.src[i] = P.src[nsrc - 1];
    P.Cin = cin; P.Cout = Cout; P.N = N; P.Hi = Hi; P.Wi = Wi; P.Ho = Ho; P.Wo = Wo;
    P.sh = sh; P.sw = sw; P.ph = (KH - 1) / 2; P.pw = (KW - 1) / 2;
    TPSPP_REQUIRE(Ho == (Hi + 2 * P.ph - KH) / sh + 1 && Wo == (Wi + 2 * P.pw - KW) / sw + 1,
                  "tpspp_conv2d_fwd: output size does not match input size / stride ('same' padding)");
    TPSPP_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "tpspp_conv2d_fwd: post_scale/post_shift come together");
    P.wt = weight_t; P.bias = bias; P.res = residual; P.out = out;
    P.post_scale = post_scale; P.post_shift = post_shift;
    TPSPP_REQUIRE(relu >= 0 && relu <= 2, "tpspp_conv2d_fwd: activation code must be 0 (none), 1 (ReLU) or 2 (GELU)");
    P.relu = relu; P.res_mode = res_mode;
    if (N == 0) return TPSPP_OK;
    hipStream_t st = tpspp::as_stream(stream);
    if (g_conv_force_generic == 0 && skinny_applies(P, KH)) {
        ConvParams Q = P;
        Q.wt = weight_t ? weight_t : weight_tiled;         // identical layouts for a 1x1 kernel (k-major)
        launch_skinny(Q, st);
        return tpspp::check_launch("tpspp_conv2d_fwd(skinny)");
    }
    if (g_conv_force_generic == 0 && wide_applies(P, KH)) {
        ConvParams Q = P;
        Q.wt = weight_t ? weight_t : weight_tiled;
        launch_wide(Q, st);
        return tpspp::check_launch("tpspp_conv2d_fwd(wide)");
    }
    if (weight_tiled && g_conv_force_generic == 0) {
        ConvParams Q = P;
        Q.wt = weight_tiled;
        if (launch_tiled_any(Q, KH, st)) return tpspp::check_launch("tpspp_conv2d_fwd(tiled)");
    }
    TPSPP_REQUIRE(weight_t, "tpspp_conv2d_fwd: shape needs the generic kernel but weight_t is NULL");
    if (KH == 1) launch_conv<1, 1, 32>(P, st);
    else         launch_conv<3, 3, 8>(P, st);
    return tpspp::check_launch("tpspp_conv2d_fwd");
}

TPSPP_EXPORT int tpspp_linear_ln_fwd(const float* x, int K, int M, float eps, const float* w_gamma,
                                     const float* w_colsum, int Cout, const float* bias_eff, int act,
                                     const float* residual, int token_major, float* out, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(x && w_gamma && w_colsum && out, "tpspp_linear_ln_fwd: null pointer");
    TPSPP_REQUIRE(K > 0 && M > 0 && Cout > 0 && act >= 0 && act <= 2, "tpspp_linear_ln_fwd: bad sizes / activation");
    TPSPP_REQUIRE(!token_major || (act == 0 && !residual),
                  "tpspp_linear_ln_fwd: the token-major form takes no activation / residual");
    ConvParams P;
    P.nsrc = 1;
    P.src[0].C = K; P.src[0].H = 1; P.src[0].uh = 1; P.src[0].uw = 1;
    P.Cin = K; P.N = 1; P.Hi = 1; P.Ho = 1; P.sh = P.sw = 1; P.ph = P.pw = 0;
    P.bias = bias_eff; P.res = residual; P.res_mode = residual ? 1 : 0; P.relu = act;
    P.post_scale = nullptr; P.post_shift = nullptr; P.out = out;
    const LnArgs L{w_colsum, eps};
    hipStream_t st = tpspp::as_stream(stream);
    if (!token_major) {                    // out (Cout, M)
        P.src[0].p = x; P.src[0].W = M; P.Wi = M; P.Wo = M; P.Cout = Cout; P.wt = w_gamma;
        P.src[1] = P.src[0]; P.src[2] = P.src[0];
        const dim3 grid((unsigned)((M + 31) / 32), (unsigned)((Cout + 31) / 32), 1);
        TPSPP_REQUIRE(grid.y <= 65535, "tpspp_linear_ln_fwd: too many output features");
        // 16 wavefronts split K unless that leaves more workgroups than the chip holds at once (2 x 1024 threads per CU):
        // a second round costs more than the twice longer MFMA chain of an 8-way split
        const bool two_rounds = (long)grid.x * grid.y > 512;
        if (K >= 256 && !two_rounds) hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<false, 16>), grid, dim3(16 * kWave), 0, st, P, L);
        else if (K >= 256) hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<false, 8>), grid, dim3(8 * kWave), 0, st, P, L);
        else          hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<false, 4>), grid, dim3(4 * kWave), 0, st, P, L);
    } else {                               // out (M, Cout): the weight plays the image, x the weights
        P.src[0].p = w_gamma; P.src[0].W = Cout; P.Wi = Cout; P.Wo = Cout; P.Cout = M; P.wt = x;
        P.src[1] = P.src[0]; P.src[2] = P.src[0];
        const dim3 grid((unsigned)((Cout + 31) / 32), (unsigned)((M + 31) / 32), 1);
        TPSPP_REQUIRE(grid.y <= 65535, "tpspp_linear_ln_fwd: too many columns");
        const bool two_rounds = (long)grid.x * grid.y > 512;
        if (K >= 256 && !two_rounds) hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<true, 16>), grid, dim3(16 * kWave), 0, st, P, L);
        else if (K >= 256) hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<true, 8>), grid, dim3(8 * kWave), 0, st, P, L);
        else          hipLaunchKernelGGL((conv1x1_skinny_ln_f32_kernel<true, 4>), grid, dim3(4 * kWave), 0, st, P, L);
    }
    return tpspp::check_launch("tpspp_linear_ln_fwd");
}
