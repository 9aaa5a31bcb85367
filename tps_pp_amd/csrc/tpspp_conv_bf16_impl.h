// Kernel template + launch helpers of the bf16 / "bf16x3" convolution (see tpspp_conv_bf16.hip for the design notes).
// Included by tpspp_conv_bf16.hip (plain bf16 instantiations + the C entry point) and tpspp_conv_bf16x3.hip (the
// three-term-split instantiations): two translation units so that the ~70 instantiations compile in parallel.
#pragma once
#include "tpspp_common.h"
#include <type_traits>

typedef unsigned int tpspp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tpspp_u32x2 __attribute__((ext_vector_type(2)));
typedef short tpspp_s16x4 __attribute__((ext_vector_type(4)));

namespace tpspp {          // shared by the two translation units: external linkage

struct BSrc {
    const void* p;
    int C, H, W;              // stored size
    int lh, lw;               // log2 of the nearest upsampling factors
    int f32;                  // element type / layout: 0 bf16 NCHW, 1 fp32 NCHW, 2 bf16 blocked (N, C/8, H, W, 8),
                              // 3 fp32 blocked (the same shape in fp32: the three-term-split configuration's maps)
};

struct BParams {
    BSrc src[3];
    int nsrc;
    const tpspp_u32x4* wt;          // [ctile][chunk][tap][k group][64][8] bf16, 16-B units
    const float* bias;        // (Cout) fp32 or null
    const void* res;          // (N, Cout, Ho, Wo) or null
    const float* post_scale;
    const float* post_shift;
    void* out;
    int res_f32, out_f32;     // 0 bf16 NCHW, 1 fp32 NCHW, 2 bf16 blocked (N, C/8, H, W, 8), 3 fp32 blocked
    int N, Cin, Cout, Hi, Wi, Ho, Wo, ph, pw;
    int relu, res_mode;
    int nchunks;
};

}  // namespace tpspp

namespace {

using tpspp::BSrc;
using tpspp::BParams;

constexpr int kWave = 64;
constexpr int kThreads = 256;
constexpr int BM = 256;       // output pixels per workgroup (NF = 2 fragments per wavefront; 128 with NF = 1)
constexpr int BN = 64;        // output channels per workgroup

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ unsigned f32_to_bf16_bits(float f)
{
    // round to nearest even (inputs are finite activations)
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32: two fp32 -> packed bf16, round to nearest even
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h)
{
    return __builtin_bit_cast(float, (unsigned)h << 16);
}

template <int KH, int SH, int SW, int TH, int TW, int NI, int KC>
struct BCfg {
    static constexpr int KW = KH;
    static constexpr int TAPS = KH * KW;
    // a 1x1 kernel stages exactly the pixels it reads (one patch position per output pixel, input step =
    // stride); a 3x3 kernel stages the dense patch (input step 1, output pixels `stride` positions apart)
    static constexpr int OSH = KH == 1 ? 1 : SH, OSW = KH == 1 ? 1 : SW;   // patch positions per output step
    static constexpr int ISH = KH == 1 ? SH : 1, ISW = KH == 1 ? SW : 1;   // input pixels per patch position
    static constexpr int PH = (TH - 1) * OSH + KH;
    static constexpr int PW = (TW - 1) * OSW + KW;
    static constexpr int PS = PH * PW;                 // positions per image
    static constexpr int PSN = NI * PS;                // positions per tile
    static constexpr int NPOS = (PSN + kThreads - 1) / kThreads;
    static constexpr int KG = KC / 8;                  // channel groups of 8
    static constexpr int WSLAB = TAPS * KG * BN;       // 16-B units of a chunk's weight slab
    static constexpr int NW = (WSLAB + kThreads - 1) / kThreads;
};

// X3: "bf16x3" -- every fp32 operand is split into two bf16 halves (hi = bf16(x), lo = bf16(x - hi)) and a product
// is hi*hi + hi*lo + lo*hi, accumulated in fp32: 16-17 mantissa bits per operand (relative error ~5e-6 per layer
// against 3e-7 for fp32 and 2.5e-3 for plain bf16) at three matrix instructions that are each 16x faster than the
// fp32 one.  For the fp32 tensors of the parity-bound (1e-4) path: activations stay fp32 in HBM, the split happens
// in the staging registers; the weight arrives as two arranged slabs per chunk.
// NF: 32-pixel fragments per wavefront.  2 (a 256-pixel workgroup tile: one LDS fragment read per MFMA) is the default;
// 1 (128 pixels) halves the patch and the accumulators: more, lighter workgroups for the layers whose patch is large
// against their work (stride 2: the patch holds 4x the output positions) or whose maps are so small that 256-pixel tiles
// leave CUs idle -- those kernels wait on memory, not on the matrix pipe.
// WIDE (3x3, bf16 sources at full resolution, rows of whole 16-byte pieces): the patch is not gathered element by
// element (KC 2-byte loads and KC/2 packing instructions per position -- the instruction stream, not the memory,
// is what that staging is bound by: removing its loads takes 28 us off a 75 us layer that moves 30 us' worth of
// bytes) but arrives as 16-byte pieces of the NCHW rows, laid down as they come ([channel][image][row][aligned
// pixels]), and is brought into the channel-innermost patch image by ds_read_b64_tr_b16: a 16-lane group points at
// 4 channel rows x 16 pixels and each lane receives 4 consecutive channels of its pixel.  Per chunk and thread: 4-6
// loads of 16 bytes + ~4 x (2 transposing reads + 1 16-byte write) instead of 32 loads + 32 packs.
// NB (round 6): 32-channel accumulators per fragment -- 2 = the 64-channel tile; 1 for layers with at most 32 output channels
// (instantiated for the three-term split only: the bf16x3 backbone's first stage; its second accumulator multiplied zeros).
template <int KH, int SH, int SW, int TH, int TW, int NI, int KC, bool X3 = false, int NF = 2, bool WIDE = false, int NB = 2>
__global__ void __launch_bounds__(kThreads, 2)
conv_tiled_bf16_kernel(const BParams P)
{
    using Cfg = BCfg<KH, SH, SW, TH, TW, NI, KC>;
    constexpr int KW = Cfg::KW, TAPS = Cfg::TAPS, PW = Cfg::PW, PS = Cfg::PS, PSN = Cfg::PSN;
    constexpr int NPOS = Cfg::NPOS, KG = Cfg::KG, WSLAB = Cfg::WSLAB, NW = Cfg::NW;
    static_assert(NI * TH * TW == 128 * NF, "tile must hold 128 * NF pixels");
    static_assert(KC % 16 == 0, "whole MFMA k-steps");
    static_assert(!WIDE || (KH == 3 && !X3), "wide staging: 3x3 kernels on bf16 tensors");
    constexpr int NS = X3 ? 2 : 1;                     // hi (and lo) images
    // one block: the patch [hi|lo][k group][position] x 8 bf16, the weight slab [hi|lo][tap][k group][cout] x 8 bf16,
    // the raw tile of the wide staging; after the last chunk the same bytes stage the output tile of the wide epilogue
    // wide staging: the aligned patch rows start XOFF pixels left of the patch (tile origins are multiples of 16
    // pixels and the padding is 1, so that start is a multiple of 8) and are PWA pixels long
    constexpr int XOFF = 7;
    constexpr int PWA = ((XOFF + PW + 15) / 16) * 16;
    constexpr int RP = PWA / 8;                        // 16-byte pieces per row
    constexpr int PH_ = Cfg::PH;
    constexpr int NPIECE = KC * NI * PH_ * RP;
    constexpr int NPL = (NPIECE + kThreads - 1) / kThreads;
    constexpr int CHROW = NI * PH_ * PWA;              // 16-bit elements per channel of the raw tile
    constexpr int kOutPitch = 76;                      // 16-bit elements per pixel row of a wavefront's output tile
    constexpr int LDS16 = NS * KG * PSN + NS * WSLAB + (WIDE ? NPIECE : 0);
    constexpr int OUT16 = (kThreads / kWave) * 32 * kOutPitch * 2 / 16;
    __shared__ u32x4 sAll[LDS16 > OUT16 ? LDS16 : OUT16];
    u32x4* const sP = sAll;
    u32x4* const sW = sAll + NS * KG * PSN;
    u32x4* const sR = sW + NS * WSLAB;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int ctiles = (P.Cout + BN - 1) / BN;
    const int ngrp = blockIdx.z / ctiles;
    const int ctile = blockIdx.z - ngrp * ctiles;
    const int n0 = ngrp * NI;
    const int co_base = ctile * BN;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int iy_base = oy0 * SH - P.ph, ix_base = ox0 * SW - P.pw;
    const int HoWo = P.Ho * P.Wo;

    // the two 32-pixel fragments of this wavefront: tile-linear pixel -> (image, row, column)
    int fimg[NF], fty[NF], ftx[NF], fpos[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int tp = (wv * NF + f) * 32 + l31;
        fimg[f] = tp / (TH * TW);
        const int tpi = tp - fimg[f] * (TH * TW);
        fty[f] = tpi / TW;
        ftx[f] = tpi - fty[f] * TW;
        fpos[f] = half * PSN + fimg[f] * PS + fty[f] * Cfg::OSH * PW + ftx[f] * Cfg::OSW;
    }

    // staging: this thread's patch positions (fixed over the chunks): logical input coordinates, or -1
    int piy[NPOS], pix[NPOS], pim[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        const int e = tid + i * kThreads;
        const int im = e / PS, e1 = e - im * PS;
        const int py = e1 / PW, px = e1 - py * PW;
        const int iy = iy_base + py * Cfg::ISH, ix = ix_base + px * Cfg::ISW;
        const bool ok = e < PSN && iy >= 0 && iy < P.Hi && ix >= 0 && ix < P.Wi && (n0 + im) < P.N;
        piy[i] = ok ? iy : -1;
        pix[i] = ix;
        pim[i] = im;
    }

    // wide staging: this thread's pieces (fixed over the chunks): channel inside the chunk, image, element offset of
    // the piece inside a channel plane (or -1: outside the image)
    int qc[WIDE ? NPL : 1], qim[WIDE ? NPL : 1], qpix[WIDE ? NPL : 1];
    if constexpr (WIDE) {
        const int ax0 = ix_base - XOFF;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int e = tid + i * kThreads;
            const int c = e / (NI * PH_ * RP), r1 = e - c * (NI * PH_ * RP);
            const int im = r1 / (PH_ * RP), r2 = r1 - im * (PH_ * RP);
            const int py = r2 / RP, rp_ = r2 - py * RP;
            const int iy = iy_base + py, ixa = ax0 + rp_ * 8;
            const bool ok = e < NPIECE && iy >= 0 && iy < P.Hi && ixa >= 0 && ixa < P.Wi && (n0 + im) < P.N;
            qc[i] = c; qim[i] = im; qpix[i] = ok ? iy * P.Wi + ixa : -1;
        }
    }

    f32x16 acc[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][h2][i] = 0.0f;

    unsigned rp[WIDE ? 1 : NPOS][WIDE ? 1 : KC / 2];          // packed channel pairs of each position
    unsigned rpl[X3 ? NPOS : 1][X3 ? KC / 2 : 1];   // their low halves (X3)
    u32x4 rq[WIDE ? NPL : 1];                        // wide staging: this thread's pieces
    u32x4 rw[NW];
    u32x4 rwl[X3 ? NW : 1];
    int cbase = 0, s = 0;
    BSrc cur = P.src[0];
    const u32x4* wbase = P.wt + (size_t)ctile * P.nchunks * (NS * WSLAB);

    auto prefetch = [&](int chunk) {
        const int c0 = chunk * KC;
        while (c0 >= cbase + cur.C) { cbase += cur.C; ++s; cur = P.src[s]; }
        const int plane = cur.H * cur.W;
        const int cleft = min(KC, cur.C - (c0 - cbase));           // channels of this chunk that exist
        const size_t img_stride = (size_t)cur.C * plane;
        const size_t chan0 = (size_t)n0 * img_stride + (size_t)(c0 - cbase) * plane;   // uniform
        if constexpr (WIDE) {
            const unsigned short* sp = reinterpret_cast<const unsigned short*>(cur.p) + chan0;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const bool ok = qpix[i] >= 0 && qc[i] < cleft;
                const unsigned lo = ok ? (unsigned)(qim[i] * (int)img_stride + qc[i] * plane + qpix[i]) : 0u;
                const u32x4 v = *reinterpret_cast<const u32x4*>(sp + lo);          // unconditional, as below
                const u32x4 z = {0u, 0u, 0u, 0u};
                rq[i] = ok ? v : z;
            }
        } else if (!X3 && cur.f32 == 2) {
            // blocked source (N, C/8, H, W, 8): a patch position's 8 channels of a group ARE one 16-byte unit of the LDS
            // patch image -- KG loads per position, no packing, no transposition
            const u32x4* sp = reinterpret_cast<const u32x4*>(cur.p) + ((size_t)n0 * (cur.C >> 3) + ((c0 - cbase) >> 3)) * plane;
            const int img_units = (cur.C >> 3) * plane;
#pragma unroll
            for (int i = 0; i < NPOS; ++i) {
                const bool ok = piy[i] >= 0;
                const unsigned lo = ok ? (unsigned)(pim[i] * img_units + (piy[i] >> cur.lh) * cur.W + (pix[i] >> cur.lw)) : 0u;
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    const bool have = 8 * g < cleft;                                 // uniform
                    const u32x4 v = sp[(size_t)(have ? g : 0) * plane + lo];
                    rp[i][4 * g] = (ok && have) ? v[0] : 0u; rp[i][4 * g + 1] = (ok && have) ? v[1] : 0u;
                    rp[i][4 * g + 2] = (ok && have) ? v[2] : 0u; rp[i][4 * g + 3] = (ok && have) ? v[3] : 0u;
                }
            }
        } else if (X3 && cur.f32 == 3) {
            // fp32 blocked source (N, C/8, H, W, 8): a position's 8 channels of a group are 32 contiguous bytes -- two 16-byte
            // loads instead of eight 4-byte ones from eight planes; split into hi / lo halves here as for NCHW fp32
            const u32x4* sp = reinterpret_cast<const u32x4*>(cur.p) + 2 * (((size_t)n0 * (cur.C >> 3) + ((c0 - cbase) >> 3)) * plane);
            const int img_units = (cur.C >> 3) * plane;
#pragma unroll
            for (int i = 0; i < NPOS; ++i) {
                const bool ok = piy[i] >= 0;
                const unsigned lo = ok ? (unsigned)(pim[i] * img_units + (piy[i] >> cur.lh) * cur.W + (pix[i] >> cur.lw)) : 0u;
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    const bool have = 8 * g < cleft;                                 // uniform
                    const float4* q = reinterpret_cast<const float4*>(sp + 2 * ((size_t)(have ? g : 0) * plane + lo));
                    const float4 v0 = q[0], v1 = q[1];
                    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        const unsigned pk = pack2_bf16(f[2 * c2], f[2 * c2 + 1]);
                        const float h0 = __builtin_bit_cast(float, pk << 16), h1 = __builtin_bit_cast(float, pk & 0xffff0000u);
                        const unsigned pl = pack2_bf16(f[2 * c2] - h0, f[2 * c2 + 1] - h1);
                        rp[i][4 * g + c2] = (ok && have) ? pk : 0u;
                        if constexpr (X3) rpl[i][4 * g + c2] = (ok && have) ? pl : 0u;
                    }
                }
            }
        } else
        // every load is unconditional (a predicate per load would put each one in its own basic block and
        // serialise them behind s_waitcnt): channels beyond the source's last one re-read that last channel
        // and are zeroed by a select, padding positions read the chunk's first element.  Addresses are
        // wave-uniform channel base + one 32-bit lane offset per position (no 64-bit vector arithmetic).
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            const bool ok = piy[i] >= 0;
            const unsigned lo = ok ? (unsigned)(pim[i] * (int)img_stride + (piy[i] >> cur.lh) * cur.W + (pix[i] >> cur.lw)) : 0u;
            if (cur.f32) {
                const float* sp = reinterpret_cast<const float*>(cur.p) + chan0;
                float v[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) v[c] = (sp + (size_t)min(c, cleft - 1) * plane)[lo];
#pragma unroll
                for (int c2 = 0; c2 < KC / 2; ++c2) {
                    const unsigned pk = pack2_bf16(v[2 * c2], v[2 * c2 + 1]);
                    const unsigned m = (2 * c2 + 1 < cleft) ? 0xffffffffu : ((2 * c2 < cleft) ? 0x0000ffffu : 0u);
                    rp[i][c2] = ok ? (pk & m) : 0u;
                    if constexpr (X3) {
                        const float h0 = __builtin_bit_cast(float, pk << 16), h1 = __builtin_bit_cast(float, pk & 0xffff0000u);
                        const unsigned pl = pack2_bf16(v[2 * c2] - h0, v[2 * c2 + 1] - h1);
                        rpl[i][c2] = ok ? (pl & m) : 0u;
                    }
                }
            } else {
                const unsigned short* sp = reinterpret_cast<const unsigned short*>(cur.p) + chan0;
                unsigned short v[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) v[c] = (sp + (size_t)min(c, cleft - 1) * plane)[lo];
#pragma unroll
                for (int c2 = 0; c2 < KC / 2; ++c2) {
                    const unsigned lo16 = (2 * c2 < cleft) ? (unsigned)v[2 * c2] : 0u;
                    const unsigned hi16 = (2 * c2 + 1 < cleft) ? (unsigned)v[2 * c2 + 1] : 0u;
                    rp[i][c2] = ok ? (lo16 | (hi16 << 16)) : 0u;
                    if constexpr (X3) rpl[i][c2] = 0u;         // a bf16 tensor has no low half
                }
            }
        }
        const u32x4* wp = wbase + (size_t)chunk * (NS * WSLAB);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = tid + i * kThreads;
            rw[i] = wp[e < WSLAB ? e : 0];
            if constexpr (X3) rwl[i] = wp[WSLAB + (e < WSLAB ? e : 0)];
        }
    };
    auto commit = [&]() {
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int e = tid + i * kThreads;
                if ((i + 1) * kThreads <= NPIECE || e < NPIECE) sR[e] = rq[i];      // only a partial last round is predicated
            }
        } else
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            const int e = tid + i * kThreads;
            if ((i + 1) * kThreads <= PSN || e < PSN) {
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    u32x4 v;
                    v[0] = rp[i][4 * g]; v[1] = rp[i][4 * g + 1]; v[2] = rp[i][4 * g + 2]; v[3] = rp[i][4 * g + 3];
                    sP[g * PSN + e] = v;
                    if constexpr (X3) {
                        v[0] = rpl[i][4 * g]; v[1] = rpl[i][4 * g + 1]; v[2] = rpl[i][4 * g + 2]; v[3] = rpl[i][4 * g + 3];
                        sP[KG * PSN + g * PSN + e] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = tid + i * kThreads;
            if ((i + 1) * kThreads <= WSLAB || e < WSLAB) {
                sW[e] = rw[i];
                if constexpr (X3) sW[WSLAB + e] = rwl[i];
            }
        }
    };

    prefetch(0);
    for (int chunk = 0; chunk < P.nchunks; ++chunk) {
        commit();
        __syncthreads();
        if (chunk + 1 < P.nchunks) prefetch(chunk + 1);       // in flight during the MFMA phase
        if constexpr (WIDE) {
            // raw tile -> channel-innermost patch: a 16-lane group takes (k group, image, row, 16 aligned pixels)
            constexpr int NBLK = PWA / 16, T16 = KG * NI * PH_ * NBLK;
            const unsigned short* R = reinterpret_cast<const unsigned short*>(sR);
            const int a = tid & 15;
            for (int t = tid >> 4; t < T16; t += kThreads / 16) {
                const int g = t / (NI * PH_ * NBLK), r1 = t - g * (NI * PH_ * NBLK);
                const int im = r1 / (PH_ * NBLK), r2 = r1 - im * (PH_ * NBLK);
                const int py = r2 / NBLK, blk = r2 - py * NBLK;
                const unsigned short* p0 = R + (8 * g + (a >> 2)) * CHROW + (im * PH_ + py) * PWA + blk * 16 + 4 * (a & 3);
                const tpspp_u32x2 k0 = __builtin_bit_cast(tpspp_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) tpspp_s16x4*)p0));
                const tpspp_u32x2 k1 = __builtin_bit_cast(tpspp_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) tpspp_s16x4*)(p0 + 4 * CHROW)));
                const int px = blk * 16 + a - XOFF;
                if (px >= 0 && px < PW) {
                    u32x4 v;
                    v[0] = k0[0]; v[1] = k0[1]; v[2] = k1[0]; v[3] = k1[1];
                    sP[g * PSN + im * PS + py * PW + px] = v;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, sW[(tap * KG + 2 * ks + half) * BN + l31]);
                bf16x8 a1 = a0, a0l = a0, a1l = a0;
                if constexpr (NB == 2) a1 = __builtin_bit_cast(bf16x8, sW[(tap * KG + 2 * ks + half) * BN + 32 + l31]);
                if constexpr (X3) {
                    a0l = __builtin_bit_cast(bf16x8, sW[WSLAB + (tap * KG + 2 * ks + half) * BN + l31]);
                    if constexpr (NB == 2) a1l = __builtin_bit_cast(bf16x8, sW[WSLAB + (tap * KG + 2 * ks + half) * BN + 32 + l31]);
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, sP[fpos[f] + (2 * ks) * PSN + ky * PW + kx]);
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[f][0], 0, 0, 0);
                    if constexpr (NB == 2) acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[f][1], 0, 0, 0);
                    if constexpr (X3) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, sP[KG * PSN + fpos[f] + (2 * ks) * PSN + ky * PW + kx]);
                        acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[f][0], 0, 0, 0);
                        if constexpr (NB == 2) acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[f][1], 0, 0, 0);
                        acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, acc[f][0], 0, 0, 0);
                        if constexpr (NB == 2) acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, acc[f][1], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, residual, ReLU, affine; half-wavefronts store rows of 32 consecutive pixels ----
    // For a 1x1 layer with 32 input channels the epilogue IS the kernel (8 MFMAs against 64 outputs per lane),
    // so it is kept off the vector ALU: addresses are a wave-uniform base (image group, channel tile, channel:
    // scalar arithmetic) plus ONE 32-bit lane offset per fragment, the bias arrives as eight float4 per lane,
    // bf16 pairs are rounded by v_cvt_pk_bf16_f32 and stored from the two halves of one register.
    const bool full_c = co_base + BN <= P.Cout;                  // uniform
    float bq[2][4][4];
#pragma unroll
    for (int h2 = 0; h2 < NB; ++h2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = co_base + 32 * h2 + 8 * g + 4 * half;
            if (P.bias && full_c) {
                const float4 b4 = *reinterpret_cast<const float4*>(P.bias + co);
                bq[h2][g][0] = b4.x; bq[h2][g][1] = b4.y; bq[h2][g][2] = b4.z; bq[h2][g][3] = b4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[h2][g][e] = (P.bias && co + e < P.Cout) ? P.bias[co + e] : 0.0f;
            }
        }
    const size_t ubase = ((size_t)n0 * P.Cout + co_base) * HoWo;  // uniform: first image / channel of the tile
    // (GELU goes through the general path: with erff inlined into the common one the compiler no longer keeps that
    //  path branch-free and every convolution pays for it -- measured: 1x1 32->64 0.11 -> 0.21 ms)
    const bool simple = P.res_mode == 0 && P.post_scale == nullptr && P.relu != 2;
    // wide epilogue (bf16 output, whole tile inside the tensor, rows of whole 16-byte pieces): results go through the
    // wavefront's [pixel][64 channels] LDS tile (8 bytes per lane and channel quad) and leave as 16-byte pieces of the
    // NCHW rows, brought into pixel order by transposing reads -- 4 stores per lane and fragment instead of 64 (which
    // were 36 % of a 3x3 64->64 layer's time)
    const bool blk_out = P.out_f32 == 2, blk_res = P.res_f32 == 2;                                        // uniform
    const bool blk32_out = P.out_f32 == 3, blk32_res = P.res_f32 == 3;                                    // fp32 blocked
    const bool wide_out = !P.out_f32 && full_c && (P.Wo & 7) == 0 && (TW & 7) == 0 && oy0 + TH <= P.Ho && ox0 + TW <= P.Wo &&
                          n0 + NI <= P.N && (reinterpret_cast<size_t>(P.out) & 15) == 0;       // uniform
    unsigned short* const otile = reinterpret_cast<unsigned short*>(sAll) + wv * (32 * kOutPitch);
    // The two forms of the value computation are two separate loops: with the test inside the (fully unrolled) loop the
    // 128 copies of the general form -- residual loads, inlined erff -- sit between the few instructions the plain
    // bias + ReLU layers execute, and a workgroup spent 11 k cycles (a third of its life) walking that code.
    const bool relu1 = P.relu == 1;
    auto epilogue = [&](auto simple_c) {
        constexpr bool SIMPLE = decltype(simple_c)::value;
    #pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int oy = oy0 + fty[f], ox = ox0 + ftx[f], n = n0 + fimg[f];
            const bool valid = oy < P.Ho && ox < P.Wo && n < P.N;
            const unsigned lo = valid ? (unsigned)((fimg[f] * P.Cout + 4 * half) * HoWo + oy * P.Wo + ox) : 0u;
    #pragma unroll
            for (int h2 = 0; h2 < NB; ++h2) {
                tpspp_u32x2 bpk[4];
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    const int cu = 32 * h2 + 8 * g;                   // + 4*half (in `lo`) + e
                    // blocked tensors: this lane's four channels are 8 bytes of the unit (image, channel group, pixel)
                    const size_t bunit = (((size_t)n * (P.Cout >> 3) + ((co_base + cu) >> 3)) * HoWo + (size_t)(valid ? oy * P.Wo + ox : 0)) * 8 + 4 * half;
                    tpspp_u32x2 rb; rb[0] = rb[1] = 0u;
                    float4 rb32 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if constexpr (!SIMPLE) {
                        if (blk_res && P.res_mode && valid && co_base + cu + 4 * half < P.Cout)
                            rb = *reinterpret_cast<const tpspp_u32x2*>(reinterpret_cast<const unsigned short*>(P.res) + bunit);
                        if (blk32_res && P.res_mode && valid && co_base + cu + 4 * half < P.Cout)
                            rb32 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(P.res) + bunit);
                    }
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[f][h2][4 * g + e] + bq[h2][g][e];
                        if constexpr (!SIMPLE) {
                            const size_t o = ubase + (size_t)(cu + e) * HoWo + lo;
                            const int co = co_base + cu + 4 * half + e;
                            float rv = 0.0f;
                            if (blk_res)
                                rv = bf16_bits_to_f32((unsigned short)((rb[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                            else if (blk32_res)
                                rv = e == 0 ? rb32.x : (e == 1 ? rb32.y : (e == 2 ? rb32.z : rb32.w));
                            else if (P.res_mode && valid && co < P.Cout)
                                rv = P.res_f32 ? reinterpret_cast<const float*>(P.res)[o]
                                               : bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(P.res)[o]);
                            if (P.res_mode == 2) v[e] = v[e] + rv;
                            if (P.relu == 1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                            else if (P.relu == 2) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                            if (P.res_mode == 1) v[e] = v[e] + rv;
                            if (P.post_scale && co < P.Cout) v[e] = v[e] * P.post_scale[co] + P.post_shift[co];
                        } else if (relu1) {
                            v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                        }
                    }
                    const int co4 = co_base + cu + 4 * half;          // this lane's first channel of the quad
                    if (blk_out) {
                        // the two half-wavefronts hold the two halves of a 16-byte unit: v_permlane32_swap pairs them up
                        // (lower half-wavefront: the unit of group g, upper: that of group g + 1), one 16-byte store
                        // per two groups
                        bpk[g][0] = pack2_bf16(v[0], v[1]); bpk[g][1] = pack2_bf16(v[2], v[3]);
                        if (g & 1) {
                            const tpspp_u32x2 d0 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][0], bpk[g][0], false, false);
                            const tpspp_u32x2 d1 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][1], bpk[g][1], false, false);
                            u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                            const int kg = ((co_base + 32 * h2) >> 3) + (g - 1) + half;
                            const size_t bu = (((size_t)n * (P.Cout >> 3) + kg) * HoWo + (size_t)(valid ? oy * P.Wo + ox : 0)) * 8;
                            if (valid && 8 * kg < P.Cout)
                                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(P.out) + bu) = unit;
                        }
                    } else if (blk32_out) {
                        // fp32 blocked: this lane's four channels are 16 bytes of the unit (image, channel group, pixel)
                        if (valid && co4 < P.Cout)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.out) + bunit) = make_float4(v[0], v[1], v[2], v[3]);
                    } else if (P.out_f32) {
                        float* ob = reinterpret_cast<float*>(P.out) + ubase + (size_t)cu * HoWo;      // uniform
    #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (valid && (full_c || co4 + e < P.Cout)) (ob + (size_t)e * HoWo)[lo] = v[e];
                    } else {
                        unsigned short* ob = reinterpret_cast<unsigned short*>(P.out) + ubase + (size_t)cu * HoWo;
                        const unsigned p01 = pack2_bf16(v[0], v[1]), p23 = pack2_bf16(v[2], v[3]);
                        if (wide_out) {
                            tpspp_u32x2 pk; pk[0] = p01; pk[1] = p23;
                            *reinterpret_cast<tpspp_u32x2*>(otile + l31 * kOutPitch + cu + 4 * half) = pk;
                            continue;
                        }
                        if (valid && (full_c || co4 < P.Cout)) ob[lo] = (unsigned short)(p01 & 0xffffu);
                        if (valid && (full_c || co4 + 1 < P.Cout)) (ob + (size_t)HoWo)[lo] = (unsigned short)(p01 >> 16);
                        if (valid && (full_c || co4 + 2 < P.Cout)) (ob + (size_t)2 * HoWo)[lo] = (unsigned short)(p23 & 0xffffu);
                        if (valid && (full_c || co4 + 3 < P.Cout)) (ob + (size_t)3 * HoWo)[lo] = (unsigned short)(p23 >> 16);
                    }
                }
            }
            if (wide_out) {
                asm volatile("" ::: "memory");
                const int a = lane & 15, q = lane >> 4;
                const int tp = (wv * NF + f) * 32 + 8 * q;               // first pixel of this lane's pieces (tile-linear)
                const int pim_ = tp / (TH * TW), tpi = tp - pim_ * (TH * TW);
                const int pty = tpi / TW, ptx = tpi - pty * TW;
                unsigned short* orow = reinterpret_cast<unsigned short*>(P.out) + ubase +
                                       ((size_t)pim_ * P.Cout) * HoWo + (size_t)(oy0 + pty) * P.Wo + (ox0 + ptx);
                u32x4 pv[4];
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned short* pp = otile + (8 * q + (a >> 2)) * kOutPitch + 16 * i + 4 * (a & 3);
                    const tpspp_u32x2 lo2 = __builtin_bit_cast(tpspp_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) tpspp_s16x4*)pp));
                    const tpspp_u32x2 hi2 = __builtin_bit_cast(tpspp_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) tpspp_s16x4*)(pp + 4 * kOutPitch)));
                    pv[i][0] = lo2[0]; pv[i][1] = lo2[1]; pv[i][2] = hi2[0]; pv[i][3] = hi2[1];
                }
    #pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(orow + (size_t)(16 * i + a) * HoWo) = pv[i];
                asm volatile("" ::: "memory");
            }
        }
    };
    if (simple) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// wide staging applies: every source bf16 at full resolution, rows made of whole, aligned 16-byte pieces
inline bool wide_staging_applies(const BParams& P)
{
    for (int i = 0; i < P.nsrc; ++i) {
        const BSrc& s = P.src[i];
        if (s.f32 || s.lh || s.lw || (s.W & 7) || (reinterpret_cast<size_t>(s.p) & 15)) return false;
    }
    return true;
}

template <int KH, int SH, int SW, int TH, int TW, int NI, int KC, bool X3, int NF = 2>
void launch_b(const BParams& P, hipStream_t st)
{
    const int ctiles = (P.Cout + BN - 1) / BN;
    const dim3 grid((unsigned)((P.Wo + TW - 1) / TW), (unsigned)((P.Ho + TH - 1) / TH),
                    (unsigned)(((P.N + NI - 1) / NI) * ctiles));
    // (stride 2: the dense patch is 4x the output, the raw tile and its transposition cost more than the gather saves:
    //  127 -> 154 us on the 32x128 -> 16x64 layers)
    if constexpr (KH == 3 && SH == 1 && SW == 1 && !X3 && (TW % 16) == 0) {
        if (wide_staging_applies(P)) {
            hipLaunchKernelGGL((conv_tiled_bf16_kernel<KH, SH, SW, TH, TW, NI, KC, X3, NF, true>), grid, dim3(kThreads), 0, st, P);
            return;
        }
    }
    if constexpr (X3) {
        if (P.Cout <= 32) {                                   // one accumulator per fragment (see the kernel, NB)
            hipLaunchKernelGGL((conv_tiled_bf16_kernel<KH, SH, SW, TH, TW, NI, KC, X3, NF, false, 1>), grid, dim3(kThreads), 0, st, P);
            return;
        }
    }
    hipLaunchKernelGGL((conv_tiled_bf16_kernel<KH, SH, SW, TH, TW, NI, KC, X3, NF>), grid, dim3(kThreads), 0, st, P);
}

constexpr int kKC3 = 16;      // channels per chunk, 3x3 kernels
constexpr int kKC1 = 32;      // channels per chunk, 1x1 kernels

// picks the tile by output width / height; false when no instantiation fits
template <int KH, int SH, int SW, int KC, bool X3>
bool launch_by_shape(const BParams& P, hipStream_t st)
{
    if constexpr (KH == 3 && SH == 2) {
        // stride 2: the patch holds 4x the output positions -- 128-pixel tiles (half the LDS and registers, twice the
        // workgroups); the same for the maps of at most 8 rows of 16, where 256-pixel tiles start too few workgroups
        if (P.Wo > 64)      launch_b<KH, SH, SW, 1, 128, 1, KC, X3, 1>(P, st);
        else if (P.Wo > 32) launch_b<KH, SH, SW, 2, 64, 1, KC, X3, 1>(P, st);
        else if (P.Wo > 16) launch_b<KH, SH, SW, 4, 32, 1, KC, X3, 1>(P, st);
        else if (P.Ho > 4)  launch_b<KH, SH, SW, 8, 16, 1, KC, X3, 1>(P, st);
        else if (P.Ho > 2)  launch_b<KH, SH, SW, 4, 16, 2, KC, X3, 1>(P, st);
        else                launch_b<KH, SH, SW, 2, 16, 4, KC, X3, 1>(P, st);
        return true;
    } else if constexpr (KH == 3 && X3) {
        // three-term split, stride 1: 128-pixel tiles as well (two 256-pixel workgroups would need 2 x 62 KB of LDS)
        if (P.Wo > 64)      launch_b<KH, SH, SW, 1, 128, 1, KC, X3, 1>(P, st);
        else if (P.Wo > 32) launch_b<KH, SH, SW, 2, 64, 1, KC, X3, 1>(P, st);
        else if (P.Wo > 16) launch_b<KH, SH, SW, 4, 32, 1, KC, X3, 1>(P, st);
        else if (P.Ho > 4)  launch_b<KH, SH, SW, 8, 16, 1, KC, X3, 1>(P, st);
        else if (P.Ho > 2)  launch_b<KH, SH, SW, 4, 16, 2, KC, X3, 1>(P, st);
        else                launch_b<KH, SH, SW, 2, 16, 4, KC, X3, 1>(P, st);
        return true;
    } else {
        if (P.Ho == 1 && P.Wo > 128) launch_b<KH, SH, SW, 1, 256, 1, KC, X3>(P, st);   // a row of tokens
        else if (P.Wo > 64) launch_b<KH, SH, SW, 2, 128, 1, KC, X3>(P, st);
        else if (P.Wo > 32) launch_b<KH, SH, SW, 4, 64, 1, KC, X3>(P, st);
        else if (P.Wo > 16) {
            // 3x3 on 32-wide maps (backbone layer 3-4): 128-pixel tiles measured 3 % faster (64-wide maps: neutral)
            if constexpr (KH == 3) launch_b<KH, SH, SW, 4, 32, 1, KC, X3, 1>(P, st);
            else launch_b<KH, SH, SW, 8, 32, 1, KC, X3>(P, st);
        }
        else if (P.Ho > 8)  launch_b<KH, SH, SW, 16, 16, 1, KC, X3>(P, st);
        else if (P.Ho > 4)  launch_b<KH, SH, SW, 8, 16, 2, KC, X3>(P, st);
        else if (P.Ho > 2)  launch_b<KH, SH, SW, 4, 16, 2, KC, X3, 1>(P, st);          // 128-pixel tiles: 2 images
        else                launch_b<KH, SH, SW, 2, 16, 4, KC, X3, 1>(P, st);          //                  4 images
        return true;
    }
}

}  // namespace

namespace tpspp {
// defined in tpspp_conv_bf16x3.hip: dispatches the split3 instantiations; false when no kernel fits
bool conv_bf16x3_launch(const BParams& P, int KH, int sh, int sw, hipStream_t st);
// defined in tpspp_conv_bf16_persist.hip: the persistent kernel for blocked 3x3 layers; false when it does not apply
bool conv_bf16_persist_launch(const BParams& P, int sh, int sw, hipStream_t st);
// defined in tpspp_conv1x1_blk.hip: 1x1 layers between blocked maps; false when it does not apply
bool conv1x1_blk_launch(const BParams& P, hipStream_t st);
// defined in tpspp_conv3_wide.hip: 3x3 stride-1 layers with >= 128 output channels between blocked maps (128 x 64 wavefront
// tiles, weights streamed from L2 into registers); false when it does not apply
bool conv3_wide_launch(const BParams& P, hipStream_t st);
bool conv1x1_wide_launch(const BParams& P, hipStream_t st);   // the same kernel for the 1x1 layers with >= 256 input channels
// defined in tpspp_conv_stem.hip: the backbone's stem (3x3, <= 3 fp32 input channels -> 32 bf16 NCHW); false when it does not apply
bool conv_stem_launch(const BParams& P, hipStream_t st);
extern int g_conv_bf16_no_persist;           // tpspp_conv_set_tuning bit 1
extern int g_conv_bf16_no_wide;              // tpspp_conv_set_tuning bit 2
}  // namespace tpspp
