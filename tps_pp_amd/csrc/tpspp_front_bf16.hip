// The pointwise front of the TPS++ regressor (ResNet45v2 wiring) in ONE kernel on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation), register-chained: the bf16 twin of tpspp_front.hip.
//
//     feat0 = relu(W0 outs0[p] + b0)                       1x1, 32 -> 64, full resolution (32x128)
//     feat1 = relu(W1 outs1[p] + b1)                       1x1, 32 -> 64
//     feat2 = relu(W2 x[p/2] + b2)                         1x1, 64 -> 64, half resolution (16x64)
//     feat_grid = relu(Wg [feat0; feat1; feat2] + bg)      1x1, 192 -> 64  (cat + nearest Upsample)
//
// As four separate launches these layers are HBM-bound at 2-3 TB/s (a single K-chunk each, feat0 / feat1 /
// feat2 written and read back: 4.1 MB per image); fused, an image moves 0.66 MB in and 1.7 MB out.
//
// A wavefront owns 32 consecutive full-resolution pixels of a row; a lane is (pixel l31, k-half h).  The B
// operand of a 32x32x16 step is 8 consecutive k per lane: for the three small GEMMs those are 8 input channels
// of the lane's pixel (coalesced 2-B row loads, packed in pairs).  Their results come back in the C/D layout --
// lane (pixel, h) holds output channels 8g + 4h + {0..3} -- and two consecutive groups g are exactly the 8 values
// the lane must supply as B operand of one k-step of the 192-deep GEMM, provided its weight slab is permuted on
// the host to that k-slot order (slot 8h+e of a 16-channel group = channel [0,1,2,3,8,9,10,11,4,5,6,7,12,..,15][8h+e]).
// So feat0 / feat1 / feat2 are rounded to bf16 once (the same rounding their stores to HBM use: the unfused
// composition computes feat_grid from exactly these values), packed with v_cvt_pk_bf16_f32 and fed straight
// back; no LDS, no barrier.  Weight fragments are 16-B reads of host-arranged [k-step][h][64 cout][8 k] slabs
// (44 KB in all: L1 / L2 resident).
//
// Reference: TPS_PP.forward / TPS_PP.grid, mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:560-562,580-585.
// Bound: HBM (2.4 MB per image at 17 kMAC per pixel).
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FrontBParams {
    const unsigned short* o0; const unsigned short* o1;   // (N, 32, H, W) bf16
    const unsigned short* x;                              // (N, 64, H/2, W/2) bf16
    const u32x4* w0; const u32x4* w1;                     // [2 k-steps][2][64][8] bf16
    const u32x4* w2;                                      // [4][2][64][8]
    const u32x4* wg;                                      // [12][2][64][8], k-slots in chain order
    const float* b0; const float* b1; const float* b2; const float* bg;
    unsigned short* feat0; unsigned short* feat1;         // (N, 64, H, W) bf16
    unsigned short* feat2;                                // (N, 64, H/2, W/2) bf16
    void* feat_grid;                                      // (N, 64, H, W) bf16 or fp32
    int fg_f32;
    int N, H, W;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Loads the B operand of NK k-steps: channels 16j + 8h + e of one pixel.  `base` is wave-uniform (first channel
// plane of the image, at the wavefront's row segment), `lo` the lane's 32-bit element offset (pixel + 8h planes):
// every load is  scalar base + one shared VGPR offset, and a channel pair lands in the two halves of one
// register (d16 / d16_hi loads): no 64-bit vector addresses, no packing instructions.
template <int NK>
__device__ __forceinline__ void load_b(const unsigned short* __restrict__ base, unsigned lo, int plane, u32x4 (&b)[NK])
{
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned short* p0 = base + (size_t)(16 * j + 2 * q) * plane;      // uniform
            u16x2 v;
            v[0] = p0[lo];
            v[1] = p0[(size_t)plane + lo];
            b[j][q] = __builtin_bit_cast(unsigned, v);
        }
}

// out[t] = slab^T in  over NK k-steps, both 32-channel tiles
template <int NK>
__device__ __forceinline__ void gemm(const u32x4* __restrict__ slab, const u32x4 (&in)[NK], int half, int l31,
                                     f32x16 (&acc)[2])
{
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bb = __builtin_bit_cast(bf16x8, in[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1], 0, 0, 0);
    }
}

// bias + ReLU + one rounding to bf16; returns the 64 features as chain-ordered B fragments (4 k-steps) and
// stores them (NCHW, 64-B row segments per half-wavefront) when `st` is set.  `dst` is wave-uniform, `so` the
// lane's element offset (pixel + 4h planes).
__device__ __forceinline__ void finish(const f32x16 (&acc)[2], const float* __restrict__ bias, int half,
                                       unsigned short* __restrict__ dst, unsigned so, int plane, bool st,
                                       u32x4* __restrict__ out)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = acc[t][4 * g + e] + bias[32 * t + 8 * g + 4 * half + e];
                v[e] = s > 0.0f ? s : 0.0f;
            }
            const unsigned p01 = pack_bf16(v[0], v[1]), p23 = pack_bf16(v[2], v[3]);
            // group g of tile t is k-step 2t + g/2, slots 4(g&1) .. 4(g&1)+3
            out[2 * t + (g >> 1)][2 * (g & 1)] = p01;
            out[2 * t + (g >> 1)][2 * (g & 1) + 1] = p23;
            if (st) {
                unsigned short* d = dst + (size_t)(32 * t + 8 * g) * plane;          // uniform
                d[so] = (unsigned short)(p01 & 0xffffu);
                d[(size_t)plane + so] = (unsigned short)(p01 >> 16);
                d[(size_t)2 * plane + so] = (unsigned short)(p23 & 0xffffu);
                d[(size_t)3 * plane + so] = (unsigned short)(p23 >> 16);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
front_bf16_kernel(const FrontBParams P)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int segs_per_row = P.W >> 5;
    const long seg = (long)blockIdx.x * 4 + wv;                       // 32-pixel row segment (wave-uniform)
    const long nseg = (long)P.N * P.H * segs_per_row;
    if (seg >= nseg) return;
    const int sx = (int)(seg % segs_per_row);
    const long row = seg / segs_per_row;
    const int y = (int)(row % P.H);
    const int n = (int)(row / P.H);
    const int plane = P.H * P.W, W2 = P.W >> 1, plane2 = (P.H >> 1) * W2;
    const size_t seg0 = (size_t)y * P.W + sx * 32;                    // first pixel of the segment (uniform)
    const size_t seg2 = (size_t)(y >> 1) * W2 + sx * 16;
    const int xx = sx * 32 + l31;

    // the half-resolution input first (its 32 loads stay in flight across the two small GEMMs)
    u32x4 in2[4];
    load_b<4>(P.x + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 8 * half * plane2), plane2, in2);
    u32x4 in0[2], in1[2];
    const unsigned lo = (unsigned)(l31 + 8 * half * plane);
    load_b<2>(P.o0 + (size_t)n * 32 * plane + seg0, lo, plane, in0);
    load_b<2>(P.o1 + (size_t)n * 32 * plane + seg0, lo, plane, in1);

    u32x4 f[12];                                   // [feat0 | feat1 | feat2] as B fragments of the 192-deep GEMM
    f32x16 acc[2];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    };
    const size_t obase = (size_t)n * 64 * plane + seg0;               // uniform
    const unsigned so = (unsigned)(l31 + 4 * half * plane);
    zero();
    gemm<2>(P.w0, in0, half, l31, acc);
    finish(acc, P.b0, half, P.feat0 + obase, so, plane, true, f);
    zero();
    gemm<2>(P.w1, in1, half, l31, acc);
    finish(acc, P.b1, half, P.feat1 + obase, so, plane, true, f + 4);
    zero();
    gemm<4>(P.w2, in2, half, l31, acc);
    // one lane of every 2x2 block writes the half-resolution pixel
    finish(acc, P.b2, half, P.feat2 + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 4 * half * plane2),
           plane2, ((y | xx) & 1) == 0, f + 8);
    zero();
    gemm<12>(P.wg, f, half, l31, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = 32 * t + (r & 3) + 8 * (r >> 2);           // + 4*half: in the lane offset
            float v = acc[t][r] + P.bg[cu + 4 * half];
            v = v > 0.0f ? v : 0.0f;
            if (P.fg_f32) (reinterpret_cast<float*>(P.feat_grid) + obase + (size_t)cu * plane)[so] = v;
            else (reinterpret_cast<unsigned short*>(P.feat_grid) + obase + (size_t)cu * plane)[so] =
                (unsigned short)(pack_bf16(v, 0.0f) & 0xffffu);
        }
    }
}

// ---- the same kernel for fp32 tensors with the three-term bf16 split ("bf16x3", tpspp_conv_bf16.hip) ---------------
// Inputs, feat0 / feat1 / feat2 and feat_grid are fp32 in memory; every operand is split hi = bf16(v), lo = bf16(v - hi)
// in registers and a product is hi*hi + hi*lo + lo*hi.  feat0 / feat1 / feat2 are chained WITHOUT an intermediate
// rounding (their fp32 values are split, exactly what the separate convolutions would do with the stored tensors).
// Slabs: [hi|lo][k-steps][2][64][8].
struct FrontXParams {
    const float* o0; const float* o1; const float* x;
    const u32x4* w0; const u32x4* w1; const u32x4* w2; const u32x4* wg;
    const float* b0; const float* b1; const float* b2; const float* bg;
    float* feat0; float* feat1; float* feat2; float* feat_grid;
    int N, H, W;
};

__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(v0, v1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(v0 - h0, v1 - h1);
}

template <int NK>
__device__ __forceinline__ void load_b3(const float* __restrict__ base, unsigned lo, int plane, u32x4 (&bh)[NK], u32x4 (&bl)[NK])
{
    float v[NK][8];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = (base + (size_t)(16 * j + e) * plane)[lo];      // uniform base + lane offset
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2(v[j][2 * q], v[j][2 * q + 1], h, l);
            bh[j][q] = h; bl[j][q] = l;
        }
}

template <int NK>
__device__ __forceinline__ void gemm3(const u32x4* __restrict__ slab, const u32x4* __restrict__ inh, const u32x4* __restrict__ inl,
                                      int half, int l31, f32x16 (&acc)[2])
{
    constexpr int LO = NK * 128;                       // 16-B units between the hi and the lo slab
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 a0l = __builtin_bit_cast(bf16x8, slab[LO + (2 * j + half) * 64 + l31]);
        const bf16x8 a1l = __builtin_bit_cast(bf16x8, slab[LO + (2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, inh[j]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, inl[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, acc[1], 0, 0, 0);
    }
}

// bias + ReLU in fp32, fp32 store, and the result as chain-ordered hi / lo operands (4 k-steps)
__device__ __forceinline__ void finish3(const f32x16 (&acc)[2], const float* __restrict__ bias, int half,
                                        float* __restrict__ dst, unsigned so, int plane, bool st,
                                        u32x4* __restrict__ outh, u32x4* __restrict__ outl)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = acc[t][4 * g + e] + bias[32 * t + 8 * g + 4 * half + e];
                v[e] = s > 0.0f ? s : 0.0f;
            }
            unsigned h01, l01, h23, l23;
            split2(v[0], v[1], h01, l01);
            split2(v[2], v[3], h23, l23);
            outh[2 * t + (g >> 1)][2 * (g & 1)] = h01; outl[2 * t + (g >> 1)][2 * (g & 1)] = l01;
            outh[2 * t + (g >> 1)][2 * (g & 1) + 1] = h23; outl[2 * t + (g >> 1)][2 * (g & 1) + 1] = l23;
            if (st) {
                float* d = dst + (size_t)(32 * t + 8 * g) * plane;                   // uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) (d + (size_t)e * plane)[so] = v[e];
            }
        }
    }
}

__global__ void __launch_bounds__(256)
front_x3_kernel(const FrontXParams P)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int segs_per_row = P.W >> 5;
    const long seg = (long)blockIdx.x * 4 + wv;
    const long nseg = (long)P.N * P.H * segs_per_row;
    if (seg >= nseg) return;
    const int sx = (int)(seg % segs_per_row);
    const long row = seg / segs_per_row;
    const int y = (int)(row % P.H);
    const int n = (int)(row / P.H);
    const int plane = P.H * P.W, W2 = P.W >> 1, plane2 = (P.H >> 1) * W2;
    const size_t seg0 = (size_t)y * P.W + sx * 32;
    const size_t seg2 = (size_t)(y >> 1) * W2 + sx * 16;
    const int xx = sx * 32 + l31;
    const unsigned lo = (unsigned)(l31 + 8 * half * plane);
    const size_t obase = (size_t)n * 64 * plane + seg0;
    const unsigned so = (unsigned)(l31 + 4 * half * plane);

    u32x4 fh[12], fl[12];
    f32x16 acc[2];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    };
    {
        u32x4 ih[2], il[2];
        load_b3<2>(P.o0 + (size_t)n * 32 * plane + seg0, lo, plane, ih, il);
        zero();
        gemm3<2>(P.w0, ih, il, half, l31, acc);
        finish3(acc, P.b0, half, P.feat0 + obase, so, plane, true, fh, fl);
    }
    {
        u32x4 ih[2], il[2];
        load_b3<2>(P.o1 + (size_t)n * 32 * plane + seg0, lo, plane, ih, il);
        zero();
        gemm3<2>(P.w1, ih, il, half, l31, acc);
        finish3(acc, P.b1, half, P.feat1 + obase, so, plane, true, fh + 4, fl + 4);
    }
    {
        u32x4 ih[4], il[4];
        load_b3<4>(P.x + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 8 * half * plane2), plane2, ih, il);
        zero();
        gemm3<4>(P.w2, ih, il, half, l31, acc);
        finish3(acc, P.b2, half, P.feat2 + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 4 * half * plane2),
                plane2, ((y | xx) & 1) == 0, fh + 8, fl + 8);
    }
    zero();
    gemm3<12>(P.wg, fh, fl, half, l31, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cu = 32 * t + (r & 3) + 8 * (r >> 2);
            float v = acc[t][r] + P.bg[cu + 4 * half];
            v = v > 0.0f ? v : 0.0f;
            (P.feat_grid + obase + (size_t)cu * plane)[so] = v;
        }
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_front_bf16_fwd(const void* outs0, const void* outs1, const void* x,
                                      const void* w0, const float* b0, const void* w1, const float* b1,
                                      const void* w2, const float* b2, const void* wg, const float* bg,
                                      void* feat0, void* feat1, void* feat2, void* feat_grid, int feat_grid_f32,
                                      int N, int H, int W, int split3, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(outs0 && outs1 && x && w0 && w1 && w2 && wg && b0 && b1 && b2 && bg && feat0 && feat1 && feat2 &&
                  feat_grid, "tpspp_front_bf16_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 32) == 0,
                  "tpspp_front_bf16_fwd: needs an even height and a width that is a multiple of 32");
    if (N == 0) return TPSPP_OK;
    const long nseg = (long)N * H * (W / 32);
    const long blocks = (nseg + 3) / 4;
    TPSPP_REQUIRE(blocks <= 0x7fffffffL, "tpspp_front_bf16_fwd: grid too large");
    if (split3) {
        FrontXParams X;
        X.o0 = static_cast<const float*>(outs0); X.o1 = static_cast<const float*>(outs1); X.x = static_cast<const float*>(x);
        X.w0 = static_cast<const u32x4*>(w0); X.w1 = static_cast<const u32x4*>(w1);
        X.w2 = static_cast<const u32x4*>(w2); X.wg = static_cast<const u32x4*>(wg);
        X.b0 = b0; X.b1 = b1; X.b2 = b2; X.bg = bg;
        X.feat0 = static_cast<float*>(feat0); X.feat1 = static_cast<float*>(feat1); X.feat2 = static_cast<float*>(feat2);
        X.feat_grid = static_cast<float*>(feat_grid);
        X.N = N; X.H = H; X.W = W;
        hipLaunchKernelGGL(front_x3_kernel, dim3((unsigned)blocks), dim3(256), 0, tpspp::as_stream(stream), X);
        return tpspp::check_launch("tpspp_front_bf16_fwd(x3)");
    }
    FrontBParams P;
    P.o0 = static_cast<const unsigned short*>(outs0); P.o1 = static_cast<const unsigned short*>(outs1);
    P.x = static_cast<const unsigned short*>(x);
    P.w0 = static_cast<const u32x4*>(w0); P.w1 = static_cast<const u32x4*>(w1);
    P.w2 = static_cast<const u32x4*>(w2); P.wg = static_cast<const u32x4*>(wg);
    P.b0 = b0; P.b1 = b1; P.b2 = b2; P.bg = bg;
    P.feat0 = static_cast<unsigned short*>(feat0); P.feat1 = static_cast<unsigned short*>(feat1);
    P.feat2 = static_cast<unsigned short*>(feat2); P.feat_grid = feat_grid; P.fg_f32 = feat_grid_f32 ? 1 : 0;
    P.N = N; P.H = H; P.W = W;
    hipLaunchKernelGGL(front_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_front_bf16_fwd");
}
