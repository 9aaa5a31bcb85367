// The pointwise front of the TPS++ regressor (ResNet45v2 wiring) in ONE kernel, fp32 MFMA,
// register-chained (no activation goes through LDS; see tpspp_dgab.hip for the idiom):
//
//     feat0 = relu(W0 outs0[p] + b0)                       1x1, 32 -> 64, full resolution (32x128)
//     feat1 = relu(W1 outs1[p] + b1)                       1x1, 32 -> 64
//     feat2 = relu(W2 x[p/2] + b2)                         1x1, 64 -> 64, half resolution (16x64)
//     feat_grid = relu(Wg [feat0; feat1; feat2] + bg)      1x1, 192 -> 64  (cat + nearest Upsample)
//
// A lane owns one full-resolution pixel.  feat_grid never sees feat0/feat1/feat2 in memory: the
// result registers of the three small GEMMs are the B fragments of the 192-deep one (weight slabs
// pre-permuted on the host into the MFMA k-slot order).  feat0 / feat1 are also written out (the
// two 3x3 stride-2 convolutions that follow need pixel neighbourhoods) and feat2 once per 2x2 block.
// Compared with five separate convolutions this removes the re-reads of feat0 / feat1 / feat2 by
// `down_feat` and the launch of `down2`.
//
// Reference: TPS_PP.forward / TPS_PP.grid, mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:560-562,
// 580-585 (down0, down1, down2, up_sample, torch.cat, down_feat).
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
#ifndef FRONT_FENCE
#define FRONT_FENCE 0
#endif
constexpr bool FENCE = FRONT_FENCE;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FrontParams {
    const float* o0; const float* o1;      // (N, 32, H, W)
    const float* x;                        // (N, 64, H/2, W/2)
    const float* w0_s; const float* w1_s;  // [32 k-slots (natural)][64]
    const float* w2_s;                     // [64 k-slots (natural)][64]
    const float* wg_s;                     // [3 blocks][64 k-slots (MFMA order)][64]
    const float* b0; const float* b1; const float* b2; const float* bg;   // (64) each
    float* feat0; float* feat1;            // (N, 64, H, W)
    float* feat2;                          // (N, 64, H/2, W/2)
    float* feat_grid;                      // (N, 64, H, W)
    int H, W;                              // full resolution
};

__device__ __forceinline__ constexpr int feat(int ks, int half)
{
    return 32 * (ks >> 4) + (ks & 3) + 8 * ((ks & 15) >> 2) + 4 * half;
}

// out (64 features, MFMA layout) = relu(slab^T in + bias);  NK k-slot pairs
template <int NK>
__device__ __forceinline__ void dense_relu(const float* __restrict__ slab, const float* __restrict__ bias,
                                           const float (&in)[NK], int half, int l31, float (&out)[32])
{
    f32x16 a0, a1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = 0.0f; a1[i] = 0.0f; }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(slab[(2 * ks + half) * 64 + l31], in[ks], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(slab[(2 * ks + half) * 64 + 32 + l31], in[ks], a1, 0, 0, 0);
        // fence the scheduler every 8 steps: left alone it hoists every weight-fragment read of the
        // whole tile to the top and spills
        if (FENCE && (ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        const float v = (ks < 16 ? a0[ks & 15] : a1[ks & 15]) + bias[feat(ks, half)];
        out[ks] = v > 0.0f ? v : 0.0f;
    }
}

constexpr int kFrontThreads = 512;     // 8 wavefronts = 2 per SIMD: one computes while the other waits on memory
constexpr int kFrontTile = 256;        // pixels per workgroup pass

// g += slab^T f   (64 x 64 block of the 192-deep feat_grid GEMM; f in the MFMA register layout)
__device__ __forceinline__ void accumulate_block(const float* __restrict__ slab, const float (&f)[32], int half,
                                                 int l31, f32x16& g0, f32x16& g1)
{
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(slab[(2 * ks + half) * 64 + l31], f[ks], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(slab[(2 * ks + half) * 64 + 32 + l31], f[ks], g1, 0, 0, 0);
        if (FENCE && (ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ void __launch_bounds__(kFrontThreads)
front_kernel(const FrontParams P)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW0 = smem;                 // 32*64
    float* sW1 = sW0 + 32 * 64;        // 32*64
    float* sW2 = sW1 + 32 * 64;        // 64*64
    float* sWg = sW2 + 64 * 64;        // 3*64*64
    float* sB = sWg + 3 * 64 * 64;     // b0 | b1 | b2 | bg
    const int tid = threadIdx.x;
    for (int i = tid; i < 32 * 64 / 4; i += kFrontThreads) {
        reinterpret_cast<float4*>(sW0)[i] = reinterpret_cast<const float4*>(P.w0_s)[i];
        reinterpret_cast<float4*>(sW1)[i] = reinterpret_cast<const float4*>(P.w1_s)[i];
    }
    for (int i = tid; i < 64 * 64 / 4; i += kFrontThreads) reinterpret_cast<float4*>(sW2)[i] = reinterpret_cast<const float4*>(P.w2_s)[i];
    for (int i = tid; i < 3 * 64 * 64 / 4; i += kFrontThreads) reinterpret_cast<float4*>(sWg)[i] = reinterpret_cast<const float4*>(P.wg_s)[i];
    if (tid < 64) { sB[tid] = P.b0[tid]; sB[64 + tid] = P.b1[tid]; sB[128 + tid] = P.b2[tid]; sB[192 + tid] = P.bg[tid]; }
    __syncthreads();

    const int lane = tid & (kWave - 1), wv = tid / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = P.H * P.W, h2 = P.H >> 1, w2 = P.W >> 1;
    const int n = blockIdx.y;
    const int tiles_per_img = (HW + kFrontTile - 1) / kFrontTile;
    for (int tile = blockIdx.x; tile < tiles_per_img; tile += gridDim.x) {
        // the slabs do not change from tile to tile; hidden from the optimiser, which would otherwise hoist their reads
        // out of this loop and spill them (tpspp_dgab.hip)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const float* tW0 = sW0 + opaque; const float* tW1 = sW1 + opaque; const float* tW2 = sW2 + opaque;
        const float* tWg = sWg + opaque; const float* tB = sB + opaque;
        const int px = tile * kFrontTile + wv * 32 + l31;
        const bool live = px < HW;
        const int pxc = live ? px : HW - 1;
        const int y = pxc / P.W, xx = pxc - y * P.W;
        const int par = (y >> 1) * w2 + (xx >> 1);             // parent pixel at half resolution
        // ---- addressing: wave-uniform base (+ uniform per-channel term) + one 32-bit lane offset, so the
        // loads / stores use the scalar-base addressing mode and no per-access 64-bit VALU arithmetic ----
        const unsigned in_off = 4u * (unsigned)(half * HW + pxc);                 // channel 2*ks + half
        const unsigned in2_off = 4u * (unsigned)(half * h2 * w2 + par);
        const unsigned out_off = 4u * (unsigned)(4 * half * HW + pxc);             // channel feat(ks, half)
        const unsigned out2_off = 4u * (unsigned)(4 * half * h2 * w2 + par);
        const char* p0 = reinterpret_cast<const char*>(P.o0 + (size_t)n * 32 * HW);
        const char* p1 = reinterpret_cast<const char*>(P.o1 + (size_t)n * 32 * HW);
        const char* p2 = reinterpret_cast<const char*>(P.x + (size_t)n * 64 * h2 * w2);
        const size_t cstride = (size_t)HW * 4, cstride2 = (size_t)h2 * w2 * 4;
        float i0[16], i1[16], i2[32];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) i0[ks] = *reinterpret_cast<const float*>(p0 + 2 * ks * cstride + in_off);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) i1[ks] = *reinterpret_cast<const float*>(p1 + 2 * ks * cstride + in_off);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) i2[ks] = *reinterpret_cast<const float*>(p2 + 2 * ks * cstride2 + in2_off);

        f32x16 g0, g1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { g0[i] = 0.0f; g1[i] = 0.0f; }
        float f[32];
        // feat0: compute, store, fold into feat_grid; then feat1, feat2 -- one feature set live at a time
        dense_relu<16>(tW0, tB, i0, half, l31, f);
        if (live && P.feat0) {                              // (feat0 = feat1 = NULL: tpspp_down_fused_f32_fwd recomputes them)
            char* q = reinterpret_cast<char*>(P.feat0 + (size_t)n * 64 * HW);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) *reinterpret_cast<float*>(q + feat(ks, 0) * cstride + out_off) = f[ks];
        }
        accumulate_block(tWg, f, half, l31, g0, g1);
        dense_relu<16>(tW1, tB + 64, i1, half, l31, f);
        if (live && P.feat1) {
            char* q = reinterpret_cast<char*>(P.feat1 + (size_t)n * 64 * HW);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) *reinterpret_cast<float*>(q + feat(ks, 0) * cstride + out_off) = f[ks];
        }
        accumulate_block(tWg + 64 * 64, f, half, l31, g0, g1);
        dense_relu<32>(tW2, tB + 128, i2, half, l31, f);
        if (live && ((y | xx) & 1) == 0) {                      // one lane of each 2x2 block keeps feat2
            char* q = reinterpret_cast<char*>(P.feat2 + (size_t)n * 64 * h2 * w2);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) *reinterpret_cast<float*>(q + feat(ks, 0) * cstride2 + out2_off) = f[ks];
        }
        accumulate_block(tWg + 2 * 64 * 64, f, half, l31, g0, g1);
        if (live) {
            char* q = reinterpret_cast<char*>(P.feat_grid + (size_t)n * 64 * HW);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const float v = (ks < 16 ? g0[ks & 15] : g1[ks & 15]) + tB[192 + feat(ks, half)];
                *reinterpret_cast<float*>(q + feat(ks, 0) * cstride + out_off) = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_front_fwd(const float* outs0, const float* outs1, const float* x,
                                 const float* w0_slab, const float* b0, const float* w1_slab, const float* b1,
                                 const float* w2_slab, const float* b2, const float* wg_slab, const float* bg,
                                 float* feat0, float* feat1, float* feat2, float* feat_grid,
                                 int N, int H, int W, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(outs0 && outs1 && x && w0_slab && b0 && w1_slab && b1 && w2_slab && b2 && wg_slab && bg &&
                  feat2 && feat_grid && (feat0 != nullptr) == (feat1 != nullptr), "tpspp_front_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && N <= 65535,
                  "tpspp_front_fwd: need even H, W");
    if (N == 0) return TPSPP_OK;
    FrontParams P;
    P.o0 = outs0; P.o1 = outs1; P.x = x; P.w0_s = w0_slab; P.w1_s = w1_slab; P.w2_s = w2_slab; P.wg_s = wg_slab;
    P.b0 = b0; P.b1 = b1; P.b2 = b2; P.bg = bg;
    P.feat0 = feat0; P.feat1 = feat1; P.feat2 = feat2; P.feat_grid = feat_grid; P.H = H; P.W = W;
    const size_t lds = (size_t)(2 * 32 * 64 + 64 * 64 + 3 * 64 * 64 + 256) * sizeof(float);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const int tiles = (H * W + kFrontTile - 1) / kFrontTile;
    // a few tiles per workgroup so the 81 KB of weight slabs are staged once per several tiles
    const int gx = tiles >= 8 ? (tiles + 3) / 4 : tiles;
    hipLaunchKernelGGL(front_kernel, dim3((unsigned)gx, (unsigned)N), dim3(kFrontThreads), lds,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_front_fwd");
}
