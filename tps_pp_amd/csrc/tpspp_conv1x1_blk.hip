// 1x1 convolution (stride 1) between maps in the BLOCKED bf16 layout (N, C/8, H, W, 8): the first convolution of the
// backbone's BasicBlocks on the 16x64 and 8x32 levels in the bf16 configuration (64 -> 64, 128 -> 128, 256 -> 256, ...).
// Replaces (when the module runs bf16): conv1x1 + BN + ReLU of BasicBlock.forward,
// mmocr/models/textrecog/layers/conv_layer.py as used by backbones/resnet_v2_large.py (ResNetABI_v2_large).
//
// Same arithmetic as conv_tiled_bf16_kernel (channels ascending through v_mfma_f32_32x32x16_bf16, one accumulator per
// 32 x 32 tile, bias + ReLU in fp32, one rounding), so the two agree bit for bit; what changes is the data movement.  The
// tiled kernel is built for 3x3 patches: 64 output channels per workgroup (a 256-wide layer stages every activation four
// times), patch staging through LDS, and it took 54 us for the 134 MB of a 256 -> 256 layer at batch 512 (22 us at 6 TB/s).
// For a 1x1 kernel on a blocked map nothing needs staging:
//   * a 16-byte unit of the map (8 channels of a pixel) IS the B operand's lane image: lane (pixel, k half) loads the unit of
//     channel group 2 ks + half straight from global memory, 512 contiguous bytes per half-wavefront; all of a wavefront's
//     Cin / 16 loads are in flight together;
//   * a wavefront owns 32 pixels and ALL output channels (up to 8 accumulators), so every activation is read once;
//   * the whole weight (<= 128 KB, the convolution's arranged copy) sits in LDS for the life of the persistent workgroup:
//     an A fragment is one ds_read_b128;
//   * results leave as 16-byte units of the blocked output (v_permlane32_swap pairs the two half-wavefronts' halves).
// Bound: HBM.
#include "tpspp_conv_bf16_impl.h"
#include <cstdlib>

namespace {

constexpr int kMaxLds = 160 * 1024;

// NT: 32-output tiles (Cout / 32); NKS: 16-channel k-steps (Cin / 16); NW: wavefronts per workgroup = 32-pixel fragments per tile.
// NW = 8 (256 registers per wavefront): the next tile's units are requested once this tile's products are issued.
// NW = 4 (512 registers; the 256 -> 256 layer, whose 128 accumulator + 64 operand registers do not leave room at 256): the
// next tile's units are requested BEFORE this tile's products and land under them.
template <int NT, int NKS, int NW>
__global__ void __launch_bounds__(NW * 64, 1)
conv1x1_blk_kernel(const BParams P, int ntiles, int cg_total, int cg0)
{
    // (round 6: a launch may own a SLICE of the output channels -- groups cg0 .. cg0 + 4 NT - 1 of cg_total -- so that layers whose
    // whole weight does not fit the LDS (256 -> 512, 512 -> 512) run as 2 / 4 launches of this kernel, each streaming the
    // activations once; P.wt / P.bias then point at the slice; and a tile is 32 NW pixels of the flat (image, pixel) index, so
    // maps smaller than a tile (4x16) are taken as well)
    extern __shared__ u32x4 sAll[];
    constexpr int WUNITS = ((NT + 1) / 2) * 64 * NKS * 2;   // (Cout rounded up to whole 64-channel tiles) * Cin / 8 units of 16 bytes
    u32x4* const sW = sAll;
    float* const sBias = reinterpret_cast<float*>(sAll + WUNITS);
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < WUNITS; i += NW * 64) sW[i] = P.wt[i];
    for (int i = tid; i < NT * 32; i += NW * 64) sBias[i] = P.bias ? P.bias[i] : 0.0f;
    __syncthreads();

    const int HW = P.Ho * P.Wo;
    constexpr int TPX = NW * 32;                             // pixels per tile
    constexpr bool EARLY = NW == 4;
    const u32x4* const src = reinterpret_cast<const u32x4*>(P.src[0].p);
    constexpr int nch = NKS / 2;                             // 32-channel chunks of the arranged weight
    const bool relu1 = P.relu == 1;

    u32x4 b[NKS], bn[EARLY ? NKS : 1];
    auto fetch = [&](int tile, u32x4* dst) {
        const int gp = tile * TPX + 32 * wv + l31;           // (HW is a multiple of 32: a fragment's pixels share an image)
        const int n = gp / HW, px = gp - n * HW;
        const u32x4* s = src + ((size_t)n * (2 * NKS) + half) * HW + px;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) dst[ks] = s[(size_t)(2 * ks) * HW];
    };
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile, b);
    for (; tile < ntiles; tile += gridDim.x) {
        const int gp = tile * TPX + 32 * wv + l31;
        const int n = gp / HW, px = gp - n * HW;
        const int next = tile + (int)gridDim.x;
        if (EARLY && next < ntiles) fetch(next, bn);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        // the weight does not change from tile to tile; the optimiser must not see that (it would hoist NT x NKS fragments --
        // up to 512 registers -- out of this loop and spill them), nor read more than one k-step's fragments ahead
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const u32x4* const w = sW + opaque;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 B = __builtin_bit_cast(bf16x8, b[ks]);
            bf16x8 A[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                // arranged weight: [64-output tile][32-channel chunk][k group of the chunk (4)][64 outputs][8]
                A[t] = __builtin_bit_cast(bf16x8, w[(((t >> 1) * nch + (ks >> 1)) * 4 + 2 * (ks & 1) + half) * 64 + 32 * (t & 1) + l31]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t], B, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EARLY) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) b[ks] = bn[ks];
        } else if (next < ntiles) {
            fetch(next, b);                                  // in flight under the epilogue and the other wavefronts' products
        }
        unsigned short* const ob = reinterpret_cast<unsigned short*>(P.out) + (((size_t)n * cg_total + cg0) * HW + px) * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            tpspp_u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[t][4 * g + e] + sBias[32 * t + 8 * g + 4 * half + e];
                    if (relu1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                }
                pk[g][0] = pack2_bf16(v[0], v[1]); pk[g][1] = pack2_bf16(v[2], v[3]);
            }
            // the two half-wavefronts hold the two halves of a 16-byte unit: after the swap the lower one owns the unit of
            // channel group 4 t + g, the upper one that of 4 t + g + 1
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const tpspp_u32x2 d0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const tpspp_u32x2 d1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                *reinterpret_cast<u32x4*>(ob + (size_t)(4 * t + g + half) * HW * 8) = unit;
            }
        }
    }
}

// SL: launches per layer, each owning NT x 32 of the SL x NT x 32 output channels
template <int NT, int NKS, int SL = 1>
bool launch(const BParams& P0, hipStream_t st)
{
    constexpr int NW = (NT * 16 + NKS * 4 > 160) ? 4 : 8;      // accumulators + operands per lane
    const size_t lds = (size_t)((NT + 1) / 2) * 64 * NKS * 2 * 16 + (size_t)NT * 32 * 4;
    if (lds > (size_t)kMaxLds) return false;
    const int HW = P0.Ho * P0.Wo;
    const long total = (long)P0.N * HW;
    if (HW % 32 || total % (NW * 32)) return false;
    const long nt = total / (NW * 32);
    if (nt <= 0 || nt > 0x3fffffffL) return false;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        return false;
    }
    static bool attr[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_blk_kernel<NT, NKS, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    }
    for (int sl = 0; sl < SL; ++sl) {
        BParams P = P0;
        // arranged weight: [64-output tile][32-channel chunk][4 k groups][64 outputs][8]: a slice of NT x 32 outputs is NT / 2 tiles
        P.wt = P0.wt + (size_t)sl * (NT / 2) * (NKS / 2) * 4 * 64;
        if (P0.bias) P.bias = P0.bias + sl * NT * 32;
        hipLaunchKernelGGL((conv1x1_blk_kernel<NT, NKS, NW>), dim3((unsigned)(nt < ncu ? nt : ncu)), dim3(NW * 64), lds, st, P, (int)nt,
                           SL * NT * 4, sl * NT * 4);
    }
    return true;
}

}  // namespace

namespace tpspp {

// true when the kernel took the layer: 1x1, stride 1, one blocked bf16 source at its own resolution, blocked bf16 output,
// bias / ReLU only, Cin and Cout in {64, 128, 256} (+ 256 -> 512, 512 -> 512 as output-channel slices), whole 32 NW-pixel tiles
// of the flat (image, pixel) index
bool conv1x1_blk_launch(const BParams& P, hipStream_t st)
{
    if (P.nsrc != 1 || P.src[0].f32 != 2 || P.out_f32 != 2 || P.res || P.res_mode || P.post_scale || P.relu > 1) return false;
    if (P.src[0].lh || P.src[0].lw || P.src[0].H != P.Ho || P.src[0].W != P.Wo || P.src[0].C != P.Cin) return false;
    if ((reinterpret_cast<size_t>(P.src[0].p) | reinterpret_cast<size_t>(P.out) | reinterpret_cast<size_t>(P.wt)) & 15) return false;
#define TPSPP_C1(CI, CO) if (P.Cin == CI && P.Cout == CO) return launch<CO / 32, CI / 16>(P, st);
    TPSPP_C1(64, 64) TPSPP_C1(64, 128) TPSPP_C1(128, 128) TPSPP_C1(128, 256) TPSPP_C1(256, 256)
    TPSPP_C1(32, 32) TPSPP_C1(32, 64)             // (round 6: the backbone's first stage, when its maps are blocked)
#undef TPSPP_C1
    // the first layer of the last stage: the weight of an output-channel slice in LDS, one launch per slice (132 -> 82 us at batch
    // 512).  Not 512 -> 512 on the 4x16 maps: four slices of 131 KB for ONE 128-pixel tile per workgroup measured 74 us against the
    // tiled kernel's 55 (TPSPP_C1X1_512=1 selects it for the bit-identity test).
    if (P.Cin == 256 && P.Cout == 512) return launch<8, 16, 2>(P, st);
    static const bool c512 = getenv("TPSPP_C1X1_512") != nullptr;
    if (c512 && P.Cin == 512 && P.Cout == 512) return launch<4, 32, 4>(P, st);
    return false;
}

}  // namespace tpspp

// ---- blocked bf16 (N, C/8, HW, 8) -> NCHW bf16 (N, C, HW) (round 6) -----------------------------------------------------------
// For the one map of the bf16 backbone that both convolutions and the sampler read (the second stage's result: TPS++'s down2
// takes the blocked form, the warp needs channel planes): the 3x3 layer that produces it runs on the persistent kernel (blocked
// output, 52 us against 115 us for the tiled kernel's NCHW epilogue) and this copy makes the planes (67 MB in, 67 MB out).
// A wavefront moves 64 pixels of one channel group: 64 units of 16 bytes in (1 KB contiguous), through its own 1-KB LDS tile, out
// as eight 16-byte pieces per channel plane (128 contiguous bytes per plane).
namespace {
__global__ void __launch_bounds__(256)
blocked_to_nchw_bf16_kernel(const u32x4* __restrict__ in, unsigned short* __restrict__ out, int CG, int HW, long nseg)
{
    __shared__ unsigned short tile[4][64 * 8 + 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long seg = (long)blockIdx.x * 4 + wv;                // (image, channel group, 64-pixel segment)
    if (seg >= nseg) return;
    const int spp = HW / 64;                                   // segments per plane
    const long plane_id = seg / spp;                           // n * CG + cg
    const int p0 = (int)(seg - plane_id * spp) * 64;
    const u32x4 u = in[plane_id * HW + p0 + lane];
    unsigned short* t = tile[wv];
    *reinterpret_cast<u32x4*>(t + lane * 8) = u;               // [pixel][8 channels]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the tile is private to the wavefront)
    const int c = lane >> 3, chunk = lane & 7;                 // this lane: channel c of the group, pixels 8 chunk .. 8 chunk + 7
    unsigned v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = (unsigned)t[(8 * chunk + 2 * i) * 8 + c] | ((unsigned)t[(8 * chunk + 2 * i + 1) * 8 + c] << 16);
    const long n = plane_id / CG;
    const int cg = (int)(plane_id - n * CG);
    unsigned short* o = out + ((n * CG + cg) * 8 + c) * (long)HW + p0 + 8 * chunk;
    *reinterpret_cast<u32x4*>(o) = u32x4{v[0], v[1], v[2], v[3]};
}
}  // namespace

TPSPP_EXPORT int tpspp_blocked_to_nchw_bf16(const void* in_blocked, int N, int C, int HW, void* out_nchw, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in_blocked && out_nchw && N >= 0 && C > 0 && HW > 0, "tpspp_blocked_to_nchw_bf16: bad argument");
    TPSPP_REQUIRE(C % 8 == 0 && HW % 64 == 0, "tpspp_blocked_to_nchw_bf16: channels a multiple of 8, pixels per plane a multiple of 64");
    TPSPP_REQUIRE(((reinterpret_cast<size_t>(in_blocked) | reinterpret_cast<size_t>(out_nchw)) & 15) == 0,
                  "tpspp_blocked_to_nchw_bf16: 16-byte aligned tensors");
    if (N == 0) return TPSPP_OK;
    const long nseg = (long)N * (C / 8) * (HW / 64);
    TPSPP_REQUIRE((nseg + 3) / 4 <= 0x7fffffffL, "tpspp_blocked_to_nchw_bf16: too large");
    hipLaunchKernelGGL(blocked_to_nchw_bf16_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, tpspp::as_stream(stream),
                       reinterpret_cast<const u32x4*>(in_blocked), reinterpret_cast<unsigned short*>(out_nchw), C / 8, HW, nseg);
    return tpspp::check_launch("tpspp_blocked_to_nchw_bf16");
}
