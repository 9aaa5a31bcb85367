// The backbone's stem in the bf16 configuration as its own kernel (round 6): conv3x3(3 -> 32) + BatchNorm (folded) + ReLU on the
// fp32 image, bf16 NCHW result (the map TPS++ receives as outs[0]).
//
// Replaces (when the backbone runs bf16): self.conv1 / self.bn1 / self.relu of ResNetABI_v2_large.forward,
// mmocr/models/textrecog/backbones/resnet_v2_large.py:183-186 (the stem before the first stage).
//
// Why its own kernel.  On the tiled kernel this layer took 189 us per 512 images for 25 MB in and 134 MB out (~35 us at the
// memory's speed): 3 input channels ride in a 16-channel chunk, so every patch position was staged with 16 four-byte loads of
// which 13 re-read the last channel to be zeroed, and a 64-channel tile carried 32 empty output channels through the epilogue.
// Here
//   * persistent workgroups walk tiles of 4 rows x 128 columns; the patch (6 x 130 positions x 3 channels) is loaded with
//     coalesced row loads -- a thread owns <= 4 positions, 3 loads each, the NEXT tile's in flight during this tile's products --,
//     rounded to bf16 and laid down as one 16-byte unit per position {c0, c1, c2, 0 ...};
//   * the roles of the matrix operands are swapped -- D[pixel][cout] = X W -- so that a lane ends up with 4 CONSECUTIVE PIXELS of
//     one output channel per register quad: 8-byte pieces of an NCHW row.  A wavefront owns one 128-pixel row; its 32 x 128
//     results go through a per-wavefront LDS tile [channel][128 pixels] and leave as 16-byte stores, 256 contiguous bytes per
//     channel row;
//   * the 9 weight fragments (one per tap) live in registers for the life of the workgroup; the k-half of the zero-padded
//     channels 8 - 15 reads a zero unit.
// Same products in the same order as the tiled kernel (one 16-deep k-step per tap, taps ascending, fp32 accumulation, bias, ReLU,
// one rounding): BIT-IDENTICAL results (tests/test_gpu_conv_bf16.py::test_stem_kernel_is_the_tiled_kernel_bit_for_bit;
// tpspp_conv_set_tuning bit 2 switches it off).
// Bound: HBM (25 MB + 134 MB per 512 images).
#include "tpspp_conv_bf16_impl.h"

namespace {

constexpr int kSW = 128, kSTH = 4;                   // tile: 4 rows x 128 columns
constexpr int kSPW = kSW + 2, kSPH = kSTH + 2, kSPS = kSPW * kSPH;      // 780 patch positions
constexpr int kSNP = (kSPS + 255) / 256;             // positions per thread
constexpr int kOPitch = 66;                          // dwords per channel row of a wavefront's output tile (64 + 2: 8-byte
                                                     // writes of 32 lanes, one row each, fall on all 64 banks)

struct StemP {
    const float* in; const u32x4* wt; const float* bias; unsigned short* out;
    int N, C, H, ntiles, relu;
};

// BLK: blocked bf16 output (N, 4, H, W, 8) instead of NCHW -- when only convolutions read the map (the backbone hands TPS++ its
// first two maps blocked when TPS++ says it takes them: tps_pp_amd/resnet_v2_large.py).  The products then run in the usual
// orientation (D[cout][pixel]: a lane = a pixel, its registers = channels), results leave as 16-byte units through
// v_permlane32_swap, no LDS tile.
template <bool BLK>
__global__ void __launch_bounds__(256, 2)
conv_stem_bf16_kernel(const StemP P)
{
    __shared__ u32x4 sP[kSPS + 1];                   // + the zero unit
    __shared__ unsigned sO[4][32 * kOPitch];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int H = P.H, plane = H * kSW;
    const int tiles_per_img = H / kSTH;

    // weights as the B operand: lane (cout l31, k half) holds the tap's 8 k of its output channel -- the arranged weight's unit
    // (tap * 2 + half) * 64 + l31 of the only (cout tile, chunk) slab
    bf16x8 wf[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[t] = __builtin_bit_cast(bf16x8, P.wt[(t * 2 + half) * BN + l31]);
    const float bias = P.bias ? P.bias[l31] : 0.0f;
    float bq[4][4];                                          // BLK: this lane's 16 channels 8 g + 4 half + e
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[g][e] = (BLK && P.bias) ? P.bias[8 * g + 4 * half + e] : 0.0f;
    if (tid == 0) sP[kSPS] = u32x4{0u, 0u, 0u, 0u};

    // this thread's patch positions (fixed over the tiles)
    int ppy[kSNP], ppx[kSNP];
#pragma unroll
    for (int i = 0; i < kSNP; ++i) {
        const int e = tid + 256 * i;
        ppy[i] = e / kSPW;
        ppx[i] = e - ppy[i] * kSPW;
    }
    float rv[kSNP][3];
    auto prefetch = [&](int tile) {
        const int n = tile / tiles_per_img, y0 = (tile - n * tiles_per_img) * kSTH;
        const float* ip = P.in + (size_t)n * P.C * plane;
#pragma unroll
        for (int i = 0; i < kSNP; ++i) {
            const int iy = y0 + ppy[i] - 1, ix = ppx[i] - 1;
            const bool ok = tid + 256 * i < kSPS && iy >= 0 && iy < H && ix >= 0 && ix < kSW;
            const int off = ok ? iy * kSW + ix : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ip[(size_t)(c < P.C ? c : 0) * plane + off];         // (unconditional load, select behind it)
                rv[i][c] = (ok && c < P.C) ? v : 0.0f;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < kSNP; ++i) {
            const int e = tid + 256 * i;
            if (e < kSPS) sP[e] = u32x4{pack2_bf16(rv[i][0], rv[i][1]), pack2_bf16(rv[i][2], 0.0f), 0u, 0u};
        }
    };

    int tile = blockIdx.x;
    if (tile < P.ntiles) prefetch(tile);
    for (; tile < P.ntiles; tile += gridDim.x) {
        __syncthreads();                                     // the previous tile's fragment reads are done
        commit();
        __syncthreads();
        const int next = tile + (int)gridDim.x;
        if (next < P.ntiles) prefetch(next);                 // in flight during the products and the stores
        const int n = tile / tiles_per_img, y0 = (tile - n * tiles_per_img) * kSTH;
        // row wv of the tile: 4 fragments of 32 pixels; activations as the A operand (lane = pixel; the upper k half is zero)
        f32x16 acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int pos = half ? kSPS : (wv + ky) * kSPW + 32 * f + l31 + kx;
                const bf16x8 a = __builtin_bit_cast(bf16x8, sP[pos]);
                if constexpr (BLK) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap], a, acc[f], 0, 0, 0);
                else acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[tap], acc[f], 0, 0, 0);
            }
        }
        if constexpr (BLK) {
            // D[cout][pixel]: register 4 g + e of lane (pixel l31, half) = channel 8 g + 4 half + e; the two half-wavefronts hold
            // the halves of a 16-byte unit
            unsigned short* const ob = P.out + (((size_t)n * 4 * H + (y0 + wv)) * kSW) * 8;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                tpspp_u32x2 pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[f][4 * g + e] + bq[g][e];
                        if (P.relu) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    }
                    pk[g][0] = pack2_bf16(v[0], v[1]); pk[g][1] = pack2_bf16(v[2], v[3]);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const tpspp_u32x2 d0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    const tpspp_u32x2 d1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                    *reinterpret_cast<u32x4*>(ob + ((size_t)(g + half) * plane + 32 * f + l31) * 8) = unit;
                }
            }
            continue;
        }
        // D[pixel][cout]: register 4 g + e of lane (cout l31, half) = pixel 32 f + 8 g + 4 half + e: four consecutive pixels
        unsigned* const ot = sO[wv];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[f][4 * g + e] + bias;
                    if (P.relu) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                }
                tpspp_u32x2 pk; pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]);
                *reinterpret_cast<tpspp_u32x2*>(ot + l31 * kOPitch + (32 * f + 8 * g + 4 * half) / 2) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is private to the wavefront: no barrier)
        unsigned short* const orow = P.out + ((size_t)n * 32 * H + (y0 + wv)) * kSW;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = lane + 64 * i, co = u >> 4, chunk = u & 15;        // 16 pieces of 8 pixels per channel row
            const unsigned* src = ot + co * kOPitch + chunk * 4;
            u32x4 val; val[0] = src[0]; val[1] = src[1]; val[2] = src[2]; val[3] = src[3];
            *reinterpret_cast<u32x4*>(orow + (size_t)co * plane + chunk * 8) = val;
        }
    }
}

}  // namespace

namespace tpspp {

// true when the stem kernel took the layer: 3x3 stride 1, ONE fp32 NCHW source of <= 3 channels at its own resolution, 32 output
// channels, bf16 NCHW or blocked bf16 output, bias / ReLU only, 128 columns, rows a multiple of 4
bool conv_stem_launch(const BParams& P, hipStream_t st)
{
    if (P.nsrc != 1 || P.src[0].f32 != 1 || P.src[0].lh || P.src[0].lw || (P.out_f32 != 0 && P.out_f32 != 2) || P.res_mode || P.post_scale ||
        P.relu > 1) return false;
    if (P.Cin > 3 || P.Cout != 32 || P.Wo != kSW || P.Wi != kSW || P.Ho != P.Hi || (P.Ho % kSTH) || P.nchunks != 1) return false;
    if ((reinterpret_cast<size_t>(P.out) | reinterpret_cast<size_t>(P.wt)) & 15) return false;
    const long nt = (long)P.N * (P.Ho / kSTH);
    if (nt <= 0 || nt > 0x3fffffffL) return false;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        return false;
    }
    StemP S;
    S.in = reinterpret_cast<const float*>(P.src[0].p); S.wt = P.wt; S.bias = P.bias; S.out = reinterpret_cast<unsigned short*>(P.out);
    S.N = P.N; S.C = P.Cin; S.H = P.Ho; S.ntiles = (int)nt; S.relu = P.relu;
    const long slots = (long)ncu * 3;
    if (P.out_f32 == 2) hipLaunchKernelGGL(conv_stem_bf16_kernel<true>, dim3((unsigned)(nt < slots ? nt : slots)), dim3(256), 0, st, S);
    else hipLaunchKernelGGL(conv_stem_bf16_kernel<false>, dim3((unsigned)(nt < slots ? nt : slots)), dim3(256), 0, st, S);
    return true;
}

}  // namespace tpspp
