// 3x3 stride-1 bf16 convolution with 128 x 64 WAVEFRONT tiles for the wide blocked layers of the backbone (round 6): the second
// convolution of the BasicBlocks of stages 3 - 5 in the bf16 configuration -- 128 -> 128 and 256 -> 256 on 8x32 maps,
// 512 -> 512 on 4x16 maps (BASELINE.json configs[4]; 1.6 of the backbone's 4.4 ms at batch 512).
//
// Replaces (when the module runs bf16): conv3x3 + BN + residual + ReLU of BasicBlock.forward,
// mmocr/models/textrecog/backbones/resnet_v2_large.py:109-129 with layers/conv_layer.py:12-33.
//
// Why another kernel (round-5 review, item 2).  conv_tiled_bf16_kernel gives a wavefront 64 pixels x 64
// channels and reads BOTH operands from LDS: per 16-deep k-step 2 A + 2 B fragments for 4 matrix instructions -- one
// ds_read_b128 per v_mfma_f32_32x32x16_bf16.  A wavefront fragment read is 1 KB and the LDS moves 128 B per clock and CU, a
// matrix instruction occupies its SIMD for 32 clocks (8 passes x 4): four SIMDs x one read per instruction IS the LDS
// bandwidth, so the pipe cannot pass ~50 % however the loop is scheduled (measured: 34 - 38 % on these layers), and the kernel
// paid two barriers + a register-staged refill per 36 instructions on top.  Here
//   * a wavefront owns 128 pixels x 64 output channels: 8 accumulators (128 registers), 4 B + 2 A fragments per tap;
//   * the A operand (weights) never touches LDS: a lane's fragment of the arranged weight ([cout tile][chunk][tap][k half][64
//     cout][8 k]: tpspp_conv2d_bf16_fwd's layout, unchanged) is one 16-byte global load, 512 contiguous bytes per
//     half-wavefront, served by L2 / L1 (every workgroup of the launch streams the same 0.3 - 4.7 MB); the loads run FIVE taps
//     ahead of their use through a ring of six register slots -- 0.5 LDS reads per matrix instruction are left;
//   * the B operand (the blocked map's 16-byte units = units of the channel-innermost patch) arrives by LDS-DMA
//     (global_load_lds_dwordx4, 64 patch positions per instruction, padding positions pointing at a zero unit) into a ring of
//     three 16-channel buffers (11 - 14 KB each), two chunks ahead; every wavefront issues its quarter of a chunk's DMA, one
//     s_barrier per chunk (raw: a __syncthreads would also wait for the weight prefetch);
//   * 4 wavefronts per workgroup (2 pixel halves x 2 channel halves: 256 pixels x 128 channels), <= 256 registers, ~40 KB of
//     LDS: two workgroups per CU, one wavefront of each per SIMD.
// Same products in the same order as the tiled kernel (chunk by chunk, tap by tap, one k-step per tap): BIT-IDENTICAL results
// (tests/test_gpu_conv_bf16.py::test_conv3_wide_kernel_is_the_tiled_kernel_bit_for_bit; tpspp_conv_set_tuning bit 2 switches it off).
// Loads only until the epilogue: a wavefront's loads return in issue order, so a weight fragment that has arrived proves
// every older DMA of that wavefront complete (the explicit counted wait in front of the barrier says the same).
// Measured (batch 512, 30 launches back to back, scripts/debug/bench_wide.py; tiled kernel -> this one): 128 -> 128 @8x32 58.7 ->
// 40.8 us, 256 -> 256 @8x32 196 -> 143 us (1.08 PFLOP/s), 512 -> 512 @4x16 192 -> 148 us; matrix pipe busy 54 - 55 % on the two
// large shapes before the last two changes (SQ_VALU_MFMA_BUSY_CYCLES at the ~1.75 GHz the chip holds under this load; tiled:
// 34 - 38 %).  Where the rest goes, by switching parts off (-DTPSPP_WIDE_LAB, results then wrong): no weight loads inside the
// loop 121 us, the same loads but always hitting L1 148 us (it is the vector-memory -> register path itself, not L2: 16 KB per
// wavefront and chunk), no DMA 139 us, no epilogue 132 us, neither loads nor epilogue 102 us (1.5 PFLOP/s: what the loop
// with its LDS reads and one barrier per chunk delivers); the weight loads 8 taps ahead instead of 5, or spread between the
// matrix instructions: no change.  The epilogue requests all of its 32 residual units before the first result is formed (it
// was 12 % with the loads next to their use), a chunk's DMA instructions are issued one per tap.
// Bound: the matrix pipe (0.5 LDS reads and 0.25 global 16-byte loads per instruction) -- at 43 - 55 % of it.
#include "tpspp_conv_bf16_impl.h"
#include <cstdlib>

namespace {

// KS = 3: a chunk is 16 channels (2 groups of 8) x 9 taps, one 16-deep k-step per tap; KS = 1 (round 6, late: the 1x1 layers of
// the last stage): a chunk is 32 channels (4 groups) = 2 k-steps, no halo.  Either way step u of chunk c is k-step c U + u of the
// whole sum and its A fragments sit 128 (c U + u) units into the cout tile's arranged weight.
template <int KS> struct WK {
    static constexpr int U = KS == 3 ? 9 : 2;        // inner steps (matrix k-steps) per chunk
    static constexpr int KG = KS == 3 ? 2 : 4;       // channel groups of 8 per chunk
    static constexpr int PD = KS == 3 ? 5 : 3;       // k-steps the weight loads run ahead; ring of PD + 1 slots divides 2 U
};
// (PD = taps the weight loads run ahead of their use; ring of PD + 1 register slots, PD + 1 dividing the 18 taps of an
// unrolled chunk pair: 5 -> 6 slots, 8 -> 9 slots)
constexpr int kGNB = 3;                          // patch buffers

__device__ u32x4 g_zero_unit_wide;               // what padding positions read

__device__ __forceinline__ void wdma16(const void* g, unsigned lds_byte)
{
    // (s_nop: a SALU write of M0 needs a wait state before an LDS-DMA reads it; m0 declared clobbered -- as in
    // tpspp_conv_bf16_persist.hip)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g) : "memory", "m0");
}

template <int TH, int TW, int NI, int KS = 3>
struct WCfg {
    static constexpr int kGKG = WK<KS>::KG;
    static constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PS = PH * PW, PSN = NI * PS;
    static constexpr int NPI = (PSN + kWave - 1) / kWave;      // DMA instructions per channel group
    static constexpr int NDMA = kGKG * NPI;                     // per chunk
    static constexpr int MAXJ = (NDMA + 3) / 4;                 // per wavefront and chunk
    // the upper half-wavefront's k group sits HS units behind the lower one's: HS = 4 mod 8 keeps a ds_read_b128's lanes l and
    // l + 32 on different banks (PSN = 432 of the 4x16 geometry is 0 mod 8: every fragment read was a 2-way conflict -- half of
    // the LDS pipe's busy cycles by SQ_LDS_BANK_CONFLICT)
    static constexpr int HS = PSN + ((12 - PSN % 8) % 8);
    static constexpr int BUFU = kGKG * HS;                      // units per patch buffer
    static_assert(NI * TH * TW == 256 && PSN >= kWave && TW % 16 == 0 && (32 % TW == 0 || TW % 32 == 0), "tile = 256 pixels");
};

// EPI bit 0: + blocked bf16 residual (res_mode 1 / 2); bit 1: fp32 NCHW output (the backbone's last block) instead of blocked bf16.
// LAB (timing experiments only, TPSPP_WIDE_LAB in the environment; results are WRONG for LAB != 0): bit 0 no barrier per chunk,
// bit 1 no weight loads inside the loop, bit 2 no DMA inside the loop, bit 3 no epilogue stores / residual loads, bit 4 the
// weight loads re-read the same 8 KB (L1 hits); TPSPP_WIDE_LAB=108: the product kernel with the weight loads 8 taps ahead
template <int TH, int TW, int NI, int EPI, int kGPD, int LAB = 0, int KS = 3>
__global__ void __launch_bounds__(256, 2)
conv3_wide_kernel(const BParams P)
{
    using Cfg = WCfg<TH, TW, NI, KS>;
    constexpr int PW = Cfg::PW, PS = Cfg::PS, PSN = Cfg::PSN, NPI = Cfg::NPI, MAXJ = Cfg::MAXJ, BUFU = Cfg::BUFU, HS = Cfg::HS;
    constexpr int kGKG = Cfg::kGKG, U = WK<KS>::U;
    constexpr int kGRing = kGPD + 1;
    constexpr int DPS = (MAXJ + U - 1) / U;                     // DMA instructions a wavefront issues per inner step
    static_assert((2 * U) % kGRing == 0, "the ring must turn a whole number of times per unrolled chunk pair");
    __shared__ u32x4 sB[kGNB * BUFU];

    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ph = wv & 1, ch = wv >> 1;                       // pixel half, channel half of the 256 x 128 workgroup tile
    // block -> (image group, 128-channel tile), XCD-aware: blocks are dealt to the 8 XCDs round-robin, and the CT workgroups
    // that share an image group's patch should share an L2 -- block L = 8 (CT q + t) + x is tile t of image group 8 q + x, so the
    // CT readers of a patch are the blocks x, x + 8, ... of one XCD (PMC, 512 -> 512 @4x16: 179 MB fetched per launch with the
    // tiles of a group on four XCDs, against 72 MB algorithmic)
    const int CT = P.Cout >> 7;
    const int L = (int)blockIdx.x, xq = L >> 3;
    const int grp = (xq / CT) * 8 + (L & 7), t128 = xq % CT;
    const int ctile = 2 * t128 + ch;                           // 64-channel tile of the arranged weight
    const int n0 = grp * NI;
    if (n0 >= P.N) return;                                     // (the last group of eight may be incomplete; uniform per block)
    const int nchunks = P.nchunks;                             // even (checked by the launcher)
    const int HW = TH * TW;
    const int CG = P.Cin >> 3;                                 // channel groups of the source

    // ---- this wavefront's share of a chunk's DMA: instruction j = wv + 4 i of the chunk's NDMA -------------------------
    const char* const zero = reinterpret_cast<const char*>(&g_zero_unit_wide);
    const char* dsrc[MAXJ];
    unsigned ddst[MAXJ];
    bool dok[MAXJ];
#pragma unroll
    for (int i = 0; i < MAXJ; ++i) {
        const int j = wv + 4 * i;
        const int g = j / NPI, ii = j - g * NPI;
        constexpr int kLast = PSN - kWave;
        const int start = ii * kWave < kLast ? ii * kWave : kLast;    // (the last window is shifted back: it ends at PSN)
        const int e = start + lane;
        const int im = e / PS, r = e - im * PS;
        const int py = r / PW, px = r - py * PW;
        const int iy = py - (KS - 1) / 2, ix = px - (KS - 1) / 2, n = n0 + im;
        dok[i] = j < Cfg::NDMA && n < P.N && iy >= 0 && iy < TH && ix >= 0 && ix < TW;
        dsrc[i] = reinterpret_cast<const char*>(reinterpret_cast<const u32x4*>(P.src[0].p) +
                                                ((size_t)(dok[i] ? n : 0) * CG + g) * HW + (dok[i] ? iy * TW + ix : 0));
        ddst[i] = (unsigned)((g * HS + start) * 16);
    }
    const unsigned sB0 = (unsigned)(size_t)sB;
    auto dma_one = [&](int c, int i) {                         // this wavefront's i-th DMA instruction of chunk c
        const unsigned base = sB0 + (unsigned)((c % kGNB) * BUFU * 16);
        const size_t coff = (size_t)c * kGKG * HW * 16;
        if (wv + 4 * i < Cfg::NDMA)                              // uniform
            wdma16(dok[i] ? dsrc[i] + coff : zero, base + ddst[i]);
    };
    auto dma_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < MAXJ; ++i) dma_one(c, i);
    };

    // ---- fragments: pixel tp = 128 ph + 32 f + l31 of the tile -> (image, row, column) -----------------------------------
    int fpos[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int tp = 128 * ph + 32 * f + l31;
        const int im = tp / HW, r = tp - im * HW;
        const int ty = r / TW, tx = r - ty * TW;
        fpos[f] = half * HS + im * PS + ty * PW + tx;
    }
    // the weight stream of this wavefront: slab (ctile, chunk) follows (ctile, chunk - 1): tap T of the whole sum is at
    // wA[128 T + 32 h2] (units)
    const u32x4* const wA = P.wt + (size_t)ctile * nchunks * (U * 2 * BN) + half * BN + l31;
    const int taps_total = nchunks * U;

    f32x16 acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][h2][i] = 0.0f;

    u32x4 fa[kGRing][2];
    bf16x8 fb[2][4];

    // ---- prologue: two chunks of patch in flight, the first kGPD taps of weight, then everybody's chunk 0 has landed ------
    dma_chunk(0);
    if (nchunks > 1) dma_chunk(1);
#pragma unroll
    for (int t = 0; t < kGPD; ++t) {
        fa[t][0] = wA[(size_t)t * 128];
        fa[t][1] = wA[(size_t)t * 128 + 32];
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc;
            // buffer (c + 2) % 3 was read during chunk c - 1: every wavefront is past the barrier that ended it, so chunk c + 2 may
            // land there -- its DMA instructions are issued one per tap (taps 0 .. MAXJ - 1), not in a burst
            const bool more = c + 2 < nchunks && !(LAB & 4);
            const u32x4* const pb = sB + (c % kGNB) * BUFU;
            auto fetch_b = [&](int tap, int slot) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int boff = KS == 3 ? ky * PW + kx : tap * 2 * HS;      // (1x1: k-step `tap` reads channel groups 2 tap, 2 tap + 1)
#pragma unroll
                for (int f = 0; f < 4; ++f) fb[slot][f] = __builtin_bit_cast(bf16x8, pb[fpos[f] + boff]);
            };
            fetch_b(0, 0);
#pragma unroll
            for (int tap = 0; tap < U; ++tap) {
                const int tl = cc * U + tap;                      // step of the unrolled pair: ring slot tl % kGRing (compile time)
                const int T = c * U + tap;
                if (more) {
#pragma unroll
                    for (int i = tap * DPS; i < (tap + 1) * DPS && i < MAXJ; ++i) dma_one(c + 2, i);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(LAB & 2)) {   // the weight fragments of tap T + kGPD -> the slot tap T - 1 has just freed
                    const int Tn = T + kGPD;
                    int Tc = Tn < taps_total ? Tn : taps_total - 1;            // (past the end: a harmless re-read)
                    if constexpr (LAB & 16) Tc &= 3;                           // (lab: the same 8 KB again and again -- L1 hits)
                    fa[(tl + kGPD) % kGRing][0] = wA[(size_t)Tc * 128];
                    fa[(tl + kGPD) % kGRing][1] = wA[(size_t)Tc * 128 + 32];
                }
                if (tap + 1 < U) fetch_b(tap + 1, (tap + 1) & 1);
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[tl % kGRing][0]);
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, fa[tl % kGRing][1]);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, fb[tap & 1][f], acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, fb[tap & 1][f], acc[f][1], 0, 0, 0);
                }
                // (measured without effect: spreading the tap's two weight loads and four fragment reads over its eight matrix
                // instructions with sched_group_barrier patterns -- 143.2 / 147.1 us with / without on the 256 -> 256 layer)
                __builtin_amdgcn_sched_barrier(0);
            }
            // end of chunk c: this wavefront's reads of buffer c % 3 are complete (their data has been multiplied); its DMA of
            // chunk c + 1 -- older than the 2 kGPD weight loads and the MAXJ DMA instructions that may still be in flight -- is
            // complete; then the barrier makes both true for the workgroup
            if constexpr (LAB & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * kGPD + MAXJ) : "memory");
        }
    }

    // ---- epilogue: bias, residual, ReLU; blocked bf16 units (the two half-wavefronts hold the halves of a 16-byte unit) -------
    if constexpr (LAB & 8) {
        float s_ = 0.0f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int i = 0; i < 16; ++i) s_ += acc[f][h2][i];
        if (s_ == 123.456f) reinterpret_cast<float*>(P.out)[0] = s_;
        return;
    }
    const bool relu1 = P.relu == 1;
    const int CGo = P.Cout >> 3;
    float bq[2][4][4];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + ctile * BN + 32 * h2 + 8 * g + 4 * half);
            bq[h2][g][0] = b4.x; bq[h2][g][1] = b4.y; bq[h2][g][2] = b4.z; bq[h2][g][3] = b4.w;
        }
    // every residual unit of the wavefront's 128 x 64 tile is requested BEFORE the first result is formed (32 loads of 8 bytes
    // per lane in flight together: one memory latency per workgroup instead of one per (fragment, channel half) -- the
    // epilogue was 12 % of the 256 -> 256 layer with the loads next to their use)
    constexpr bool RES = (EPI & 1) != 0, F32OUT = (EPI & 2) != 0;
    tpspp_u32x2 rres[RES ? 4 : 1][2][4];
    int pixo[4], nimg[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int tp = 128 * ph + 32 * f + l31;
        const int im = tp / HW;
        pixo[f] = tp - im * HW;
        nimg[f] = n0 + im;
        if constexpr (RES) {
            const bool valid = nimg[f] < P.N;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const size_t bunit = (((size_t)(valid ? nimg[f] : 0) * CGo + ctile * 8 + 4 * h2 + g) * HW + pixo[f]) * 8 + 4 * half;
                    rres[f][h2][g] = *reinterpret_cast<const tpspp_u32x2*>(reinterpret_cast<const unsigned short*>(P.res) + bunit);
                }
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int pix_o = pixo[f], n = nimg[f];
        const bool valid = n < P.N;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            tpspp_u32x2 bpk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
                float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (RES) {
                    const tpspp_u32x2 rb = rres[f][h2][g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = bf16_bits_to_f32((unsigned short)((rb[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[f][h2][4 * g + e] + bq[h2][g][e];
                    if constexpr (RES) { if (P.res_mode == 2) v[e] = v[e] + rv[e]; }
                    if (relu1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    if constexpr (RES) { if (P.res_mode == 1) v[e] = v[e] + rv[e]; }
                }
                if constexpr (F32OUT) {
                    // fp32 NCHW: a half-wavefront's 32 pixels of one channel are 128 contiguous bytes
                    float* ob = reinterpret_cast<float*>(P.out) + ((size_t)(valid ? n : 0) * P.Cout + ctile * BN + 32 * h2 + 8 * g + 4 * half) * HW + pix_o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (valid) ob[(size_t)e * HW] = v[e];
                    continue;
                }
                bpk[g][0] = pack2_bf16(v[0], v[1]); bpk[g][1] = pack2_bf16(v[2], v[3]);
                if (g & 1) {
                    const tpspp_u32x2 d0 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][0], bpk[g][0], false, false);
                    const tpspp_u32x2 d1 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][1], bpk[g][1], false, false);
                    u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                    const int kg = ctile * 8 + 4 * h2 + (g - 1) + half;
                    if (valid)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(P.out) + (((size_t)n * CGo + kg) * HW + pix_o) * 8) = unit;
                }
            }
        }
    }
}

template <int TH, int TW, int NI, int KS = 3>
bool launch_w(const BParams& P, hipStream_t st)
{
    const long groups = (P.N + NI - 1) / NI;
    const long blocks = ((groups + 7) / 8) * 8 * (P.Cout / 128);     // (see the kernel: block -> (image group, channel tile))
    if (blocks > 0x7fffffffL) return false;
    const dim3 grid((unsigned)blocks);
    constexpr int PD = WK<KS>::PD;
#ifdef TPSPP_WIDE_LAB
    if constexpr (KS == 3) {
    if (const char* lv = getenv("TPSPP_WIDE_LAB")) {
        switch (atoi(lv)) {
        case 1: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 1>), grid, dim3(256), 0, st, P); return true;
        case 2: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 2>), grid, dim3(256), 0, st, P); return true;
        case 4: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 4>), grid, dim3(256), 0, st, P); return true;
        case 8: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 8>), grid, dim3(256), 0, st, P); return true;
        case 7: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 7>), grid, dim3(256), 0, st, P); return true;
        case 15: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 15>), grid, dim3(256), 0, st, P); return true;
        case 16: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 16>), grid, dim3(256), 0, st, P); return true;
        case 24: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 24>), grid, dim3(256), 0, st, P); return true;
        case 10: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 5, 10>), grid, dim3(256), 0, st, P); return true;
        case 108: hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, 8, 0>), grid, dim3(256), 0, st, P); return true;   // weight loads 8 taps ahead
        default: break;
        }
    }
    }
#endif
    if (P.out_f32 == 1) {
        if (P.res_mode) hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 3, PD, 0, KS>), grid, dim3(256), 0, st, P);
        else hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 2, PD, 0, KS>), grid, dim3(256), 0, st, P);
        return true;
    }
    if (P.res_mode) hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 1, PD, 0, KS>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((conv3_wide_kernel<TH, TW, NI, 0, PD, 0, KS>), grid, dim3(256), 0, st, P);
    return true;
}

}  // namespace

namespace tpspp {

// true when the wide-tile kernel took the layer: 3x3 stride 1, plain bf16, ONE blocked source at full resolution, Cin a multiple
// of 32, Cout a multiple of 128, blocked bf16 or fp32 NCHW output (+ blocked bf16 residual), bias / ReLU only, 8x32 or 4x16 maps
bool conv3_wide_launch(const BParams& P, hipStream_t st)
{
    if (P.nsrc != 1 || P.src[0].f32 != 2 || P.src[0].lh || P.src[0].lw || (P.out_f32 != 2 && P.out_f32 != 1) || P.post_scale || P.relu > 1) return false;
    if ((P.Cin % 32) || (P.Cout % 128) || P.Cin < 64 || P.src[0].C != P.Cin) return false;
    if (P.res_mode && P.res_f32 != 2) return false;
    if (P.Ho != P.Hi || P.Wo != P.Wi) return false;
    if (P.Ho == 8 && P.Wo == 32) return launch_w<8, 32, 1>(P, st);
    if (P.Ho == 4 && P.Wo == 16) return launch_w<4, 16, 4>(P, st);
    return false;
}

// the same kernel for 1x1 layers (KS = 1: 32-channel chunks, no halo), where it measures faster than the blocked 1x1 kernel
// (tpspp_conv1x1_blk.hip: whole weight in LDS) or that kernel's weight does not fit: Cin >= 256 -- 256 -> 256 44 -> 33 us,
// 256 -> 512 133 (tiled) / 82 (sliced) -> 53 us on 8x32 maps, 512 -> 512 on 4x16 maps 55 (tiled) -> 25 us at batch 512; the
// 128-channel layers measure the same on both and stay where they were.  Stride 1, one blocked source, Cin a multiple of 64,
// Cout a multiple of 128, blocked bf16 or fp32 NCHW output
bool conv1x1_wide_launch(const BParams& P, hipStream_t st)
{
    if (P.nsrc != 1 || P.src[0].f32 != 2 || P.src[0].lh || P.src[0].lw || (P.out_f32 != 2 && P.out_f32 != 1) || P.post_scale || P.relu > 1) return false;
    static const bool all = getenv("TPSPP_C1X1_WIDE_ALL") != nullptr;        // (lab: every qualifying 1x1 layer, not only Cin >= 256)
    if ((P.Cin % 64) || (P.Cout % 128) || P.src[0].C != P.Cin || (P.Cin < 256 && !all)) return false;
    if (P.res_mode && P.res_f32 != 2) return false;
    if (P.Ho != P.Hi || P.Wo != P.Wi) return false;
    if (P.Ho == 8 && P.Wo == 32) return launch_w<8, 32, 1, 1>(P, st);
    if (P.Ho == 4 && P.Wo == 16) return launch_w<4, 16, 4, 1>(P, st);
    return false;
}

}  // namespace tpspp
