// Persistent, software-pipelined 3x3 bf16 convolution for maps in the BLOCKED layout (N, C/8, H, W, 8) -- the
// big TPS++ MSFA layers of BASELINE.json configs[2] (down0_1 / down1_1: 64 -> 64 stride 2 on 32x128; enc0: 192 -> 64,
// dec2 / dec3: 64 -> 64 on 16x64).  Same arithmetic as conv_tiled_bf16_kernel (tpspp_conv_bf16_impl.h: the same
// v_mfma_f32_32x32x16_bf16 products accumulated in the same order, chunk by chunk, tap by tap), so the two kernels
// agree bit for bit; what changes is how operands reach the matrix pipe:
//
//  * a 16-byte unit of a blocked source IS a unit of the channel-innermost LDS patch, so the patch is not staged
//    through registers at all: LOADER wavefronts issue global_load_lds_dwordx4 (LDS-DMA, 64 patch positions per
//    instruction; padding positions point at a 16-byte zero unit in global memory) into a ring of NB (4-7) buffers, one
//    slot of 16 or 32 channels of a tile per buffer; a loader fills, waits (s_waitcnt vmcnt(0)), announces ("landed") and
//    moves on to its next slot, so every buffer that is not being multiplied is in flight, across tile boundaries;
//  * eight MFMA wavefronts in two TEAMS of four (one wavefront of each team per SIMD, 32 x NF pixels x 64 channels
//    per wavefront), each team on its own tile, half a tile out of phase: one team's epilogue (conversion, stores)
//    and flag waits sit under the other team's matrix instructions;
//  * no barrier in the steady state: two LDS counters per buffer ("landed": its fill is complete; "drained": the
//    four wavefronts of the consuming team are done reading it);
//  * workgroups are persistent (one per CU, tile pairs strided by the grid): with Cin <= 64 the whole weight (74 KB)
//    stays in LDS for the life of the workgroup, otherwise the chunk's slab travels with the patch;
//  * stride 2: a patch row is stored [even columns | odd columns] (the DMA lanes fetch in that order), so a
//    fragment's 32 pixels read 32 consecutive units for every tap.
// Measured (batch 512, scripts/debug/bench_conv16.py; tiled kernel -> this one): dec2 (upsampled source + skip) 86 ->
// 49 us, dec3 (fp32 output) 64 -> 49, 64 -> 64 47 -> 41, enc0 (192 -> 64, streamed weight) 124 -> 121, stride 2 77 ->
// 70, dec1 (8x32) 26 -> 18.  DESIGN.md section 4e has what was varied without effect and the phase stamps
// (scripts/debug/trace_conv.py, -DTPSPP_CONV_TRACE): in the steady state the multiply runs at the matrix pipe's rate at
// the ~1.7 GHz the chip holds under this load; the first chunk lands 5-6 us after launch.
//
// The DMA and the flag traffic are inline asm: an LDS-DMA the compiler can see makes it order every later LDS access
// of the wavefront behind s_waitcnt vmcnt(0) (tpspp_warp_pair.h has the same note), which is exactly the overlap
// this kernel exists for.
//
// Replaces (reference, mmocr/models/textrecog/backbones/tps_pp/tps_pp.py): :126-131 (conv3x3_block / MSFA encoder),
// :149-154, :156-169 (decoder convolutions + skip additions), :538-552 (down0_1 / down1_1), when the module runs bf16.
// Bound: HBM for the stride-2 layers (4x the output is read), the matrix pipe for the 64 -> 64 layers (15.5 us at
// the bf16 peak), the LDS-DMA fill path for the streamed-weight layer.
#include "tpspp_conv_bf16_impl.h"

namespace {

constexpr int kPD = 2;                           // taps whose fragments are requested ahead of the products
// loader wavefronts: 8 where the MFMA wavefronts fit 128 registers (one fragment each: the stride-2 layers, which are
// bound by their fill: 73 -> 67 us), else 4
// MFMA wavefronts per team (two teams): the tile's fragments / NF
template <int TH, int TW, int NF> struct PTeam {
    static constexpr int W = TH * TW / (32 * NF);
    static_assert(W == 4, "a workgroup is at most 16 wavefronts: two teams of four and up to eight loaders");
};
template <int NF, int PTW> struct PLoaders { static constexpr int N = (NF == 1 && PTW == 4) ? 8 : 4; };
constexpr int kPKC = 16, kPKG = 2;               // channels / channel groups per chunk (the arranged weight's chunking)
constexpr int kPSlab = 9 * kPKG * BN;            // 16-byte units of a chunk's weight slab
constexpr int kPSlabDma = kPSlab / kWave;        // 18 LDS-DMA instructions
constexpr int kFlagUnits = 4 + 16;               // 64 bytes of counters and the 64 biases in front of the buffers
constexpr int kPLdsMax = 160 * 1024;

__device__ u32x4 g_zero_unit;                    // what padding positions read (zero-initialised device memory)

// -DTPSPP_CONV_TRACE: workgroup 0 records s_memtime at its phase boundaries (256 stamps per wavefront), read back with
// tpspp_debug_conv_trace -- how the pipeline was tuned (scripts/debug/trace_conv.py); compiled out of the product.
#ifdef TPSPP_CONV_TRACE
__device__ long long g_trace[16 * 256];
#define TRACE_INIT() long long* trp_ = g_trace + wv * 256; int tri_ = 0; const bool tr_ = blockIdx.x == 0 && lane == 0
#define STAMP() do { if (tr_ && tri_ < 256) trp_[tri_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE_INIT() do {} while (0)
#define STAMP() do {} while (0)
#endif

// Patch geometry.  Stride 2 (along x): a patch row is stored as [even columns | odd columns], so that the 32 output
// pixels of a fragment read 32 consecutive units for every tap (dense rows would put them 32 bytes apart: 2-way bank
// conflicts on every B fragment); the DMA lanes simply fetch the positions in that order.
template <int SH, int SW, int TH, int TW, bool WRES, int CPB>
struct PCfg {
    static constexpr int PH = (TH - 1) * SH + 3, PW = (TW - 1) * SW + 3;
    static constexpr int PWH = (PW + 1) / 2;               // columns per parity (stride 2)
    static constexpr int PWL = SW == 2 ? 2 * PWH : PW;     // units per patch row in LDS
    static constexpr int PS = PH * PWL;                    // units per channel group
    static constexpr int NPI = (PS + kWave - 1) / kWave;   // DMA instructions per channel group (the last one's window is
                                                           // shifted back so that it ends at PS: no padding in LDS)
    static constexpr int PATCH = CPB * kPKG * PS;          // units of a slot's patch: CPB chunks of 16 channels
    // a buffer: resident weight -- one team's patch; streamed weight -- BOTH teams' patches of the same chunk index and
    // the chunk's slab once (the slab is then fetched per 2 tiles: the 192 -> 64 layer is bound by the bytes through the
    // LDS-DMA path, 0.75 GB per launch at 6.4 TB/s with a slab per tile)
    static constexpr int BUF = WRES ? PATCH : 2 * PATCH + CPB * kPSlab;
    // buffers in the ring: what fits beside the flags / bias and a resident weight of up to 64 input channels
    static constexpr int NBFIT = (kPLdsMax / 16 - kFlagUnits - (WRES ? 4 * kPSlab : 0)) / BUF;
    static constexpr int NB = NBFIT > 7 ? 7 : NBFIT;          // (2 NB + 1 flag words)
    static_assert(NB >= 1, "no room for a ring");
    static_assert(PS >= kWave, "a DMA window is 64 positions");
    static __device__ __forceinline__ constexpr int tap_off(int ky, int kx)
    {
        return ky * PWL + (SW == 2 ? (kx & 1) * PWH + (kx >> 1) : kx);
    }
};

// s_nop 0: a SALU write of M0 needs one wait state before an LDS-DMA instruction reads it (the hazard recogniser does
// not look inside inline asm); m0 is declared clobbered so that the compiler re-materialises its own uses of it.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_byte)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ int lds_peek(const int* p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke(int* p, int v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_bump(int* p)
{
    int one = 1;
    asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(one) : "memory");
}
__device__ __forceinline__ void lds_await(const int* p, int want)
{
    while (lds_peek(p) < want) __builtin_amdgcn_s_sleep(1);
}

// Two teams of kPTW MFMA wavefronts (one wavefront of each team per SIMD; 32 x NF pixels x 64 channels per wavefront), each
// team on its own tile; 4 or 8 loader wavefronts; NB buffers in one ring.
// Ring order ("slots"): team A's chunks and team B's chunks alternate, B lagging half a tile (D = nchunks / 2 chunks), so
// that one team's epilogue and flag waits sit under the other team's matrix instructions:
//     slot q < D: A[q];   slot D + 2 j: A[D + j];   slot D + 2 j + 1: B[j]          (A[k] = team A's k-th chunk)
// Slot q lives in buffer q % NB and is its (q / NB)-th fill; loader q % (number of loaders) fills it.  A slot past its team's last
// chunk is empty: its loader counts it as drained and nobody waits for it.
// EPI: 0 blocked bf16 output; 1 blocked output + blocked residual (res_mode 1 / 2); 2 fp32 NCHW output.
// flags (ints at the start of LDS):  landed[b] = flags[b]: fills of buffer b that are complete;
//                                    drained[b] = flags[NB + b]: (MFMA wavefronts x fills) that are done reading it
//                                    wready = flags[2 NB]: MFMA wavefronts whose part of the resident weight is in LDS
template <int SH, int SW, int TH, int TW, int NF, bool WRES, int EPI, int CPB>
__global__ void __launch_bounds__((2 * PTeam<TH, TW, NF>::W + PLoaders<NF, PTeam<TH, TW, NF>::W>::N) * kWave, 1)
conv3_blk_persist_kernel(const BParams P, int ntx, int nty, int ntiles)
{
    using Cfg = PCfg<SH, SW, TH, TW, WRES, CPB>;
    constexpr int PW = Cfg::PW, PWL = Cfg::PWL, PS = Cfg::PS, NPI = Cfg::NPI, BUF = Cfg::BUF, NB = Cfg::NB;
    constexpr int kPTW = PTeam<TH, TW, NF>::W;
    constexpr int CW = 2 * kPTW;
    static_assert(TH * TW == kPTW * 32 * NF, "tile = kPTW wavefronts x NF fragments of 32 pixels");
    static_assert(TW % 32 == 0, "a fragment is 32 pixels of one row");
    static_assert(2 * NB + 1 <= 16, "flag words");
    extern __shared__ u32x4 sAll[];
    int* const flags = reinterpret_cast<int*>(sAll);
    float* const sBias = reinterpret_cast<float*>(sAll + 4);
    u32x4* const sB = sAll + kFlagUnits;                    // NB buffers
    u32x4* const sWr = sB + NB * BUF;                       // WRES: every chunk's slab

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);      // uniform: buffer addresses live in SGPRs
    const int nchunks = P.nchunks / CPB;                  // slots per tile: CPB chunks of 16 channels each
    const int HoWo = P.Ho * P.Wo;

    TRACE_INIT();
    STAMP();
    if (tid < 2 * NB + 1) flags[tid] = 0;
    if (tid >= kWave && tid < 2 * kWave) sBias[tid - kWave] = (P.bias && tid - kWave < P.Cout) ? P.bias[tid - kWave] : 0.0f;
    __syncthreads();                                        // the only barrier of the kernel: counters are zero
    STAMP();

    // this workgroup's tiles: a CONTIGUOUS run tile0 + s, s = 0 .. my_tiles - 1 (team A takes the even s, team B the odd):
    // consecutive tiles are vertically adjacent pieces of one image, so the halo rows two tiles share are requested by the
    // same CU at about the same time (one of the two requests hits in L1 / this XCD's L2 instead of going to HBM from two
    // XCDs: with tiles strided by the grid, the 8 row tiles of an image sat on 8 different XCDs)
    const int per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = (int)blockIdx.x * per_wg;
    const int my_tiles = tile0 >= ntiles ? 0 : (ntiles - tile0 < per_wg ? ntiles - tile0 : per_wg);
    const int KA = ((my_tiles + 1) >> 1) * nchunks, KB = (my_tiles >> 1) * nchunks;   // chunks per team
    const int D = nchunks >> 1;

    if (wv >= CW) {
        // ======================= loaders =======================
        // fill, s_waitcnt vmcnt(0), announce; before refilling a buffer wait until its last content has been drained: every
        // buffer that is not being multiplied is in flight, across tile boundaries.
        // (s_setprio: a loader shares its SIMD with two MFMA wavefronts)
        __builtin_amdgcn_s_setprio(3);
        const int L = wv - CW;
        const char* const zero = reinterpret_cast<const char*>(&g_zero_unit);
        const int tailA = KA > D ? KA - D : 0;
        const int Q = WRES ? D + 2 * (tailA > KB ? tailA : KB) : KA;         // streamed weight: one slot per chunk of a tile PAIR
        for (int q = L; q < Q; q += PLoaders<NF, kPTW>::N) {
            const int b = q % NB, use = q / NB;
            int team = 0, k = q;
            if constexpr (WRES) {
                if (q >= D) { const int j = (q - D) >> 1; team = (q - D) & 1; k = team ? j : D + j; }
                if (k >= (team ? KB : KA)) {
                    // empty slot: counted as drained -- in its turn (an early count would let an EARLIER fill of this
                    // buffer start before the content before it has been read)
                    if (use > 0) lds_await(flags + NB + b, kPTW * use);
                    if (lane == 0) for (int i = 0; i < kPTW; ++i) lds_bump(flags + NB + b);
                    continue;
                }
            }
            const unsigned dst = (unsigned)(size_t)(sB + b * BUF);
            // streamed weight: a slot is read by both teams, except the slots of an odd count's last pair (one tile)
            int want = kPTW * use;
            if constexpr (!WRES) {
                const int first_single = (my_tiles & 1) ? (my_tiles >> 1) * nchunks : 0x7fffffff;
                const int singles = q >= first_single ? (q - first_single) / NB : 0;     // earlier fills of this buffer among them
                want = CW * use - kPTW * singles;
            }
            if (use > 0) lds_await(flags + NB + b, want);
            for (int tm = 0; tm < (WRES ? 1 : 2); ++tm) {
            if (!WRES) team = tm;
            const int kt = k / nchunks, chunk = k - kt * nchunks;
            if (2 * kt + team >= my_tiles) continue;
            const int tile = tile0 + 2 * kt + team;
            const int n = tile / (ntx * nty), t1 = tile - n * (ntx * nty);
            const int ty = t1 / ntx, tx = t1 - ty * ntx;
            const int iy_base = ty * TH * SH - 1, ix_base = tx * TW * SW - 1;
            for (int c2 = 0; c2 < CPB; ++c2) {
            const int c0 = (chunk * CPB + c2) * kPKC;
            int cbase = 0, s = 0;
            while (c0 >= cbase + P.src[s].C) { cbase += P.src[s].C; ++s; }
            const BSrc cur = P.src[s];
            const int plane = cur.H * cur.W;
            const char* sp = reinterpret_cast<const char*>(reinterpret_cast<const u32x4*>(cur.p) +
                                                           ((size_t)n * (cur.C >> 3) + ((c0 - cbase) >> 3)) * plane);
            const unsigned dstp = dst + (WRES ? 0u : (unsigned)team * (Cfg::PATCH * 16u)) + (unsigned)c2 * (kPKG * PS * 16u);
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                constexpr int kLast = PS - kWave;
                const int start = i * kWave < kLast ? i * kWave : kLast;
                const int e = start + lane;
                const int py = e / PWL, r = e - py * PWL;
                int px = r;
                bool ok = true;
                if constexpr (SW == 2) {
                    const int par = r >= Cfg::PWH ? 1 : 0, c = r - par * Cfg::PWH;
                    px = 2 * c + par;
                    ok = px < PW;
                }
                const int iy = iy_base + py, ix = ix_base + px;
                ok = ok && iy >= 0 && iy < P.Hi && ix >= 0 && ix < P.Wi;
                const unsigned off = (unsigned)((iy >> cur.lh) * cur.W + (ix >> cur.lw)) * 16u;
#pragma unroll
                for (int g = 0; g < kPKG; ++g) {
                    const char* src = ok ? sp + (size_t)g * plane * 16 + off : zero;
                    dma16(src, dstp + (unsigned)(g * PS + start) * 16u);
                }
            }
            }   // c2
            }   // team
            if constexpr (!WRES) {
                const int chunk = q % nchunks;
                const char* wp = reinterpret_cast<const char*>(P.wt + (size_t)chunk * CPB * kPSlab) + lane * 16;
#pragma unroll
                for (int i = 0; i < CPB * kPSlabDma; ++i) dma16(wp + i * 1024, dst + (unsigned)(2 * Cfg::PATCH * 16 + i * 1024));
            }
            STAMP();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP();
            if (lane == 0) lds_poke(flags + b, use + 1);
        }
        return;
    }

    // ======================= MFMA wavefronts =======================
    if constexpr (WRES) {
        // the resident weight: copied by the MFMA wavefronts while the loaders already fetch the first patches
        constexpr int T = CW * kWave, U = 3;
        const int total = P.nchunks * kPSlab;
        for (int base = 0; base < total; base += U * T) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int e = base + u * T + tid; v[u] = P.wt[e < total ? e : 0]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { const int e = base + u * T + tid; if (e < total) sWr[e] = v[u]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_bump(flags + 2 * NB);
    }
    const int team = wv / kPTW, w = wv - team * kPTW;
    const int half = lane >> 5, l31 = lane & 31;
    int fty[NF], ftx[NF], fpos[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int tp = (w * NF + f) * 32 + l31;
        fty[f] = tp / TW;
        ftx[f] = tp - fty[f] * TW;
        fpos[f] = half * PS + fty[f] * SH * PWL + ftx[f];    // (stride 2: the column's slot inside its parity half)
    }
    const bool relu1 = P.relu == 1;

    int k = 0;                                               // this team's chunk counter
    for (int tseq = team; tseq < my_tiles; tseq += 2) {
        const int tile = tile0 + tseq;
        const int n = tile / (ntx * nty), t1 = tile - n * (ntx * nty);
        const int ty = t1 / ntx, tx = t1 - ty * ntx;
        const int oy0 = ty * TH, ox0 = tx * TW;
        f32x16 acc[NF][2];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[f][h2][i] = 0.0f;

        for (int chunk = 0; chunk < nchunks; ++chunk, ++k) {
            // resident weight: the interleaved ring order; streamed weight: slot k holds chunk k of both teams' tiles
            const int q = !WRES ? k : (team ? D + 2 * k + 1 : (k < D ? k : D + 2 * (k - D)));
            const int b = q % NB;
            STAMP();
            lds_await(flags + b, q / NB + 1);
            if (WRES && k == 0) lds_await(flags + 2 * NB, CW);
            STAMP();
            for (int c2 = 0; c2 < CPB; ++c2) {
            const u32x4* const pb = sB + b * BUF + (WRES ? 0 : team * Cfg::PATCH) + c2 * (kPKG * PS);
            const u32x4* const wb = WRES ? sWr + (chunk * CPB + c2) * kPSlab : sB + b * BUF + 2 * Cfg::PATCH + c2 * kPSlab;
            // register double buffer over the taps: tap t + 1's fragments are requested before tap t's products are
            // issued; the scheduling barriers keep that order (without them the reads sink below the products, or all of
            // them are hoisted to the top)
            bf16x8 fa[kPD + 1][2], fb[kPD + 1][NF];
            auto fetch = [&](int tap, int slot) {
                const int ky = tap / 3, kx = tap - ky * 3;
                fa[slot][0] = __builtin_bit_cast(bf16x8, wb[(tap * kPKG + half) * BN + l31]);
                fa[slot][1] = __builtin_bit_cast(bf16x8, wb[(tap * kPKG + half) * BN + 32 + l31]);
#pragma unroll
                for (int f = 0; f < NF; ++f) fb[slot][f] = __builtin_bit_cast(bf16x8, pb[fpos[f] + Cfg::tap_off(ky, kx)]);
            };
#pragma unroll
            for (int t0 = 0; t0 < kPD; ++t0) fetch(t0, t0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + kPD < 9) fetch(tap + kPD, (tap + kPD) % (kPD + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap % (kPD + 1)][0], fb[tap % (kPD + 1)][f], acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap % (kPD + 1)][1], fb[tap % (kPD + 1)][f], acc[f][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }   // c2
            // this wavefront's LDS reads were issued before the counter update and LDS serves a wavefront in order
            asm volatile("" ::: "memory");
            if (lane == 0) lds_bump(flags + NB + b);
        }

        // ---- epilogue: bias, residual, ReLU; blocked bf16 (16-byte units) or fp32 NCHW ----
        // (round 6: Cout = 32 -- the backbone's first stage -- runs here as well: the arranged weight's cout tile is zero-padded to
        // 64, the upper half of the products is discarded, the tensors have CGo = Cout / 8 channel groups)
        const int CGo = P.Cout >> 3;
        STAMP();
        float bq[2][4][4];                                   // re-read per tile from LDS: 32 registers the multiply keeps free
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + 32 * h2 + 8 * g + 4 * half);
                bq[h2][g][0] = b4.x; bq[h2][g][1] = b4.y; bq[h2][g][2] = b4.z; bq[h2][g][3] = b4.w;
            }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int pix_o = (oy0 + fty[f]) * P.Wo + ox0 + ftx[f];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                tpspp_u32x2 bpk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    const int cu = 32 * h2 + 8 * g;              // + 4 * half + e
                    float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (EPI == 1) {
                        const int cgr = (cu >> 3) < CGo ? (cu >> 3) : 0;      // (a channel group the tensor does not have: any valid unit)
                        const size_t bunit = (((size_t)n * CGo + cgr) * HoWo + pix_o) * 8 + 4 * half;
                        const tpspp_u32x2 rb = *reinterpret_cast<const tpspp_u32x2*>(
                            reinterpret_cast<const unsigned short*>(P.res) + bunit);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            rv[e] = bf16_bits_to_f32((unsigned short)((rb[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                    }
                    if constexpr (EPI == 0) {
                        // a SIMD's time is the sum of its wavefronts' instructions (scripts/ubench/coissue_bench.hip): the
                        // bias goes on with packed adds and the ReLU is taken AFTER the rounding, on the packed pair (a
                        // negative bf16 is a negative int16, so max(., 0) is the same ReLU; -0 -> +0 either way): 6
                        // instructions per 4 results instead of 10, identical bits
                        typedef short s16x2 __attribute__((ext_vector_type(2)));
                        f32x2 lo, hi, blo, bhi;
                        lo[0] = acc[f][h2][4 * g]; lo[1] = acc[f][h2][4 * g + 1]; hi[0] = acc[f][h2][4 * g + 2]; hi[1] = acc[f][h2][4 * g + 3];
                        blo[0] = bq[h2][g][0]; blo[1] = bq[h2][g][1]; bhi[0] = bq[h2][g][2]; bhi[1] = bq[h2][g][3];
                        lo = lo + blo; hi = hi + bhi;
                        unsigned p0 = pack2_bf16(lo[0], lo[1]), p1 = pack2_bf16(hi[0], hi[1]);
                        if (relu1) {
                            const s16x2 z = {0, 0};
                            p0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p0), z));
                            p1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p1), z));
                        }
                        bpk[g][0] = p0; bpk[g][1] = p1;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[f][h2][4 * g + e] + bq[h2][g][e];
                            if constexpr (EPI == 1) { if (P.res_mode == 2) v[e] = v[e] + rv[e]; }
                            if (relu1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                            if constexpr (EPI == 1) { if (P.res_mode == 1) v[e] = v[e] + rv[e]; }
                        }
                        if constexpr (EPI == 1) { bpk[g][0] = pack2_bf16(v[0], v[1]); bpk[g][1] = pack2_bf16(v[2], v[3]); }
                    }
                    if constexpr (EPI != 2) {
                        // the two half-wavefronts hold the two halves of a 16-byte unit: v_permlane32_swap pairs them up
                        if (g & 1) {
                            const tpspp_u32x2 d0 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][0], bpk[g][0], false, false);
                            const tpspp_u32x2 d1 = __builtin_amdgcn_permlane32_swap(bpk[g - 1][1], bpk[g][1], false, false);
                            u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                            const int kg = 4 * h2 + (g - 1) + half;
                            if (kg < CGo)
                                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(P.out) +
                                                          (((size_t)n * CGo + kg) * HoWo + pix_o) * 8) = unit;
                        }
                    } else {
                        float* ob = reinterpret_cast<float*>(P.out) + ((size_t)n * P.Cout + cu + 4 * half) * HoWo + pix_o;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cu + 4 * half + e < P.Cout) ob[(size_t)e * HoWo] = v[e];
                    }
                }
            }
        }
        STAMP();
    }
}

template <typename K>
bool launch_k(K kfn, bool (&attr_done)[tpspp::kMaxDevices], int grid, int threads, size_t lds, hipStream_t st, const BParams& P,
              int ntx, int nty, int nt)
{
    if (tpspp::first_use_on_device(attr_done)) {             // per device: the opt-in is a property of the loaded code object
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(threads), lds, st, P, ntx, nty, nt);
    return true;
}

template <int SH, int SW, int TH, int TW, int NF, bool WRES, int EPI, int CPB>
bool launch_pc(const BParams& P, hipStream_t st)
{
    using Cfg = PCfg<SH, SW, TH, TW, WRES, CPB>;
    const int ntx = P.Wo / TW, nty = P.Ho / TH;
    const long nt = (long)P.N * ntx * nty;
    if (nt > 0x3fffffffL) return false;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        return false;
    }
    const long pairs = (nt + 1) / 2;                         // a workgroup works on two tiles at a time
    const int grid = (int)(pairs < ncu ? pairs : ncu);
    const size_t lds = (size_t)(kFlagUnits + Cfg::NB * Cfg::BUF + (WRES ? P.nchunks * kPSlab : 0)) * 16;   // (BUF: see PCfg)
    if (lds > (size_t)kPLdsMax) return false;
    static bool attr[tpspp::kMaxDevices] = {};
    return launch_k(conv3_blk_persist_kernel<SH, SW, TH, TW, NF, WRES, EPI, CPB>, attr, grid,
                    (2 * PTeam<TH, TW, NF>::W + PLoaders<NF, PTeam<TH, TW, NF>::W>::N) * kWave, lds, st, P, ntx, nty, (int)nt);
}

// resident weight and an even chunk count: two chunks (32 channels) per ring slot -- half the flag hand-offs, 72 matrix
// instructions per wait
template <int SH, int SW, int TH, int TW, int NF, bool WRES, int EPI>
bool launch_pe(const BParams& P, hipStream_t st)
{
    if constexpr (WRES && PCfg<SH, SW, TH, TW, true, 2>::NBFIT >= 3) {
        if (P.nchunks % 2 == 0) return launch_pc<SH, SW, TH, TW, NF, WRES, EPI, 2>(P, st);
    }
    return launch_pc<SH, SW, TH, TW, NF, WRES, EPI, 1>(P, st);
}

template <int SH, int SW, int TH, int TW, int NF>
bool launch_p(const BParams& P, hipStream_t st)
{
    if (P.Ho % TH || P.Wo % TW) return false;
    const int epi = P.out_f32 == 1 ? (P.res_mode ? -1 : 2) : (P.res_mode ? (P.res_f32 == 2 ? 1 : -1) : 0);
    const bool wres = P.Cin <= 64;
    if (epi == 0) return wres ? launch_pe<SH, SW, TH, TW, NF, true, 0>(P, st) : launch_pe<SH, SW, TH, TW, NF, false, 0>(P, st);
    if (epi == 1) return wres ? launch_pe<SH, SW, TH, TW, NF, true, 1>(P, st) : launch_pe<SH, SW, TH, TW, NF, false, 1>(P, st);
    if (epi == 2) return wres ? launch_pe<SH, SW, TH, TW, NF, true, 2>(P, st) : launch_pe<SH, SW, TH, TW, NF, false, 2>(P, st);
    return false;
}

}  // namespace

#ifdef TPSPP_CONV_TRACE
extern "C" __attribute__((visibility("default"))) int tpspp_debug_conv_trace(long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(long long) * n);
}
#endif

namespace tpspp {

// true when the persistent kernel took the layer: 3x3, plain bf16, every source blocked, 64 (or 32) output channels, blocked
// output (+ blocked residual) or fp32-NCHW output, bias / ReLU only, whole tiles
bool conv_bf16_persist_launch(const BParams& P, int sh, int sw, hipStream_t st)
{
    if ((P.Cout != 64 && P.Cout != 32) || (P.Cin % kPKC) || P.post_scale || P.relu > 1) return false;
    if (P.out_f32 != 2 && P.out_f32 != 1) return false;
    for (int i = 0; i < P.nsrc; ++i)
        if (P.src[i].f32 != 2 || (P.src[i].C % kPKC)) return false;
    // 64-wide maps: 4 x 64 (stride 2: 2 x 64) tiles; 32-wide maps (the 8x32 level of the MSFA): 8 x 32 (4 x 32)
    if (sh == 1 && sw == 1) return P.Wo % 64 == 0 ? launch_p<1, 1, 4, 64, 2>(P, st) : launch_p<1, 1, 8, 32, 2>(P, st);
    if (sh == 2 && sw == 2) return P.Wo % 64 == 0 ? launch_p<2, 2, 2, 64, 1>(P, st) : launch_p<2, 2, 4, 32, 1>(P, st);
    return false;
}

}  // namespace tpspp
