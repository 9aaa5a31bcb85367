// Classic-geometry warp for LARGE images (round 5): row bands with span staging.
//
// Replaces, bit for bit: preprocessor/tps_preprocessor.py:71-83 + 270-282 (GridGenerator.build_P_prime: two bmm,
// then F.grid_sample bilinear / border / align_corners) for a mirror-symmetric RBF table and img_size ==
// rectified_img_size, for the geometries whose quadrant needs more than one workgroup (64x200, 64x256, 96x128 ...;
// the reference takes any img_size, tps_preprocessor.py:39-58, tests/test_models/test_recog_config.py:103-157) and
// for images that do not fit the LDS at all.
//
// What tpspp_warp_geo.h does for such an image: `bands` workgroups per image, each staging the WHOLE image (taps may
// fall anywhere) -- bands x the algorithmic bytes through the LDS-DMA path, 150 KB of LDS = one workgroup per CU, load
// -> sample -> store strictly one after the other on every CU (64x200: 0.36 of the HBM peak).  Here a workgroup still
// owns a band of output rows (row groups [band RG / bands, ...) of the upper half and their mirror rows of the lower
// half: the table's 4-fold symmetry is kept), but
//   * the sampling grid of the band is expanded FIRST (T solve, packed table, FMA chains, tap descriptors: nothing of
//     that needs the image);
//   * the rows its taps actually reach -- [min y0, max y0 + 1] over the band's pixels, found with LDS min / max atomics
//     -- are the only ones staged, one REGION (upper rows, then lower rows) at a time in the same buffer, by LDS-DMA
//     (1 KB per wavefront instruction, nt);
//   * a band whose span does not fit the buffer (a transformation that folds half the image into a few rows) takes its
//     taps from global memory instead: same values, same arithmetic, slower -- the choice is per workgroup and uniform;
//   * results are staged in the same buffer in the output's layout and leave as 16-byte nt stores.
// LDS per workgroup is ~38 KB (four workgroups per CU) instead of the whole image, so one workgroup's DMA runs under
// the other's arithmetic and stores; bands of one image are dispatched to the same XCD (block -> XCD is round-robin:
// MI355X_MICROARCH.md), whose L2 then serves the rows neighbouring bands share.
// One quadrant pixel (4 mirror pixels) per thread: its own packed copy of the table (QP = 1 layout, third section of
// the prepared table).  The arithmetic is unchanged: T rows and grid coordinates are the k-ascending fp32 FMA chains
// from zero, taps as in make_taps_lite().  Compiled with -ffp-contract=off.
#pragma once
#include "tpspp_warp_img.h"

namespace tpspp_span {

using namespace tpspp_dev;
using tpspp_pair::gchar;
using tpspp_pair::make_taps_lite;
using tpspp_pair::perm_x;
using tpspp_pair::perm_y;
using tpspp_pair::store16_nt;
using tpspp_pair::TapsLite;
using tpspp_pair::v4f;
using tpspp_pair::v4f_a4;
using tpspp_pair::wait_flag_lds;

struct SpanParams {
    const float* in; const float* ctrl; const float* inv_delta_c;
    const float* packed;   // pack_img_table_kernel layout with QP = 1 for (BW, CG): thread order of the WHOLE quadrant
    int N;
    float* out; float* grid; int32_t* idx;
    int H, W;              // input = output size
    int BW, CG, RG;        // pixel block width (BH = 32 / BW rows), column groups per half-row, row groups of the quadrant
    int lg_bw;             // log2(BW)
    int bands;             // workgroups per image; each owns RG / bands row groups (and their mirror rows)
    int nthr;              // live threads per workgroup = CG (RG / bands) 32
    int span_rows;         // rows of ONE region the staging buffer holds per channel
    int chunk_floats;      // LDS floats per channel of the staging buffer (>= span_rows W, whole 1-KB DMA pieces)
    int stage_off;         // float offset of the staging buffer in LDS
    int force_gather;      // lab knob: every workgroup takes the global-memory path
    int margin;            // SPEC: rows staged above / below the band's own rows (both regions at once, at launch)
};

// LDS floats in front of the staging buffer: T (2 K, padded to 4) + 8 words (flag, span min / max of both regions)
__host__ __device__ constexpr int span_stage_off(int K) { return ((2 * K + 3) & ~3) + 8; }
inline int span_chunk_floats(int span_rows, int W) { return ((span_rows * W * 4 + 1023) / 1024) * 256; }
inline size_t span_lds_bytes(int K, int C, int W, int span_rows, int out_rows)
{
    // the buffer holds a region's span (C chunks) and, later, the band's results (2 x out_rows rows of every channel)
    size_t stage = (size_t)C * span_chunk_floats(span_rows, W);
    const size_t outb = (size_t)2 * C * out_rows * W;
    if (outb > stage) stage = outb;
    return (size_t)(span_stage_off(K) + stage + W + 4) * 4;    // + what out-of-image taps of the last row may read
}

// wavefront-wide min / max of an int as a SCALAR: four DPP steps inside the rows of 16 lanes (quad swaps, half-row and row
// mirrors), then the four rows' results through v_readlane -- no LDS traffic (a __shfl_xor butterfly is 6 ds_bpermute).
template <bool MAX>
__device__ __forceinline__ int wave_reduce(int v)
{
    auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return op(op(a, b), op(c, d));
}

// SPEC (round 5, late): the span search serialises a workgroup -- table -> chains -> taps -> min / max -> barrier -> DMA ->
// barrier -> reads, twice -- and with four workgroups per CU the counters still show the SIMDs issuing 62 % of the time.  A
// rectifier's warp is mild: a band's taps land within a few rows of the band's own rows.  So both regions' WINDOWS -- the
// band's rows +- `margin` -- are requested by LDS-DMA as the workgroup's FIRST instructions, in flight under the table
// loads, the T solve and the chains; no span search, no atomics, three barriers instead of seven.  A wavefront in which any
// lane's tap rows leave the window takes THAT mirror pixel's taps from global memory (same values, per wavefront and
// uniform): exact for any warp, fast for the ones that occur.
__device__ __forceinline__ void span_dma16(const void* g, unsigned lds_byte)
{
    // (inline asm: an LDS-DMA the compiler can see makes it order every later LDS access of the wavefront behind
    // s_waitcnt vmcnt(0); s_nop: the M0 write needs a wait state the hazard recogniser does not see inside asm)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(lds_byte), "v"(g) : "memory", "m0");
}

// Geometry policy (round 6): SpanRT takes every size from the launch arguments; SpanFix<...> makes them compile-time constants for
// the geometries that are instantiated (tpspp_warp_geo.hip: 64x200, 48x160, 64x256 -- row pitch, plane size, block shape, band
// decomposition and window margin fold into immediates and shifts: the kernel is bound by vector-ALU issue).
struct SpanRT { static constexpr bool fixed = false; static constexpr int H = 0, W = 0, BW = 0, CG = 0, RG = 0, BANDS = 0, NTHR = 0, MARGIN = 0; };
template <int H_, int W_, int BW_, int CG_, int RG_, int BANDS_, int MARGIN_>
struct SpanFix {
    static constexpr bool fixed = true;
    static constexpr int H = H_, W = W_, BW = BW_, CG = CG_, RG = RG_, BANDS = BANDS_, MARGIN = MARGIN_;
    static constexpr int NTHR = CG_ * (RG_ / BANDS_) * 32;
};

template <int F, int C, bool AUX, bool SPEC = false, typename G = SpanRT>
__global__ void __launch_bounds__(1024, (AUX || C > 3) ? 4 : 7)        // <= 72 registers: two 13-wavefront workgroups per CU
tps_warp_span_kernel(const SpanParams P)
{
    constexpr int K = F + 3;
    const int H = G::fixed ? G::H : P.H, W = G::fixed ? G::W : P.W, HW = H * W;
    const int BW = G::fixed ? G::BW : P.BW, BH = 32 / BW, CG = G::fixed ? G::CG : P.CG;
    const int gRG = G::fixed ? G::RG : P.RG, gBands = G::fixed ? G::BANDS : P.bands, gNthr = G::fixed ? G::NTHR : P.nthr;
    const int gMargin = G::fixed ? G::MARGIN : P.margin;
    const int gLgBw = G::fixed ? (G::BW == 32 ? 5 : G::BW == 16 ? 4 : G::BW == 8 ? 3 : 2) : P.lg_bw;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* sT = reinterpret_cast<float2*>(smem);           // [K]
    int* sWord = reinterpret_cast<int*>(smem + ((2 * K + 3) & ~3));   // [0] T published, [2 + r] min y, [4 + r] max y of region r
    float* sStage = smem + span_stage_off(K);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int NW = (int)(blockDim.x / kWave);
    // block -> (image, band): the bands of an image sit 8 blocks apart, i.e. on one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / gBands) * 8 + xcd, band = slot % gBands;
    if (b >= P.N) return;

    const unsigned row_bytes = (unsigned)HW * 4u;            // one plane
    const unsigned chunk_bytes = (unsigned)P.chunk_floats * 4u;
    int w0[2] = {0, 0}, wlast[2] = {0, 0};                   // SPEC: first / last staged row of each region's window
    int wbytes0 = 0;                                         // SPEC: bytes of region 0's window = LDS offset of region 1's
    if constexpr (SPEC) {
        const int rows_ = (gRG / gBands) * BH, ra_ = band * rows_;
        w0[0] = ra_ - gMargin > 0 ? ra_ - gMargin : 0;
        wlast[0] = (ra_ + rows_ + gMargin < H ? ra_ + rows_ + gMargin : H) - 1;
        w0[1] = H - ra_ - rows_ - gMargin > 0 ? H - ra_ - rows_ - gMargin : 0;
        wlast[1] = (H - ra_ + gMargin < H ? H - ra_ + gMargin : H) - 1;
        wbytes0 = C * (wlast[0] - w0[0] + 1) * W * 4;
        // A region's window in LDS: the C channels' rows back to back, EXACTLY wn W 4 bytes each (no rounding to whole 1-KB
        // pieces: with it 64x200 needs 43 KB per workgroup, three per CU; without, 39.4 KB, four).  A 1-KB piece may then
        // straddle two channels (a lane past the first one's end takes the next channel's bytes) and the last piece of a
        // region is partial: its lanes past the end are masked off -- LDS-DMA honours EXEC, a masked lane writes nothing.
        const char* img0 = reinterpret_cast<const char*>(P.in) + (size_t)b * C * row_bytes;
        const int NW_ = (int)(blockDim.x / kWave);
        int gpiece = 0;                                      // pieces so far (region 0's count: region 1 continues the round-robin)
#pragma unroll
        for (int reg = 0; reg < 2; ++reg) {
            const int nb = (wlast[reg] - w0[reg] + 1) * W * 4;               // bytes per channel (a multiple of 16)
            const int RB = C * nb;
            const int pieces = (RB + 1023) >> 10;
            const char* src0 = img0 + (size_t)w0[reg] * W * 4;
            const unsigned dst0 = (unsigned)(size_t)sStage + (reg ? (unsigned)wbytes0 : 0u);
            // (wavefront 0 issues none when there are others: it solves T, on which every wavefront waits, and memory
            // returns a wavefront's requests in order -- its control points would queue behind its pieces)
            const int ND = NW_ > 1 ? NW_ - 1 : 1, wd = NW_ > 1 ? wv - 1 : 0;
            for (int k = wd < 0 ? pieces : (wd + ND - gpiece % ND) % ND; k < pieces; k += ND) {
                const int p0 = k * 1024;
                const int ch0 = (p0 >= nb ? 1 : 0) + (p0 >= 2 * nb ? 1 : 0) + (p0 >= 3 * nb ? 1 : 0);   // (uniform: scalar compares)
                int off = p0 - ch0 * nb + lane * 16;
                int ch = ch0;
                if (off >= nb) { off -= nb; ++ch; }          // (nb >= 1 KB - 16 is not required: a second wrap cannot occur for nb >= 1008)
                if (p0 + lane * 16 < RB) span_dma16(src0 + (size_t)ch * row_bytes + off, dst0 + (unsigned)p0);
            }
            gpiece += pieces;
        }
    }

    // T-solve inputs first (wavefront 0: lane i keeps control point i and row i of inv_delta_C, 16 bytes at a time;
    // the last piece starts at column K - 4 so that the last row does not read past the matrix)
    constexpr int KGI = (K + 3) / 4;
    float hrowv[KGI * 4];
    float cx = 0.0f, cy = 0.0f;
    if (wv == 0) {
        if (lane < F) {
            const float2 cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[lane];
            cx = cc.x; cy = cc.y;
        }
        const float* row = P.inv_delta_c + (lane < K ? lane : K - 1) * K;
#pragma unroll
        for (int j = 0; j < KGI; ++j) {
            const int c0 = (j == KGI - 1) ? K - 4 : 4 * j;
            const v4f_a4 x = *reinterpret_cast<const v4f_a4*>(row + c0);
            hrowv[4 * j] = x[0]; hrowv[4 * j + 1] = x[1]; hrowv[4 * j + 2] = x[2]; hrowv[4 * j + 3] = x[3];
        }
    }
    if (tid < 8) sWord[tid] = (tid == 2 || tid == 3) ? 0x7fffffff : (tid >= 4 && tid < 6 ? -1 : 0);

    // thread -> its quadrant pixel: half-wavefront hw = block of BW columns x BH rows; row group rg (global over bands).
    // hw / CG on the scalar unit for the wavefront's first half, the second half is the next block (no vector division)
    const int nhw = gNthr >> 5;                             // live half-wavefronts (the last wavefront may be half empty)
    const int l5 = lane & 31;
    const int hw0 = 2 * wv < nhw ? 2 * wv : nhw - 1;
    const int rg0 = hw0 / CG, cg0 = hw0 - rg0 * CG;          // (uniform)
    const bool wrap = cg0 + 1 == CG;
    const bool second = lane >= 32 && 2 * wv + 1 < nhw;      // a dead second half repeats the first half's pixels
    const bool live = lane < 32 ? 2 * wv < nhw : 2 * wv + 1 < nhw;
    const int rg_l = second ? (wrap ? rg0 + 1 : rg0) : rg0;
    const int cg = second ? (wrap ? 0 : cg0 + 1) : cg0;
    const int rgpb = gRG / gBands;                         // row groups per band
    const int rg = band * rgpb + rg_l;
    const int c = cg * BW + (l5 & (BW - 1));                 // BW is a power of two; c < W (CG BW <= W)
    const int r = rg * BH + (l5 >> gLgBw);
    // (column groups may reach past the centre: such a lane's pixels are other lanes' mirror pixels, computed twice
    // with the same bits -- the table is mirror-symmetric -- and written twice with the same value)
    const int rows = rgpb * BH;                              // output rows of this band per region
    const int ra = band * rows;                              // upper region: rows [ra, ra + rows); lower: [H - ra - rows, H - ra)

    // packed table: [wavefront][KG][lane] x 16 bytes in the thread order of the whole quadrant
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const int t_glob = (rg * CG + cg) * 32 + l5;
        const v4f* pk = reinterpret_cast<const v4f*>(P.packed) + (size_t)(t_glob >> 6) * KG * kWave + (t_glob & (kWave - 1));
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const v4f x = pk[g * kWave];
            v[4 * g] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
        }
    }
    lds_only_barrier();                                      // the words are initialised
    if (wv == 0) {
        float ax = 0.0f, ay = 0.0f;
        static_for<K>([&](auto qc) {                         // ordered broadcast: the sum is the reference's FMA chain
            constexpr int q = decltype(qc)::value;
            constexpr int idx = (q / 4 < KGI - 1) ? q : 4 * (KGI - 1) + (q - (K - 4));
            ax = fmaf(hrowv[idx], readlane_f(cx, q), ax);
            ay = fmaf(hrowv[idx], readlane_f(cy, q), ay);
        });
        if (lane < K) sT[lane] = make_float2(ax, ay);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(sWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wait_flag_lds(reinterpret_cast<const float*>(sWord), 1);
    asm volatile("" ::"v"(v[KG * 4 - 1]));                   // (keeps the table's padding register from being recycled early)

    // ---- 8 FMA chains: 4 mirror pixels x (x, y), each k-ascending from zero ----
    float gx[4], gy[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m] = gy[m] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float2 t = sT[q];
        float val[4];
        if constexpr (q == 0) {
            val[0] = val[1] = val[2] = val[3] = v[0];
        } else if constexpr (q == 1) {                        // P.x flips under the x-mirror
            val[0] = v[1]; val[1] = -v[1]; val[2] = v[1]; val[3] = -v[1];
        } else if constexpr (q == 2) {                        // P.y flips under the y-mirror
            val[0] = v[2]; val[1] = v[2]; val[2] = -v[2]; val[3] = -v[2];
        } else {
            constexpr int k = q - 3;
            val[0] = v[3 + k];
            val[1] = v[3 + perm_x<F>(k)];
            val[2] = v[3 + perm_y<F>(k)];
            val[3] = v[3 + perm_x<F>(perm_y<F>(k))];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            gx[m] = fmaf(val[m], t.x, gx[m]);
            gy[m] = fmaf(val[m], t.y, gy[m]);
        }
    });

    // ---- taps: x0, y0 (packed), two fractions, two flags per mirror pixel; the regions' row spans ----
    int ty[4], tx[4];
    float tf[4][2];
    unsigned oob = 0;                                        // bit 2 m: east column outside, bit 2 m + 1: south row outside
    int ylo[2] = {0x7fffffff, 0x7fffffff}, yhi[2] = {-1, -1};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const TapsLite t = make_taps_lite(gx[m], gy[m], H, W);
        if constexpr (AUX) {
            const int rr = (m & 2) ? H - 1 - r : r, cc = (m & 1) ? W - 1 - c : c;
            const unsigned po = 4u * (unsigned)(rr * W + cc);
            if (P.grid && live)
                *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * po) = make_float2(gx[m], gy[m]);
            if (P.idx && live)
                *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * po) = make_int2(t.x0, t.y0);
        }
        tx[m] = t.x0; ty[m] = t.y0;
        tf[m][0] = t.wx; tf[m][1] = t.wy;
        oob |= (t.inx ? 0u : 1u) << (2 * m);
        oob |= (t.iny ? 0u : 2u) << (2 * m);
        const int reg = m >> 1;
        const int y1 = t.iny ? t.y0 + 1 : t.y0;
        ylo[reg] = t.y0 < ylo[reg] ? t.y0 : ylo[reg];
        yhi[reg] = y1 > yhi[reg] ? y1 : yhi[reg];
    }
    int y0r[2] = {0, 0}, y1r[2] = {0, 0};
    bool staged = true;
    if constexpr (!SPEC) {
#pragma unroll
        for (int reg = 0; reg < 2; ++reg) {
            const int lo = wave_reduce<false>(ylo[reg]), hi = wave_reduce<true>(yhi[reg]);
            if (lane == 0) {
                __hip_atomic_fetch_min(sWord + 2 + reg, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(sWord + 4 + reg, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        lds_only_barrier();                                  // both spans known to everybody
        typedef __attribute__((address_space(3))) const volatile int lds_cvint;
        lds_cvint* sw = (lds_cvint*)(size_t)(unsigned)(size_t)sWord;
        y0r[0] = __builtin_amdgcn_readfirstlane(sw[2]); y0r[1] = __builtin_amdgcn_readfirstlane(sw[3]);
        y1r[0] = __builtin_amdgcn_readfirstlane(sw[4]); y1r[1] = __builtin_amdgcn_readfirstlane(sw[5]);
        staged = !P.force_gather && (y1r[0] - y0r[0] + 1) <= P.span_rows && (y1r[1] - y0r[1] + 1) <= P.span_rows;
    }

    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const char* img = reinterpret_cast<const char*>(P.in) + (size_t)b * C * row_bytes;
    float res[4][C];

    // the region's span of every channel -> LDS: contiguous rows [y0r, y1r] of each plane, 1 KB per wavefront instruction;
    // lanes past the span's end re-read its first bytes into the chunk's own tail
    auto stage_region = [&](int reg) {
        const int nbytes = (y1r[reg] - y0r[reg] + 1) * W * 4;
        const int pieces = (nbytes + 1023) >> 10;
        const char* src0 = img + (size_t)y0r[reg] * W * 4;
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            // (wavefront w starts channel ch at piece (w + ch) % NW-ish offset so that the channels' few pieces spread over all)
            for (int k = (wv + (NW - (ch * pieces) % NW)) % NW; k < pieces; k += NW) {
                int off = k * 1024 + lane * 16;
                if (off >= nbytes) off = 0;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src0 + (size_t)ch * row_bytes + off),
                    (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sStage) + ch * chunk_bytes + k * 1024),
                    16, 0, 2 /* nt */);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const bool any_oob = __builtin_amdgcn_ballot_w64(oob != 0u) != 0;    // (rare: a pixel on the image's east / south edge)
    auto combine = [&](int m, const float (&tv)[C][4], bool fast) {
        const float w = tf[m][0], nn = tf[m][1];
        const float e = 1.0f - w, s = 1.0f - nn;
        const float nw = s * e, ne = s * w, sw_ = nn * e, se = nn * w;
        auto go = [&](auto oobc) {
            constexpr bool OOB = decltype(oobc)::value;
            const unsigned fl = oob >> (2 * m);
            const bool inx = !(fl & 1u), iny = !(fl & 2u), inxy = !(fl & 3u);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float v01 = (!OOB || inx) ? tv[ch][1] : 0.0f;
                const float v10 = (!OOB || iny) ? tv[ch][2] : 0.0f;
                const float v11 = (!OOB || inxy) ? tv[ch][3] : 0.0f;
                float acc = tv[ch][0] * nw;
                acc = fmaf(v01, ne, acc);
                acc = fmaf(v10, sw_, acc);
                acc = fmaf(v11, se, acc);
                res[m][ch] = acc;
            }
        };
        if (fast) go(std::false_type{}); else go(std::true_type{});
    };

    if constexpr (SPEC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's pieces of both windows have landed
        __builtin_amdgcn_s_barrier();                        // ... and everybody's
        asm volatile("" ::: "memory");
        const unsigned base = (unsigned)(size_t)sStage;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int reg = m >> 1;
            const unsigned fl = oob >> (2 * m);
            const int y1 = (fl & 2u) ? ty[m] : ty[m] + 1;
            const bool outw = ty[m] < w0[reg] || y1 > wlast[reg];
            float tv[C][4];
            if (!P.force_gather && __builtin_amdgcn_ballot_w64(outw) == 0) {
                // an out-of-image tap is read anyway (the word exists: next row, next chunk or the pad) and replaced by zero
                const unsigned nbr = 4u * (unsigned)((wlast[reg] - w0[reg] + 1) * W);      // bytes per channel of this window
                unsigned a0 = base + (reg ? (unsigned)wbytes0 : 0u) + 4u * (unsigned)((ty[m] - w0[reg]) * W + tx[m]);
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    lds_cfloat* p0 = (lds_cfloat*)(size_t)a0;
                    lds_cfloat* p1 = (lds_cfloat*)(size_t)(a0 + 4u * (unsigned)W);
                    tv[ch][0] = p0[0]; tv[ch][1] = p0[1]; tv[ch][2] = p1[0]; tv[ch][3] = p1[1];
                    a0 += nbr;
                }
                combine(m, tv, !any_oob);
            } else {
                // some lane's tap rows leave the window: this mirror pixel from global memory, clamped addresses
                const int dx = (fl & 1u) ? 0 : 1, dy = (fl & 2u) ? 0 : W;
                const float* p = reinterpret_cast<const float*>(img) + ty[m] * W + tx[m];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    tv[ch][0] = p[0]; tv[ch][1] = p[dx]; tv[ch][2] = p[dy]; tv[ch][3] = p[dy + dx];
                    p += HW;
                }
                combine(m, tv, false);
            }
        }
    } else if (staged) {
#pragma unroll
        for (int reg = 0; reg < 2; ++reg) {
            if (reg == 1) lds_only_barrier();                // every tap of the upper region is in registers: the buffer is free
            stage_region(reg);
            __builtin_amdgcn_s_barrier();                    // every wavefront's pieces have landed
            asm volatile("" ::: "memory");
            const unsigned base = (unsigned)(size_t)sStage;
            float tv[2][C][4];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * reg + mm;
                // an out-of-image tap is read anyway (the word exists: next row, next chunk or the pad) and replaced by zero
                unsigned a0 = base + 4u * (unsigned)((ty[m] - y0r[reg]) * W + tx[m]);
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    lds_cfloat* p0 = (lds_cfloat*)(size_t)a0;
                    lds_cfloat* p1 = (lds_cfloat*)(size_t)(a0 + 4u * (unsigned)W);
                    tv[mm][ch][0] = p0[0];
                    tv[mm][ch][1] = p0[1];
                    tv[mm][ch][2] = p1[0];
                    tv[mm][ch][3] = p1[1];
                    a0 += chunk_bytes;
                }
            }
            combine(2 * reg, tv[0], !any_oob);
            combine(2 * reg + 1, tv[1], !any_oob);
        }
    } else {
        // the band's taps reach further than the buffer holds: from global memory, clamped addresses (always readable)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned fl = oob >> (2 * m);
            const int dx = (fl & 1u) ? 0 : 1, dy = (fl & 2u) ? 0 : W;
            const float* p = reinterpret_cast<const float*>(img) + ty[m] * W + tx[m];
            float tv[C][4];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                tv[ch][0] = p[0]; tv[ch][1] = p[dx]; tv[ch][2] = p[dy]; tv[ch][3] = p[dy + dx];
                p += HW;
            }
            combine(m, tv, false);
        }
    }
    lds_only_barrier();                                      // every tap is in registers: the buffer is free
    // results in the buffer, in the output's layout: [channel][upper rows | lower rows][W]
    const unsigned out_chunk = 2u * (unsigned)(rows * W) * 4u;
    if (live) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rl = (m & 2) ? 2 * rows - 1 - (r - ra) : (r - ra);     // lower region stored behind the upper one, mirrored
            const int cc = (m & 1) ? W - 1 - c : c;
            char* dst = reinterpret_cast<char*>(sStage) + 4u * (unsigned)(rl * W + cc);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) *reinterpret_cast<float*>(dst + ch * out_chunk) = res[m][ch];
        }
    }
    lds_only_barrier();                                      // results staged
    // copy-out: per channel the upper range [ra, ra + rows) and the lower range [H - ra - rows, H - ra), 16 bytes per lane
    // (staged back to back per channel; no division: channel loop unrolled, the range is a compare)
    {
        const int seg16 = (rows * W) >> 2;                   // 16-byte pieces of one row range of one plane
        const int nct = (int)blockDim.x;
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
        const unsigned lo_delta = 4u * (unsigned)((H - ra - rows) * W) - 16u * (unsigned)seg16;   // lower range: global minus staged offset
        const unsigned up_delta = 4u * (unsigned)(ra * W);
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            for (int i = tid; i < 2 * seg16; i += nct) {
                const v4f x = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sStage) + (unsigned)ch * out_chunk + 16u * (unsigned)i);
                store16_nt(ob + (unsigned)ch * row_bytes + 16u * (unsigned)i + (i >= seg16 ? lo_delta : up_delta), x);
            }
    }
}

}  // namespace tpspp_span
