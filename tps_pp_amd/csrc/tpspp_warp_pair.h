// Classic-geometry warp, image-pair kernel (the kernel BASELINE.json configs[1] / bench.py measure).
//
// Replaces, bit for bit: preprocessor/tps_preprocessor.py:71-83 + 270-282 (GridGenerator.build_P_prime: two bmm,
// then F.grid_sample bilinear / border / align_corners) for a mirror-symmetric RBF table (see tpspp_warp.hip).
//
// One workgroup = two images (adjacent in memory), 12 compute + 3 loader wavefronts (round 4: three compute wavefronts
// per SIMD, see PairGeo).  What is different from the LDS-staged mirror kernel it supersedes -- each item measured on
// MI355X, scripts/ubench/pair_lab.hip (rounds 2-3: warp_lab.hip / img_lab.hip):
//   * the pair is pipelined: image A is sampled while image B is still landing.  HBM does not serve requests in
//     arrival order, so the loaders hold image B's requests back until most of A's have been served and raise
//     flag A before the bulk of B's are issued; flags are LDS words, polled, so no wavefront waits at a barrier
//     for an event it does not need (the loaders never wait for anyone);
//   * results go to an image-shaped LDS staging buffer and leave as a flat copy, 16 bytes per lane, 1 KB per
//     wavefront store, with the nt policy: whole 128-byte lines stream to HBM during the kernel instead of sitting
//     dirty in the L2 until the end-of-kernel write-back (2.4 us of the launch-to-launch gap); image A's stores are
//     issued while the LDS serves image B's tap reads;
//   * image DMA with the nt policy (lands ~20 % earlier: nothing is evicted to make room for lines read once);
//   * thread -> pixel: a half-wavefront owns 4 columns x 8 rows, i.e. 32 different LDS banks at a row pitch of 100
//     floats (row-major lanes wrapped into the next row 4 banks further: a third of all LDS cycles were conflicts);
//   * the RBF table is read from a packed copy, 6 x 16 bytes per thread (was 23 x 4 bytes: the vector memory unit
//     needs 16 cycles per wavefront instruction regardless of width);
//   * the control points are requested before the entry barrier that releases the loaders: ahead of this CU's image
//     traffic instead of behind it.
// The arithmetic is unchanged: T rows and grid coordinates are the k-ascending fp32 FMA chains from zero, taps and
// weights as in tpspp_warp_dev.h.  Compiled with -ffp-contract=off.
#pragma once
#include "tpspp_warp_dev.h"

namespace tpspp_pair {

using namespace tpspp_dev;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte piece at a 4-byte-aligned address
typedef __attribute__((address_space(1))) char gchar;

struct PairParams {
    const float* in; const float* ctrl; const float* inv_delta_c;
    const float* packed;   // tpspp_pack_mirror_table layout
    const float* p_hat;    // the table itself, row-major (F + 3 columns used), for the centre strip's pixels
    int p_hat_ld;
    int N;
    float* out; float* grid; int32_t* idx;
    int zero_off;          // float offsets from the staged pair: zero words for out-of-image taps,
    int out_off;           //                                      output staging buffer
    long long* trace;      // optional: 8 stamps per workgroup (tpspp_warp_set_trace)
};

template <int F>
__device__ __forceinline__ constexpr int perm_y(int k) { return (k + F / 2) % F; }
template <int F>
__device__ __forceinline__ constexpr int perm_x(int k) { return k < F / 2 ? F / 2 - 1 - k : F + F / 2 - 1 - k; }

// 16-byte store, nt policy, as inline asm with a full 64-bit per-lane address.  s_nop: the VMEM store-data hazard
// is invisible to the compiler inside inline asm.
__device__ __forceinline__ void store16_nt(gchar* p, v4f v)
{
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void wait_flag(const float* flag_word, int want)
{
    const volatile int* f = reinterpret_cast<const volatile int*>(flag_word);
    while (*f < want) __builtin_amdgcn_s_sleep(1);
}
// The same poll as an LDS instruction.  The generic-pointer form above compiles to flat_load_dword + s_waitcnt
// vmcnt(0) lgkmcnt(0): every poll goes through the vector memory unit (behind the CU's DMA issue) and waits for the
// wavefront's outstanding global stores.  Compute wavefronts only (no LDS-DMA of their own to be ordered behind).
__device__ __forceinline__ void wait_flag_lds(const float* flag_word, int want)
{
    typedef __attribute__((address_space(3))) const volatile int lds_cvint;
    lds_cvint* f = (lds_cvint*)(size_t)(unsigned)(size_t)flag_word;
    while (__builtin_amdgcn_readfirstlane(*f) < want) __builtin_amdgcn_s_sleep(1);
}

// flag word at `addr` + OFF bytes += one, without a compiler-visible LDS access (see the loader's note on LDS-DMA)
template <int OFF>
__device__ __forceinline__ void flag_add(unsigned addr, int one)
{
    asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr), "v"(one), "n"(OFF) : "memory");
}

// make_taps() with fewer vector-ALU instructions, bit for bit the same results (the image-pair kernel is bound by
// VALU issue on the SIMD that hosts four of its 13 compute wavefronts):
//   * ((g + 1) * 0.5) * (W - 1) as (g + 1) * ((W - 1) / 2): the halving is exact (g + 1 is 0 or >= 2^-24 in magnitude,
//     never subnormal; both forms overflow together) and (W - 1) / 2 is exact in fp32 for every W < 2^24, so both
//     forms round the same real number once;
//   * the upper clamp as v_min_f32 (the lower clamp already removed NaN), the fraction as v_fract_f32 (exactly
//     ix - floor(ix) for ix >= 0), the index as v_cvt_flr_i32_f32;
//   * the south row's offset as the north row's + W.
struct TapsLite {
    int o00;                  // offset of the north-west tap inside a plane
    float nw, ne, sw, se;
    float wx, wy;             // fractional parts: nw = (1-wy)(1-wx), ne = (1-wy) wx, sw = wy (1-wx), se = wy wx
    bool inx, iny;            // is the east column / south row inside the plane
    int x0, y0;
};
__device__ __forceinline__ TapsLite make_taps_lite(float gx, float gy, int H, int W)
{
    TapsLite t;
    const float limx = (float)(W - 1), limy = (float)(H - 1);
    float ix = (gx + 1.0f) * (0.5f * limx);
    float iy = (gy + 1.0f) * (0.5f * limy);
    ix = (ix > 0.0f) ? ix : 0.0f;   // NaN -> 0
    iy = (iy > 0.0f) ? iy : 0.0f;
    ix = __builtin_fminf(ix, limx);
    iy = __builtin_fminf(iy, limy);
    const float w = __builtin_amdgcn_fractf(ix), nn = __builtin_amdgcn_fractf(iy);
    int x0, y0;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(x0) : "v"(ix));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(y0) : "v"(iy));
    const float e = 1.0f - w, s = 1.0f - nn;
    t.nw = s * e; t.ne = s * w; t.sw = nn * e; t.se = nn * w;
    t.wx = w; t.wy = nn;
    t.inx = ix < limx;              // x0 + 1 < W  <=>  floor(ix) < W - 1  <=>  ix < W - 1
    t.iny = iy < limy;
    t.o00 = __mul24(y0, W) + x0;
    t.x0 = x0; t.y0 = y0;
    return t;
}

constexpr int kPairFlags = 4;   // flag words in front of the staged pair (a multiple of 4: the DMA target stays 16-byte aligned)
constexpr int kPairLoaders = 3;
constexpr int kPairAwait = 6;   // image B's requests start when <= 6 of a loader's requests for A are outstanding,
constexpr int kPairKB = 4;      // flag A is raised after 4 of them have been issued

// VAR: lab switch (scripts/ubench/pair_lab.hip); the library instantiates 0.
//   bit 0: the grid chains issue as packed fp32 FMAs (v_pk_fma_f32: (x, y) of one pixel against (T.x, T.y) of one image;
//          the same IEEE fma per component, so bit-identical).  Measured slower (profiles/r04_warp_lab.txt): a
//          v_pk_fma_f32 occupies the SIMD for two passes, and the kernel is bound by VALU time, not by issue slots.
constexpr int kPairVarPacked = 1;
//   bit 1 (round 5): the grid chains run on the MATRIX pipe.  v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4 x 4 outer
//          products a_i * b_j added to the accumulator: one IEEE fp32 fma per element, i.e. one step of the reference's
//          k-ascending chain (scripts/ubench/mfma_exact_probe.hip: the f32 matrix instructions are bit-identical to fmaf
//          chains).  Block = 4 consecutive lanes; A (lane i of the block) = component i of (TxA, TyA, TxB, TyB) of step q
//          -- one LDS word of sT --, B (lane j) = the lane's OWN table value of the mirror pixel: the accumulator of lane j
//          is (x, y) of its pixel for both images -- exactly the 4 chains it used to run on the vector ALU.  23 x 4
//          matrix instructions per wavefront replace 368 v_fmac (of ~850 vector instructions: the kernel is bound by vector-ALU
//          time).
constexpr int kPairVarMfma = 2;
typedef float v2f __attribute__((ext_vector_type(2)));

// Thread -> pixel, round 4.  The kernel is bound by vector-ALU time on the busiest SIMD (a wavefront's ~850 VALU
// instructions x the compute wavefronts of that SIMD).  A workgroup's wavefronts go to the four SIMDs round-robin, so
// the 13 compute wavefronts of the round-2 mapping (800 quads of 4 mirror pixels + 32 padding threads) put FOUR on
// one SIMD and three on the others.  Now: the 48 left columns of the left half (12 groups of 4) x 16 rows = 768 quads
// = 12 wavefronts, three per SIMD; what is left is the strip of the 4 centre columns (halfW - 2 .. halfW + 1), 128
// pixels per image, one pixel per lane (no mirror partner) for wavefronts 0 .. 3, which sit on four different SIMDs:
// wavefronts 0 / 1 take image A's strip rows 0-15 / 16-31, wavefronts 2 / 3 image B's.  A strip pixel costs ~100 VALU
// instructions; its table row comes straight from P_hat (6 x 16 bytes per lane, four wavefronts only).
template <int OH, int OW>
struct PairGeo {
    static constexpr int halfW = OW / 2;
    static constexpr int PW = (halfW + 3) & ~3;              // the packed table's padded half-row (its layout is unchanged)
    static constexpr int CGP = PW / 4;                       // 4-column groups per half-row in the packed table
    static constexpr int CG = (halfW - 2) / 4;               // groups handled as mirror quads
    static constexpr int SC0 = CG * 4;                       // first strip column
    static constexpr int SW = OW - 2 * SC0;                  // strip width
    static constexpr int nthr = (OH / 2) * CG * 4;           // quad threads
    static constexpr int NW = nthr / kWave;                  // compute wavefronts
    static_assert(OW % 4 == 0 && (OH / 2) % 8 == 0 && nthr % kWave == 0, "needs whole 4 x 8 pixel blocks, whole wavefronts");
    static_assert(SW == 4 && OH * SW == 2 * kWave, "the centre strip is 4 columns, two wavefronts per image");
    static_assert(NW % 4 == 0 && NW >= 4, "the compute wavefronts spread evenly over the four SIMDs");
};
// packed-table thread t -> (r, c): half-wavefront = block of 4 columns x 8 rows (CG here = groups of the PACKED table)
__host__ __device__ inline void pair_thread_pixel(int t, int CG, int* r, int* c)
{
    const int hw = t >> 5, l5 = t & 31, rg = hw / CG, cg = hw - rg * CG;
    *r = rg * 8 + (l5 >> 2);
    *c = cg * 4 + (l5 & 3);
}

template <int F, int C, int HC, int WC, int OH, int OW, bool AUX, bool TRACE, int VAR = 0>
__global__ void __launch_bounds__(1024)
tps_warp_pair_kernel(const PairParams P)
{
    using Geo = PairGeo<OH, OW>;
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int NW = Geo::NW, NLOAD = kPairLoaders;
    constexpr int n = OH * OW;
    static_assert(NW + NLOAD <= 16, "too many wavefronts");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sFlag = smem + 4 * K + 2 * ((K * K + 3) & ~3);  // (the gap: the T solve's LDS copy of inv_delta_C until round 3)          // [0] T rows published, [1] loaders done with A, [2] with B
    float* sImg = sFlag + kPairFlags;
    float* sZero = sImg + P.zero_off;
    float* sOut = sImg + P.out_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * 2;
    const bool hasB = (b0 + 1) < P.N;
    constexpr int HW = H * W;
    constexpr int img_elems = C * HW;

    unsigned ts[8];
#define PAIR_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) wall0 = (long long)wall_clock64();
    PAIR_STAMP(7);                                           // kernel entry

    // the T-solve inputs are requested before anything else on this CU (wavefront g -> image b0 + g; lane i keeps
    // control point i); the loaders are held at the entry barrier until these requests are on their way
    // Row `lane` of inv_delta_C comes straight from global memory, 16 bytes at a time (round 4; through an LDS copy
    // before: T ready 0.15 us later).  Pieces 0 .. KGI-2 hold columns 4j .. 4j+3; the last piece starts at column K-4
    // so that the last row does not read past the matrix.
    constexpr int KGI = (K + 3) / 4;
    float hrowv[KGI * 4];
    float cx = 0.0f, cy = 0.0f;
    if (wv < 2) {
        if (lane < F) {
            const int b = (wv == 1 && hasB) ? b0 + 1 : b0;
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
    if (tid < kPairFlags) reinterpret_cast<int*>(sFlag)[tid] = 0;
    lds_only_barrier();                                      // the only barrier every wavefront takes part in

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;                 // tail lanes re-read a valid address
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, 2 /* nt */);
        };
        // The flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler insert s_waitcnt vmcnt(0) first, which would turn "A has landed"
        // into "everything has landed".  The flag updates are inline asm for the same reason (a visible LDS access
        // is ordered behind ALL outstanding LDS-DMA of the wavefront).
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPairAwait) : "memory");
#pragma unroll
        for (int i = 0; i < kPairKB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        // vmcnt retires in order: once at most kPairKB requests are outstanding and kPairKB of image B's have been
        // issued behind image A's, A is complete.  The last workgroup of an odd batch has no image B: nothing was
        // issued behind A, so A is complete only when nothing at all is outstanding.
        if (hasB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPairKB) : "memory");
        else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) flag_add<0>(fa, one);
        for (; piece < pieces; piece += NLOAD) dma(piece);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) flag_add<0>(fb, one);
        return;
    }

    // ================= compute wavefronts =================
    // quad thread: half-wavefront hw = 4 columns x 8 rows of the upper left quadrant
    const int hw = tid >> 5, l5 = tid & 31;
    const int rg = hw / Geo::CG, cg = hw - rg * Geo::CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    unsigned poff[4];                                        // byte offsets of the 4 mirror pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));
    // centre-strip pixel of this lane (wavefronts 0 .. 3): image sim, row srow, column scol
    const bool strip = wv < 4 && (wv < 2 || hasB);
    const int sim = wv >> 1;
    const int srow = (wv & 1) * (OH / 2) + (lane >> 2), scol = Geo::SC0 + (lane & 3);
    const unsigned spoff = 4u * (unsigned)(srow * OW + scol);

    // packed table: [wavefront][KG][lane] x 16 bytes = the K values of a quad thread (zero-padded to 4 * KG), in the
    // thread order of the round-2 mapping (CGP column groups per row group): this thread's block is half-wavefront
    // rg * CGP + cg there
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const int t_old = (rg * Geo::CGP + cg) * 32 + l5;
        const v4f* pk = reinterpret_cast<const v4f*>(P.packed) + (size_t)(t_old >> 6) * KG * kWave + (t_old & (kWave - 1));
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    float u[KG * 4];                                         // the strip pixel's table row (row-major P_hat, K columns;
    if (wv < 4) {                                            // the last piece over-reads into the next row: never the
        const float* row = P.p_hat + (size_t)(srow * OW + scol) * P.p_hat_ld;   // table's last one)
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f_a4 x = *reinterpret_cast<const v4f_a4*>(row + 4 * j);
            u[4 * j] = x[0]; u[4 * j + 1] = x[1]; u[4 * j + 2] = x[2]; u[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;                     // read by out-of-image taps (before wave 0 raises flag T)
    if (wv < 2) {
        float ax = 0.0f, ay = 0.0f;
        static_for<K>([&](auto qc) {                         // ordered broadcast: the sum is the reference's FMA chain
            constexpr int q = decltype(qc)::value;
            constexpr int idx = (q / 4 < KGI - 1) ? q : 4 * (KGI - 1) + (q - (K - 4));   // column q's slot in hrowv
            ax = fmaf(hrowv[idx], readlane_f(cx, q), ax);
            ay = fmaf(hrowv[idx], readlane_f(cy, q), ay);
        });
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(sFlag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wait_flag_lds(sFlag + 0, 2);
    PAIR_STAMP(0);                                           // T ready
    // The padding component of the last 16-byte table piece is never used: without this pin the register allocator
    // recycles it right behind the load, which costs an s_waitcnt vmcnt(0) (write-after-write) in front of everything
    // that follows -- the strip's loads and, on wavefronts 0 / 1, the T solve.
    asm volatile("" ::"v"(v[KG * 4 - 1]), "v"(u[KG * 4 - 1]));

    // ---- the strip pixel's two chains first (they free the 24 registers of its table row) ----
    float sgx = 0.0f, sgy = 0.0f;
    if (wv < 4) {
        static_for<K>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const float2 t = reinterpret_cast<const float2*>(sT + q)[sim];
            sgx = fmaf(u[q], t.x, sgx);
            sgy = fmaf(u[q], t.y, sgy);
        });
        asm volatile("" : "+v"(sgx), "+v"(sgy));
    }

    // ---- 16 FMA chains: 4 mirror pixels x (image A, image B) x (x, y), each k-ascending from zero ----
    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    if constexpr ((VAR & kPairVarMfma) != 0) {
        v4f macc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) macc[m] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        const float* tcomp = reinterpret_cast<const float*>(sT) + (lane & 3);      // this lane's row of A: component lane % 4 of T[q]
        static_for<K>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const float tq = tcomp[4 * q];
            float val[4];
            if constexpr (q == 0) {
                val[0] = val[1] = val[2] = val[3] = v[0];
            } else if constexpr (q == 1) {
                val[0] = v[1]; val[1] = -v[1]; val[2] = v[1]; val[3] = -v[1];
            } else if constexpr (q == 2) {
                val[0] = v[2]; val[1] = v[2]; val[2] = -v[2]; val[3] = -v[2];
            } else {
                constexpr int k = q - 3;
                val[0] = v[3 + k];
                val[1] = v[3 + perm_x<F>(k)];
                val[2] = v[3 + perm_y<F>(k)];
                val[3] = v[3 + perm_x<F>(perm_y<F>(k))];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) macc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(tq, val[m], macc[m], 0, 0, 0);
        });
#pragma unroll
        for (int m = 0; m < 4; ++m) { gx[m][0] = macc[m][0]; gy[m][0] = macc[m][1]; gx[m][1] = macc[m][2]; gy[m][1] = macc[m][3]; }
    } else
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        if constexpr (VAR & kPairVarPacked) {
            const v2f tA = {t.x, t.y}, tB = {t.z, t.w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const v2f vv = {val[m], val[m]};
                v2f a = {gx[m][0], gy[m][0]}, b = {gx[m][1], gy[m][1]};
                a = __builtin_elementwise_fma(vv, tA, a);
                b = __builtin_elementwise_fma(vv, tB, b);
                gx[m][0] = a.x; gy[m][0] = a.y; gx[m][1] = b.x; gy[m][1] = b.y;
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
                gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
                gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
                gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
            }
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));

    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
    // Tap descriptors (4 LDS byte addresses + 4 weights per pixel) of both images are pure VALU work, done while image
    // A is still in flight.  A tap outside the image points at a zero word (one per channel plane, H*W apart, behind
    // the staged pair): a channel is then four LDS reads with immediate offsets, no per-channel select.
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const unsigned img_lds = (unsigned)(size_t)sImg, zero_lds = (unsigned)(size_t)sZero;
    auto describe_px = [&](float pgx, float pgy, int im, unsigned pix_off, bool st, unsigned (&a)[4], float (&w)[4]) {
        const TapsLite t = make_taps_lite(pgx, pgy, H, W);
        if constexpr (AUX) {
            const int b = b0 + im;
            if (P.grid && st)
                *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * pix_off) = make_float2(pgx, pgy);
            if (P.idx && st)
                *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * pix_off) = make_int2(t.x0, t.y0);
        }
        const unsigned a0 = img_lds + (unsigned)(im * img_elems * 4) + 4u * (unsigned)t.o00;
        a[0] = a0;
        a[1] = t.inx ? a0 + 4u : zero_lds;
        a[2] = t.iny ? a0 + 4u * W : zero_lds;
        a[3] = (t.inx && t.iny) ? a0 + 4u * W + 4u : zero_lds;
        w[0] = t.nw; w[1] = t.ne; w[2] = t.sw; w[3] = t.se;
    };
    unsigned ta[2][4][4];
    float tw[2][4][4];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
#pragma unroll
        for (int m = 0; m < 4; ++m) describe_px(gx[m][im], gy[m][im], im, poff[m], im == 0 || hasB, ta[im][m], tw[im][m]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(ta[im][m][q]), "+v"(tw[im][m][q]));
    }
    unsigned sa[4] = {zero_lds, zero_lds, zero_lds, zero_lds};
    float sw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (wv < 4) describe_px(sgx, sgy, sim, spoff, strip, sa, sw);
    PAIR_STAMP(1);                                           // grid + tap descriptors done

    constexpr int NOUT = (out16 + nct - 1) / nct;            // 16-byte output pieces per thread and image
    v4f ostage[NOUT];                                        // image A's pieces between their LDS read and their store
    auto store_image = [&](int b) {
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) store16_nt(ob + 16u * (unsigned)e, ostage[i]);
        }
    };
    auto read_staged = [&](const float* stage) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) ostage[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(stage) + 16 * e);
        }
    };
    auto load_taps = [&](int im, float (&tv)[4][C][4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(size_t)(ta[im][m][q]) + ch * HW);
    };
    // weighted taps of the 4 mirror pixels (+ this lane's strip pixel when `mystrip`, wavefront-uniform: its taps are
    // read after the quads' have been consumed, the registers are free then) -> staging buffer `stage`
    auto finish_image = [&](auto barc, int im, float (&tv)[4][C][4], bool mystrip, float* stage) {
        constexpr bool BAR = decltype(barc)::value;
        float res[4][C], sres[C];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[im][m][0];
                acc = fmaf(tv[m][ch][1], tw[im][m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[im][m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[im][m][3], acc);
                res[m][ch] = acc;
            }
        if (mystrip) {
            __builtin_amdgcn_sched_barrier(0);
            float sv[C][4];
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) sv[ch][q] = *((lds_cfloat*)(size_t)(sa[q]) + ch * HW);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = sv[ch][0] * sw[0];
                acc = fmaf(sv[ch][1], sw[1], acc);
                acc = fmaf(sv[ch][2], sw[2], acc);
                acc = fmaf(sv[ch][3], sw[3], acc);
                sres[ch] = acc;
            }
        }
        if (BAR) lds_only_barrier();                         // everybody has read the previous image out of `stage`
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(stage) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        if (mystrip) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(stage) + ch * (int)row_bytes + spoff) = sres[ch];
        }
    };

    // Image A, then image B.  (Tried in round 4 and rejected, profiles/r04_warp_lab.txt: image B's tap reads issued in
    // front of the "A staged" barrier with B's results staged over A's dead input -- one barrier fewer, but every
    // wavefront then waits for image B to land before image A may leave, and B lands late.)
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) { store_image(b0); break; }    // odd batch: the last group has no image B
        wait_flag_lds(sFlag + 1 + im, NLOAD);                // image `im` has landed
        PAIR_STAMP(2 + 2 * im);
        float tv[4][C][4];
        load_taps(im, tv);
        if (im == 1) {
            // image A's output pieces leave while the LDS serves image B's tap reads: their issue is back-pressured
            // by HBM and would otherwise sit on the critical path between the two images
            __builtin_amdgcn_sched_barrier(0);
            store_image(b0);
            __builtin_amdgcn_sched_barrier(0);
            finish_image(std::true_type{}, im, tv, strip && sim == im, sOut);
        } else {
            finish_image(std::false_type{}, im, tv, strip && sim == im, sOut);
        }
        lds_only_barrier();                                  // results of image `im` staged
        PAIR_STAMP(3 + 2 * im);
        read_staged(sOut);
        if (im == 1) {
            store_image(b0 + 1);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(ostage[i]));
        }
    }
    if (TRACE && wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PAIR_STAMP(6);                                       // this wavefront's stores retired
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 8;
            if (!hasB) ts[4] = ts[5] = ts[3];
#pragma unroll
            for (int i = 0; i < 7; ++i) t[i] = (long long)(ts[i] - ts[7]);
            t[7] = wall0;
        }
    }
#undef PAIR_STAMP
}

// (The packed table -- [wavefront][KG][lane] x 16 bytes in the thread order of the round-2 mapping, 13 column groups of
// 4 per row group -- is written by tpspp_img::pack_img_table_kernel with QP = 1: tpspp_prepare_mirror_table.)

// LDS bytes of one workgroup and the offsets the kernel needs
template <int F, int C, int HC, int WC, int OH, int OW>
inline size_t pair_lds_bytes(int* zero_off, int* out_off)
{
    constexpr int K = F + 3;
    const int pieces = (2 * C * HC * WC * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    *out_off = pieces * 256 + (C - 1) * HC * WC + 4;
    return (size_t)(4 * K + 2 * ((K * K + 3) & ~3) + kPairFlags) * sizeof(float) + (size_t)(*out_off) * 4 + (size_t)C * OH * OW * 4;
}

}  // namespace tpspp_pair
