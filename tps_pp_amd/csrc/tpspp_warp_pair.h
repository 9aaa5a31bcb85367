// Classic-geometry warp, image-pair kernel (the kernel BASELINE.json configs[1] / bench.py measure).
//
// Replaces, bit for bit: preprocessor/tps_preprocessor.py:71-83 + 270-282 (GridGenerator.build_P_prime: two bmm,
// then F.grid_sample bilinear / border / align_corners) for a mirror-symmetric RBF table (see tpspp_warp.hip).
//
// One workgroup = two images (adjacent in memory), 13 compute + 3 loader wavefronts.  What is different from the
// LDS-staged mirror kernel it supersedes -- each item measured on MI355X, scripts/ubench/warp_lab.hip:
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
typedef __attribute__((address_space(1))) char gchar;

struct PairParams {
    const float* in; const float* ctrl; const float* inv_delta_c;
    const float* packed;   // tpspp_pack_mirror_table layout
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

// geometry of the thread -> pixel mapping (shared with the table packing)
template <int OH, int OW>
struct PairGeo {
    static constexpr int halfW = OW / 2;
    static constexpr int PW = (halfW + 3) & ~3;              // left half-row padded to whole 16-byte groups
    static constexpr int nthr = (OH / 2) * PW;               // compute threads
    static constexpr int NW = (nthr + kWave - 1) / kWave;    // compute wavefronts
    static constexpr int CG = PW / 4;                        // 4-column groups per half-row
    static_assert(OW % 4 == 0 && (OH / 2) % 8 == 0 && nthr % 32 == 0, "needs whole 4 x 8 pixel blocks");
};
// thread t -> (r, c): half-wavefront = block of 4 columns x 8 rows
__host__ __device__ inline void pair_thread_pixel(int t, int CG, int* r, int* c)
{
    const int hw = t >> 5, l5 = t & 31, rg = hw / CG, cg = hw - rg * CG;
    *r = rg * 8 + (l5 >> 2);
    *c = cg * 4 + (l5 & 3);
}

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

constexpr int kPairLoaders = 3;
constexpr int kPairAwait = 6;   // image B's requests start when <= 6 of a loader's requests for A are outstanding,
constexpr int kPairKB = 4;      // flag A is raised after 4 of them have been issued

template <int F, int C, int HC, int WC, int OH, int OW, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_pair_kernel(const PairParams P)
{
    using Geo = PairGeo<OH, OW>;
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = Geo::halfW, nthr = Geo::nthr, NW = Geo::NW, NLOAD = kPairLoaders;
    constexpr int n = OH * OW;
    static_assert(NW + NLOAD <= 16, "too many wavefronts");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;                            // 2 x K*K (one copy per solving wavefront)
    float* sFlag = sInv + 2 * ((K * K + 3) & ~3);          // [0] T rows published, [1] loaders done with A, [2] with B
    float* sImg = sFlag + 4;
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
    constexpr int KK = K * K;
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;
    if (wv < 2) {
        if (lane < F) {
            const int b = (wv == 1 && hasB) ? b0 + 1 : b0;
            const float2 cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[lane];
            cx = cc.x; cy = cc.y;
        }
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
    }
    if (tid < 3) reinterpret_cast<int*>(sFlag)[tid] = 0;
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
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        for (; piece < pieces; piece += NLOAD) dma(piece);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    int r, c;
    pair_thread_pixel(live ? tid : nthr - 1, Geo::CG, &r, &c);
    const bool xdup = (c & ~3) + 4 > halfW;                  // middle group: its x-mirror is another thread's pixel
    unsigned poff[4];                                        // byte offsets of the 4 mirror pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][KG][lane] x 16 bytes = this thread's K values (zero-padded to 4 * KG)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.packed) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;                     // read by out-of-image taps (before wave 0 raises flag T)
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;   // stride K is odd: no conflicts
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {                        // ordered broadcast: the sum is the reference's FMA chain
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(sFlag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wait_flag(sFlag + 0, 2);
    PAIR_STAMP(0);                                           // T ready

    // ---- 16 FMA chains: 4 mirror pixels x (image A, image B) x (x, y), each k-ascending from zero ----
    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
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
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));

    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) of both images are pure VALU work, done
    // while image A is still in flight (the wavefront would idle otherwise).
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const unsigned img_lds = (unsigned)(size_t)sImg, zero_lds = (unsigned)(size_t)sZero;
    unsigned ta[2][4][4];
    float tw[2][4][4];
    auto describe = [&](auto imc) {
        constexpr int im = decltype(imc)::value;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const int b = b0 + im;
                const bool st = live && !((m & 1) && xdup) && (im == 0 || hasB);
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_int2(t.x0, t.y0);
            }
            // a tap outside the image points at a zero word (one per channel plane, H*W apart, behind the staged pair):
            // a channel is then four LDS reads with immediate offsets, no per-channel select
            const unsigned base = img_lds + (unsigned)(im * img_elems * 4);
            const bool inxy = t.inx && t.iny;
            ta[im][m][0] = base + 4u * (unsigned)t.o00;
            ta[im][m][1] = t.inx ? ta[im][m][0] + 4u : zero_lds;
            ta[im][m][2] = t.iny ? base + 4u * (unsigned)t.o10 : zero_lds;
            ta[im][m][3] = inxy ? base + 4u * (unsigned)t.o10 + 4u : zero_lds;
            tw[im][m][0] = t.nw; tw[im][m][1] = t.ne; tw[im][m][2] = t.sw; tw[im][m][3] = t.se;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(ta[im][m][q]), "+v"(tw[im][m][q]));
    };
    describe(std::integral_constant<int, 0>{});
    describe(std::integral_constant<int, 1>{});
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
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) { store_image(b0); break; }    // odd batch: the last group has no image B
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        PAIR_STAMP(2 + 2 * im);
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(size_t)(ta[im][m][q]) + ch * HW);
        if (im == 1) {
            // image A's output pieces leave while the LDS serves image B's tap reads: their issue is back-pressured
            // by HBM and would otherwise sit on the critical path between the two images
            __builtin_amdgcn_sched_barrier(0);
            store_image(b0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float res[4][C];
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
        if (im == 1) lds_only_barrier();                     // everybody has read image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        PAIR_STAMP(3 + 2 * im);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) ostage[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
        }
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

// [wavefront][KG][lane][4]: the table values of thread (wavefront, lane)'s pixel, q = 4 j .. 4 j + 3
__global__ void __launch_bounds__(256)
pack_mirror_table_kernel(const float* __restrict__ p_hat, int p_hat_ld, int OW, int CG, int nthr, int K,
                         float* __restrict__ packed)
{
    const int KG = (K + 3) / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // index into packed
    const int total = ((nthr + kWave - 1) / kWave) * KG * kWave * 4;
    if (i >= total) return;
    const int comp = i & 3, l = (i >> 2) & (kWave - 1), j = (i >> 8) % KG, w = (i >> 8) / KG;
    const int t = w * kWave + l, q = 4 * j + comp;
    float val = 0.0f;
    if (t < nthr && q < K) {
        int r, c;
        pair_thread_pixel(t, CG, &r, &c);
        val = p_hat[(size_t)(r * OW + c) * p_hat_ld + q];
    }
    packed[i] = val;
}

// LDS bytes of one workgroup and the offsets the kernel needs
template <int F, int C, int HC, int WC, int OH, int OW>
inline size_t pair_lds_bytes(int* zero_off, int* out_off)
{
    constexpr int K = F + 3;
    const int pieces = (2 * C * HC * WC * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    *out_off = pieces * 256 + (C - 1) * HC * WC + 4;
    return (size_t)(4 * K + 2 * ((K * K + 3) & ~3) + 4) * sizeof(float) + (size_t)(*out_off) * 4 + (size_t)C * OH * OW * 4;
}

}  // namespace tpspp_pair
