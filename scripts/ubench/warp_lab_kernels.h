// KERNEL LAB, not product code: the variants of the classic-geometry warp that were measured on the way to
// tps_pp_amd/csrc/tpspp_warp_pair.h (round 2).  Built only by scripts/ubench/warp_lab.hip; every variant is checked bit for
// bit against the library and timed interleaved with the others.  Kept as the record behind DESIGN.md section 4:
//   q4   thread = 4 consecutive pixels x 4 mirrors, 7 fat wavefronts         16 us   (VALU phases need >= 3 wavefronts / SIMD)
//   m    the round-1 mapping with knobs (store policy, nt DMA, loader count, debug switches: no stores / no DMA)
//   m2   16-byte stores of quad-transposed (DPP) registers                    partial lines: write-through at 4 TB/s
//   m3   image A / B pipeline through barriers, LDS-staged flat output copy  10.2 us (flag A was raised late: vmcnt(0))
//   m4   dedicated T-solver wavefront, unthrottled loaders                    13.7 us (control points behind the DMA flood)
//   m5   control points before the entry barrier, LDS flags                   10.5 us
//   m7   packed table (6 x 16-byte loads per thread)                          10.4 us
//   m8/m9 tap descriptors as 32-bit LDS addresses, hoisted; B held back behind A
//   m10  4 x 8 pixel blocks per half-wavefront (bank conflicts 33 % -> 14 %); flag raised before B's bulk   9.7 us
//   m11  every wavefront issues image A's DMA (inline asm)                    no gain: HBM-bound
//   m12  table loads deferred until T is known                                no gain: same L2 time, later
//   m13/m14 flag A before the bulk of B's requests; A's stores behind B's tap reads          9.6 us
//   m15  both descriptor sets before image A lands  = the production kernel   9.5 us (8.9 - 9.8 by box)
//   m16  image B through registers + ds_write_b128                            11.0 us
//   m17  ds_write_addtid_b32 staging in thread order                          +0.35 us
//   m18  plane-granular pipeline (6 units per pair)                           10.3 us (a barrier and 16 reads in flight per unit)
//
// (first variant) Classic-geometry warp, "q4" kernel: the LDS-staged mirror kernel re-cut for few, fat wavefronts.
//
// Replaces (same arithmetic, bit for bit): preprocessor/tps_preprocessor.py:71-83, 270-282
// (GridGenerator.build_P_prime + F.grid_sample) for a mirror-symmetric RBF table.
//
// Why another kernel: at batch 512 the 256-pair launch of tps_warp_lds_mirror_kernel is one wave of
// 1024-thread workgroups; 4096 wavefronts take the dispatcher 1.5-4 us to start, every output value
// is a 4-byte store instruction and a dedicated loader trio sits idle after the first microsecond.
// Here a thread owns a UNIT = four consecutive pixels of a quadrant row and their three mirror images
// (16 output pixels of one image):
//   * table rows arrive as one 16-byte load per RBF column (a quarter of the load instructions),
//     results leave as 16-byte stores (a quarter of the store instructions, 1 KB per wavefront store);
//   * 208 units per 32x100 image: an image pair is 416 threads = 7 wavefronts (was 16), an image 4;
//   * one (or two) loader wavefronts stream the images HBM -> LDS with global_load_lds (1 KB per
//     instruction).  The compute wavefronts must not issue that DMA themselves: the compiler orders
//     every LDS access of a wavefront behind its own outstanding LDS-DMA (it cannot prove that sT /
//     sInv do not alias the DMA target), i.e. a wavefront that loads would wait for the whole image
//     before it may read T.
// The FMA chains are unchanged: each pixel's grid coordinate is the k-ascending fp32 chain from zero.
#pragma once
#include "tpspp_warp_dev.h"

namespace tpspp_q4 {

using namespace tpspp_dev;

struct Q4Params {
    const float* in; int C, H, W;
    const float* ctrl; const float* inv_delta_c; const float* p_hat_t;
    int N, n, Ho, Wo;
    float* out; float* grid; int32_t* idx;
    int units;          // (Ho/2) * ceil(Wo/8)
    int zero_off;       // float offset (from the staged images) of the zero words for out-of-image taps
    long long* trace;   // optional: 8 shader-clock stamps per workgroup
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) char gchar;

template <int F>
__device__ __forceinline__ constexpr int perm_y(int k) { return (k + F / 2) % F; }
template <int F>
__device__ __forceinline__ constexpr int perm_x(int k) { return k < F / 2 ? F / 2 - 1 - k : F + F / 2 - 1 - k; }

// 16-byte store, wave-uniform 64-bit base + 32-bit lane offset.
// MODE 0: plain; 1: nt; 2: sc1 (agent-scope write-through); 3: sc0 sc1
// 16-byte store with a cache policy.  MODE 0: plain (compiler-generated); 1: nt; 2: sc1; 3: sc0 sc1 -- those as
// inline asm with a full 64-bit per-lane address (an "s" base operand is not safe: the compiler may keep a uniform
// pointer in VGPRs).  s_nop: the VMEM store-data hazard is not visible to the compiler inside inline asm.
template <int MODE>
__device__ __forceinline__ void store16(gchar* base, unsigned voff, v4f v)
{
    if constexpr (MODE == 0) {
        *reinterpret_cast<__attribute__((address_space(1))) v4f*>(base + voff) = v;
    } else {
        gchar* p = base + voff;
        if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }
}

// IMGS: images per workgroup (1 or 2).  NW: compute wavefronts, NLOAD: loader wavefronts (the last ones).
// FIRST: DMA pieces per loader issued before the T barrier (the rest after it; the control points then
// travel ahead of most of the image traffic of this CU).  STORE: see store16.  LDNT: image DMA with nt.
template <int F, int C, int HC, int WC, int IMGS, int NW, int NLOAD, int FIRST, int STORE, int LDNT, bool AUX>
__global__ void __launch_bounds__((NW + NLOAD) * 64)
tps_warp_q4_kernel(const Q4Params P)
{
    constexpr int K = F + 3;
    const int H = HC > 0 ? HC : P.H;
    const int W = WC > 0 ? WC : P.W;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* sT = reinterpret_cast<float2*>(smem);           // IMGS x K x (Tx, Ty)
    float* sInv = smem + 2 * IMGS * K + (2 * IMGS * K & 2); // K*K, 16-byte aligned start
    float* sImg = sInv + ((K * K + 3) & ~3);                // staged images, contiguous
    float* sZero = sImg + P.zero_off;                       // C zero words, H*W apart, behind the DMA pieces

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * IMGS;
    const int nimg = IMGS == 1 ? 1 : min(IMGS, P.N - b0);
    const int HW = H * W;
    const int img_elems = C * HW;

    if (wv == 0) stamp(P.trace, 0);

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int total_bytes = nimg * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;                // tail lanes re-read a valid address
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        int piece = wv - NW;
        for (int i = 0; i < FIRST && piece < pieces; ++i, piece += NLOAD) dma(piece);
        lds_only_barrier();      // matches the T barrier of the compute wavefronts
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (wv == NW) stamp(P.trace, 5);                    // DMA issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv == NW) stamp(P.trace, 6);                    // DMA landed
        __syncthreads();
        return;
    }

    // ---- T-solve inputs first: they head this CU's memory queue (wavefront g -> image b0 + g) ----
    constexpr int KK = K * K;
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;
    if (wv < nimg) {
        if (lane < F) {
            const float2 cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)(b0 + wv) * F * 2)[lane];
            cx = cc.x; cy = cc.y;
        }
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
    }

    // ---- this thread's unit: image `im`, quadrant row r, 16-byte group u of that row ----
    const int G = P.Wo >> 2;                                // 16-byte groups per output row
    const int U = (G + 1) >> 1;                             // units per quadrant row
    int im = 0, uid = tid;
    if (IMGS == 2 && tid >= P.units) { im = 1; uid = tid - P.units; }
    const bool live = uid < P.units && im < nimg;
    if (!live) { uid = P.units - 1; im = im < nimg ? im : 0; }
    const int r = uid / U, u = uid - r * U;
    const bool xmir = (u != G - 1 - u);                     // false: the row's middle group is its own x-mirror
    // byte offsets inside one output plane of the four 16-byte groups this unit writes
    const unsigned o_a = 4u * (unsigned)(r * P.Wo + 4 * u);                       // (r, group u)
    const unsigned o_x = 4u * (unsigned)(r * P.Wo + 4 * (G - 1 - u));             // x-mirror group
    const unsigned o_y = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + 4 * u);          // y-mirror row
    const unsigned o_xy = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + 4 * (G - 1 - u));

    // ---- table rows of the unit's four pixels: K 16-byte loads, consumed after the T barrier ----
    v4f v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const v4f*>(pht + q * row_bytes + o_a);
    }

    if (wv < nimg) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;   // stride K is odd: no conflicts
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) sT[wv * K + lane] = make_float2(ax, ay);
    }
    if (tid < C) sZero[tid * HW] = 0.0f;                      // read by out-of-image taps
    lds_only_barrier();
    if (wv == 0) stamp(P.trace, 1);                           // T ready

    // ---- 32 FMA chains: 4 pixels x 4 mirrors x (x, y), each k-ascending from zero ----
    float gx[4][4], gy[4][4];                                 // [mirror][pixel]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) gx[m][i] = gy[m][i] = 0.0f;
    const float2* tT = sT + im * K;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float2 t = tT[q];
        v4f val[4];
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
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gx[m][i] = fmaf(val[m][i], t.x, gx[m][i]);
                gy[m][i] = fmaf(val[m][i], t.y, gy[m][i]);
            }
    });
    // pin the finished grid HERE: otherwise the optimiser sinks the chains below the image barrier
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(gx[m][i]), "+v"(gy[m][i]));

    if (wv == 0) stamp(P.trace, 2);                           // grid expanded
    __syncthreads();                                          // the loaders' DMA has landed
    if (wv == 0) stamp(P.trace, 3);

    // ---- bilinear taps from LDS, 16-byte stores ----
    const size_t row_bytes = (size_t)P.n * 4;
    const int b = b0 + im;
    const float* img = sImg + im * img_elems;
    gchar* obase = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;           // wave-uniform
    const unsigned oimg = (unsigned)im * (unsigned)(C * row_bytes);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const Taps t = make_taps(gx[m][i], gy[m][i], H, W);
            if constexpr (AUX) {
                // pixel of (mirror m, slot i): x-mirrored groups hold the pixels in reverse order
                const unsigned go = (m == 0 ? o_a : m == 1 ? o_x : m == 2 ? o_y : o_xy) + 4u * ((m & 1) ? 3 - i : i);
                const bool st = live && (xmir || !(m & 1));
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * go) =
                        make_float2(gx[m][i], gy[m][i]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * go) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[i][0] = img + t.o00;
            tp[i][1] = t.inx ? tp[i][0] + 1 : sZero;
            tp[i][2] = t.iny ? img + t.o10 : sZero;
            tp[i][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[i][0] = t.nw; tw[i][1] = t.ne; tw[i][2] = t.sw; tw[i][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[i][ch][q] = tp[i][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
        const unsigned go = oimg + (m == 0 ? o_a : m == 1 ? o_x : m == 2 ? o_y : o_xy);
        const bool st = live && (xmir || !(m & 1));
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float res[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float acc = tv[i][ch][0] * tw[i][0];
                acc = fmaf(tv[i][ch][1], tw[i][1], acc);
                acc = fmaf(tv[i][ch][2], tw[i][2], acc);
                acc = fmaf(tv[i][ch][3], tw[i][3], acc);
                res[i] = acc;
            }
            v4f o;
            if (m & 1) { o[0] = res[3]; o[1] = res[2]; o[2] = res[1]; o[3] = res[0]; }
            else       { o[0] = res[0]; o[1] = res[1]; o[2] = res[2]; o[3] = res[3]; }
            if (st) store16<STORE>(obase, go + (unsigned)(ch * row_bytes), o);
        }
    }
    if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(P.trace, 4);
    }
}


// 4-byte store with a cache policy (same modes as store16)
template <int MODE>
__device__ __forceinline__ void store4(gchar* base, unsigned voff, float v)
{
    if constexpr (MODE == 0) {
        *reinterpret_cast<__attribute__((address_space(1))) float*>(base + voff) = v;
    } else if constexpr (MODE == 1) {
        asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(base) : "memory");
    } else if constexpr (MODE == 2) {
        asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(base) : "memory");
    } else {
        asm volatile("global_store_dword %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v), "s"(base) : "memory");
    }
}

// lab trace: 16 slots per workgroup
__device__ __forceinline__ void lstamp(long long* trace, int slot)
{
    if (trace && (threadIdx.x & (kWave - 1)) == 0)
        trace[(size_t)blockIdx.x * 16 + slot] = (long long)__builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void lwall(long long* trace, int slot)
{
    if (trace && (threadIdx.x & (kWave - 1)) == 0)
        trace[(size_t)blockIdx.x * 16 + slot] = (long long)wall_clock64();
}

// ---- "m" kernel: the production mapping (thread = one quadrant pixel x 4 mirrors x image pair) with knobs ----
// NW compute + NLOAD loader wavefronts; FIRST pieces per loader before the T barrier; STORE policy;
// DBG bit 0: no output stores, bit 1: no image DMA (timing experiments only).
template <int F, int C, int HC, int WC, int NW, int NLOAD, int FIRST, int STORE, int LDNT, int DBG>
__global__ void __launch_bounds__((NW + NLOAD) * 64)
tps_warp_m_kernel(const Q4Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;
    float* sImg = sInv + ((K * K + 3) & ~3);
    float* sZero = sImg + P.zero_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * 2;
    const bool hasB = (b0 + 1) < P.N;
    constexpr int HW = H * W;
    constexpr int img_elems = C * HW;

    if (wv == 0) { lstamp(P.trace, 0); lwall(P.trace, 8); }
    if (wv >= NW) {
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            if (!(DBG & 2))
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + off),
                    (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                    16, 0, LDNT ? 2 : 0);
        };
        int piece = wv - NW;
        for (int i = 0; i < FIRST && piece < pieces; ++i, piece += NLOAD) dma(piece);
        lds_only_barrier();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (wv == NW) lstamp(P.trace, 10);
        if (FIRST == 0 && NLOAD == 3) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); if (wv == NW) lstamp(P.trace, 12); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv == NW) lstamp(P.trace, 11);
        __syncthreads();
        return;
    }

    const int halfW = P.Wo >> 1;
    const int nq = (P.Ho >> 1) * halfW;
    const bool live = tid < nq;
    const int qp = live ? tid : nq - 1;
    const int r = qp / halfW, c = qp - r * halfW;
    unsigned poff[4];
    poff[0] = 4u * (unsigned)(r * P.Wo + c);
    poff[1] = 4u * (unsigned)(r * P.Wo + (P.Wo - 1 - c));
    poff[2] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + c);
    poff[3] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + (P.Wo - 1 - c));

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
    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }
    if (wv == 0) lstamp(P.trace, 1);                               // loads issued
    if (wv < 2) {
        asm volatile("" : "+v"(cx), "+v"(cy));
        if (wv == 0) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NINV + K) : "memory"); lstamp(P.trace, 2); }   // ctrl arrived
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    lds_only_barrier();
    if (wv == 0) lstamp(P.trace, 3);                               // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));

    if (wv == 0) lstamp(P.trace, 4);                               // grid expanded
    __syncthreads();
    if (wv == 0) lstamp(P.trace, 5);                               // images in LDS

    const size_t row_bytes = (size_t)P.n * 4;
    if (live) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            if (im == 1 && !hasB) break;
            const int b = b0 + im;
            const float* img = sImg + im * img_elems;
            gchar* oc[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                oc[ch] = (gchar*)(P.out) + ((size_t)b * C + ch) * row_bytes;
                asm volatile("" : "+s"(oc[ch]));
            }
            const float* tp[4][4];
            float tw[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
                const bool inxy = t.inx && t.iny;
                tp[m][0] = img + t.o00;
                tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
                tp[m][2] = t.iny ? img + t.o10 : sZero;
                tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
                tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
            }
            float tv[4][C][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
#pragma unroll
                    for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    float acc = tv[m][ch][0] * tw[m][0];
                    acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                    acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                    acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                    if (!(DBG & 1) || acc == 12345.678f) store4<STORE>(oc[ch], poff[m], acc);
                }
            }
        }
    }
    if (wv == 0) {
        lstamp(P.trace, 6);                                        // stores issued
        if (DBG & 4) asm volatile("buffer_wbl2 sc1" ::: "memory");
        if ((DBG & 8) && blockIdx.x >= 248) asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1" ::: "memory");
        if ((DBG & 16) && blockIdx.x < 8) asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lstamp(P.trace, 7);                                        // stores retired
        lwall(P.trace, 9);
    }
}


// ---- "m2" kernel: the production mapping with 16-byte stores ---------------------------------------
// Thread = one pixel (r, c) of the left half-row padded to whole 16-byte groups (PW = 4*ceil(Wo/8) columns:
// 52 for Wo = 100, so 16 rows x 52 = 832 threads = 13 full wavefronts) x 4 mirrors x image pair.  A lane's 24
// results (2 images x 4 mirrors x 3 channels) are 4x4-transposed inside its quad of lanes with DPP (two
// butterfly stages, quad_perm xor 1 / xor 2), after which lane j of a quad holds FOUR CONSECUTIVE PIXELS of
// one (image, mirror, channel) combination: one 16-byte store per lane instead of four 4-byte ones, which is
// what makes write-through (sc1) stores affordable -- the end-of-kernel write-back of 19.7 MB of dirty L2
// lines (2.4 us of the launch-to-launch gap) then happens during the kernel.
// A column c >= Wo/2 of the padded half-row is a real pixel with its own table row; its x-mirror duplicates
// another thread's pixel and is not stored.
__device__ __forceinline__ float dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
// a[i] of lane j  ->  a[j'] ... after the call lane j holds (in a[0..3]) the value a[j] of quad lanes 0..3
__device__ __forceinline__ void quad_transpose(float (&a)[4], bool odd, bool hi)
{
    {
        const float r0 = dpp_xor1(a[0]), r1 = dpp_xor1(a[1]);
        const float r2 = dpp_xor1(a[2]), r3 = dpp_xor1(a[3]);
        a[0] = odd ? r1 : a[0]; a[1] = odd ? a[1] : r0;
        a[2] = odd ? r3 : a[2]; a[3] = odd ? a[3] : r2;
    }
    {
        const float r0 = dpp_xor2(a[0]), r2 = dpp_xor2(a[2]);
        const float r1 = dpp_xor2(a[1]), r3 = dpp_xor2(a[3]);
        a[0] = hi ? r2 : a[0]; a[2] = hi ? a[2] : r0;
        a[1] = hi ? r3 : a[1]; a[3] = hi ? a[3] : r1;
    }
}

template <int F, int C, int HC, int WC, int NW, int NLOAD, int FIRST, int STORE, int LDNT, int DBG, bool AUX>
__global__ void __launch_bounds__((NW + NLOAD) * 64)
tps_warp_m2_kernel(const Q4Params P)
{
    static_assert(C == 3, "combination packing below is written for 3 channels");
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;
    float* sImg = sInv + ((K * K + 3) & ~3);
    float* sZero = sImg + P.zero_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * 2;
    const bool hasB = (b0 + 1) < P.N;
    constexpr int HW = H * W;
    constexpr int img_elems = C * HW;

    if (wv == 0) { lstamp(P.trace, 0); lwall(P.trace, 8); }
    if (wv >= NW) {
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            if (!(DBG & 2))
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + off),
                    (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                    16, 0, LDNT ? 2 : 0);
        };
        int piece = wv - NW;
        for (int i = 0; i < FIRST && piece < pieces; ++i, piece += NLOAD) dma(piece);
        lds_only_barrier();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (wv == NW) lstamp(P.trace, 10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv == NW) lstamp(P.trace, 11);
        __syncthreads();
        return;
    }

    const int halfW = P.Wo >> 1;
    const int PW = (halfW + 3) & ~3;                       // padded half-row, whole 16-byte groups
    const int nthr = (P.Ho >> 1) * PW;
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const int c4 = c & ~3;
    const bool xdup = c4 + 4 > halfW;                      // middle group: its x-mirror is itself
    const unsigned p0 = 4u * (unsigned)(r * P.Wo + c);     // byte offset of (r, c) in a plane (table row / AUX)

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
    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + p0);
    }
    if (wv == 0) lstamp(P.trace, 1);
    if (wv < 2) {
        if (wv == 0 && P.trace) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NINV + K) : "memory"); lstamp(P.trace, 2); }
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    lds_only_barrier();
    if (wv == 0) lstamp(P.trace, 3);

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));

    if (wv == 0) lstamp(P.trace, 4);
    __syncthreads();
    if (wv == 0) lstamp(P.trace, 5);

    // ---- taps: res[im][m][ch] ----
    const size_t row_bytes = (size_t)P.n * 4;
    float res[2][4][C];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const float* img = sImg + im * img_elems;
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const int rr = (m & 2) ? P.Ho - 1 - r : r, cc = (m & 1) ? P.Wo - 1 - c : c;
                const unsigned go = 4u * (unsigned)(rr * P.Wo + cc);
                const bool st = live && (im == 0 || hasB) && !((m & 1) && xdup);
                const int b = b0 + im;
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * go) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * go) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[m][0] = img + t.o00;
            tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
            tp[m][2] = t.iny ? img + t.o10 : sZero;
            tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[m][0];
                acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                res[im][m][ch] = acc;
            }
    }

    // ---- quad transposes + 16-byte stores: for each x-mirror state 12 combinations k = (im, ym, ch) ----
    {
        const bool odd = lane & 1, hi = lane & 2;
        const int j = lane & 3;
        gchar* obase = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;      // wave-uniform
        const unsigned col[2] = {4u * (unsigned)c4, 4u * (unsigned)(P.Wo - 4 - c4)};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            // this lane's combination in group g and its plane / row offset
            const int k = 4 * g + j;
            const int im = k / 6, ym = (k - 6 * im) / 3, ch = k - 6 * im - 3 * ym;
            const unsigned poff = (unsigned)(im * C + ch) * (unsigned)row_bytes +
                                  4u * (unsigned)((ym ? P.Ho - 1 - r : r) * P.Wo);
            const bool st_k = live && (im == 0 || hasB);
#pragma unroll
            for (int xm = 0; xm < 2; ++xm) {
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    constexpr int dummy = 0; (void)dummy;
                    const int kk = 4 * g + i;                      // compile-time after unrolling
                    const int imk = kk / 6, ymk = (kk - 6 * imk) / 3, chk = kk - 6 * imk - 3 * ymk;
                    a[i] = res[imk][2 * ymk + xm][chk];
                }
                quad_transpose(a, odd, hi);
                v4f o;
                if (xm) { o[0] = a[3]; o[1] = a[2]; o[2] = a[1]; o[3] = a[0]; }
                else    { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; }
                const bool st = st_k && !(xm && xdup);
                if (!(DBG & 1) || o[0] == 12345.678f) {
                    if (st) store16<STORE>(obase, poff + col[xm], o);
                }
            }
        }
    }
    if (wv == 0) {
        lstamp(P.trace, 6);
        if (DBG & 4) asm volatile("buffer_wbl2 sc1" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lstamp(P.trace, 7);
        lwall(P.trace, 9);
    }
}


// ---- "m3" kernel: image A / image B pipeline, output staged through LDS --------------------------------
// Same thread mapping as m2 (one pixel of the padded left half-row x 4 mirrors x image pair; 832 threads for
// 32x100).  Differences:
//   * the loaders request image A's pieces before image B's and signal them separately: A is sampled while B
//     is still landing (A arrives about 1 us before B when every CU asks for its A first);
//   * the 12 results per pixel and image are written to an image-shaped LDS buffer; after a barrier the
//     group copies that buffer to HBM as a flat array, 16 bytes per lane and 1 KB contiguous per wavefront
//     instruction: whole 128-byte lines, which is what a write-through / streaming store policy needs to be
//     efficient, so image A's output leaves the chip while image B is read.
// Trace (TRACE = true): s_memtime stamps are kept in SGPRs and written once at the end.
struct M3Params {
    const float* in; const float* ctrl; const float* inv_delta_c; const float* p_hat_t;
    int N, n, Ho, Wo;
    float* out; float* grid; int32_t* idx;
    int zero_off;       // float offset (from the staged pair) of the zero words
    int out_off;        // float offset (from the staged pair) of the output staging buffer
    long long* trace;
    long long* trace2;
};

template <int F, int C, int HC, int WC, int NLOAD, int FIRST, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m3_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;
    float* sImg = sInv + ((K * K + 3) & ~3);
    float* sZero = sImg + P.zero_off;
    float* sOut = sImg + P.out_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int NW = (int)(blockDim.x / kWave) - NLOAD;
    const int b0 = blockIdx.x * 2;
    const bool hasB = (b0 + 1) < P.N;
    constexpr int HW = H * W;
    constexpr int img_elems = C * HW;

    unsigned ts[10];
#define M3_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M3_STAMP(0);

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        int piece = lw, nB = 0;
        for (int i = 0; i < FIRST && piece < pieces; ++i, piece += NLOAD) { dma(piece); nB += piece >= piecesA; }
        lds_only_barrier();                                  // (T ready)
        for (; piece < pieces; piece += NLOAD) { dma(piece); nB += piece >= piecesA; }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (A landed)
        __builtin_amdgcn_s_barrier();                        // (A's results staged)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (B landed, staging buffer free)
        return;
    }

    const int halfW = P.Wo >> 1;
    const int PW = (halfW + 3) & ~3;
    const int nthr = (P.Ho >> 1) * PW;
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * P.Wo + c);
    poff[1] = 4u * (unsigned)(r * P.Wo + (P.Wo - 1 - c));
    poff[2] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + c);
    poff[3] = 4u * (unsigned)((P.Ho - 1 - r) * P.Wo + (P.Wo - 1 - c));

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
    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        const size_t row_bytes = (size_t)P.n * 4;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }
    M3_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        M3_STAMP(2);                                         // ctrl + inv arrived
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    lds_only_barrier();                                      // (T ready)
    M3_STAMP(3);

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));
    M3_STAMP(4);                                             // grid expanded

    const size_t row_bytes = (size_t)P.n * 4;
    const int out16 = (C * (int)P.n) >> 2;                   // 16-byte pieces of one output image
    const int nct = NW * kWave;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        lds_only_barrier();                                  // im 0: A landed; im 1: B landed + staging free
        M3_STAMP(5 + 2 * im);
        if (im == 1 && !hasB) break;
        const int b = b0 + im;
        const float* img = sImg + im * img_elems;
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const bool st = live && !((m & 1) && xdup);
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[m][0] = img + t.o00;
            tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
            tp[m][2] = t.iny ? img + t.o10 : sZero;
            tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[m][0];
                acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M3_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
        for (int e = tid + 3 * nct; e < out16; e += nct) {   // other geometries
            const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
            store16<STORE>(ob, 16u * (unsigned)e, o);
        }
    }
    if (TRACE && wv == 0) {
        M3_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M3_STAMP
}

inline size_t m3_lds_bytes(int F, int C, int H, int W, int Ho, int Wo, int* zero_off, int* out_off)
{
    const int K = F + 3;
    const int pieces = (2 * C * H * W * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    *out_off = pieces * 256 + (C - 1) * H * W + 4;
    return (size_t)(4 * K + ((K * K + 3) & ~3)) * sizeof(float) + (size_t)(*out_off) * 4 + (size_t)C * Ho * Wo * 4;
}


// ---- "m4" kernel: m3 with a dedicated T-solver wavefront and LDS flags instead of loader barriers ---------
// Wavefront roles: [0, NW) compute, [NW, NW + NLOAD) loaders, last = T-solver.  One workgroup barrier at entry
// (so that the three LDS flags are known to be zero), then
//   * loaders request image A's pieces, then image B's, without ever waiting for anyone; they bump flag A when
//     their share of A has landed (a counted vmcnt: requests retire in order) and flag B at the end;
//   * the T-solver loads the control points of both images (lanes 0..31 image A, 32..63 image B) and
//     inv_delta_C, solves T as the k-ascending FMA chain, publishes (TxA, TyA, TxB, TyB) rows and sets flag T;
//   * compute wavefronts request their table row, poll flag T, expand the grid of both images, poll flag A,
//     sample A into the staging buffer, barrier, copy the buffer out as 16-byte pieces, poll flag B, ...
// Barriers after the first are executed by the compute wavefronts only: a loader / T-solver that is still
// running holds them back until it terminates (it has nothing left to do but wait for its last DMA).
__device__ __forceinline__ void wait_flag(const float* flag_word, int want)
{
    const volatile int* f = reinterpret_cast<const volatile int*>(flag_word);
    while (*f < want) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void bump_flag(float* flag_word)
{
    // inline asm: the compiler would order a visible LDS access behind ALL outstanding LDS-DMA of this wavefront
    const unsigned a = (unsigned)(size_t)flag_word;          // LDS byte address (generic -> local: low 32 bits)
    const int one = 1;
    asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(one) : "memory");
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m4_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
    constexpr int n = OH * OW;
    static_assert(NW + NLOAD + 1 <= 16, "too many wavefronts");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;                            // K*K
    float* sCtrl = sInv + ((K * K + 3) & ~3);              // 2 x 32 x (x, y)
    float* sFlag = sCtrl + 128;                            // [0] T ready, [1] loaders done with A, [2] with B
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

    unsigned ts[10];
#define M4_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M4_STAMP(0);
    if (tid < 3) reinterpret_cast<int*>(sFlag)[tid] = 0;
    lds_only_barrier();

    if (wv >= NW + NLOAD) {
        // ================= T-solver =================
        constexpr int KK = K * K;
        constexpr int NINV = (KK + kWave - 1) / kWave;
        const int half = lane >> 5, l5 = lane & 31;
        float2 cc = make_float2(0.0f, 0.0f);
        if (l5 < F) {
            const int b = (half && hasB) ? b0 + 1 : b0;
            cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[l5];
        }
        float invv[NINV];
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
        reinterpret_cast<float2*>(sCtrl)[lane] = cc;         // rows F..31 of each half: zeros
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + (l5 < K ? l5 : K - 1) * K;
        const float2* crow = reinterpret_cast<const float2*>(sCtrl) + half * 32;
        float hv[K]; float2 cv[K];
#pragma unroll
        for (int q = 0; q < K; ++q) { hv[q] = hrow[q]; cv[q] = crow[q]; }
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            ax = fmaf(hv[q], cv[q].x, ax);
            ay = fmaf(hv[q], cv[q].y, ay);
        }
        if (l5 < K) {
            float* dst = reinterpret_cast<float*>(sT + l5) + 2 * half;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
        return;
    }
    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        int nB = 0;
        for (int piece = lw; piece < pieces; piece += NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
            nB += piece >= piecesA;
        }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        constexpr unsigned row_bytes = n * 4u;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M4_STAMP(1);                                             // loads issued
    wait_flag(sFlag + 0, 1);
    M4_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));
    M4_STAMP(4);                                             // grid expanded

    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1) {
            if (!hasB) break;
            lds_only_barrier();                              // everybody has copied image A out of the staging buffer
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M4_STAMP(5 + 2 * im);
        const int b = b0 + im;
        const float* img = sImg + im * img_elems;
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const bool st = live && !((m & 1) && xdup);
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[m][0] = img + t.o00;
            tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
            tp[m][2] = t.iny ? img + t.o10 : sZero;
            tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[m][0];
                acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M4_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M4_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M4_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m5_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M5_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M5_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        int nB = 0;
        for (int piece = lw; piece < pieces; piece += NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
            nB += piece >= piecesA;
        }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        constexpr unsigned row_bytes = n * 4u;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M5_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M5_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));
    M5_STAMP(4);                                             // grid expanded

    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1) {
            if (!hasB) break;
            lds_only_barrier();                              // everybody has copied image A out of the staging buffer
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M5_STAMP(5 + 2 * im);
        const int b = b0 + im;
        const float* img = sImg + im * img_elems;
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const bool st = live && !((m & 1) && xdup);
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[m][0] = img + t.o00;
            tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
            tp[m][2] = t.iny ? img + t.o10 : sZero;
            tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[m][0];
                acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M5_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M5_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M5_STAMP
}


template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m6_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M6_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M6_STAMP(0);
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
    lds_only_barrier();
    M6_STAMP(2);

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        int nB = 0;
        for (int piece = lw; piece < pieces; piece += NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
            nB += piece >= piecesA;
        }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    float v[K];
    {
        const char* pht = reinterpret_cast<const char*>(P.p_hat_t);
        constexpr unsigned row_bytes = n * 4u;
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = *reinterpret_cast<const float*>(pht + q * row_bytes + poff[0]);
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M6_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M6_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // tap descriptors of both images while image A is still in flight (pure VALU work)
    const float* tp[2][4][4];
    float tw[2][4][4];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const float* img = sImg + im * img_elems;
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
            const bool inxy = t.inx && t.iny;
            tp[im][m][0] = img + t.o00;
            tp[im][m][1] = t.inx ? tp[im][m][0] + 1 : sZero;
            tp[im][m][2] = t.iny ? img + t.o10 : sZero;
            tp[im][m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[im][m][0] = t.nw; tw[im][m][1] = t.ne; tw[im][m][2] = t.sw; tw[im][m][3] = t.se;
        }
    }
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(tp[im][m][q]), "+v"(tw[im][m][q]));
    M6_STAMP(4);                                             // grid + tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1) {
            if (!hasB) break;
            lds_only_barrier();                              // everybody has copied image A out of the staging buffer
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M6_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[im][m][q][ch * HW];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[im][m][0];
                acc = fmaf(tv[m][ch][1], tw[im][m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[im][m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[im][m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M6_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M6_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M6_STAMP
}


template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m7_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M7_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M7_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        int nB = 0;
        for (int piece = lw; piece < pieces; piece += NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
            nB += piece >= piecesA;
        }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M7_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M7_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][0] = fmaf(val[m], t.x, gx[m][0]);
            gy[m][0] = fmaf(val[m], t.y, gy[m][0]);
            gx[m][1] = fmaf(val[m], t.z, gx[m][1]);
            gy[m][1] = fmaf(val[m], t.w, gy[m][1]);
        }
    });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]), "+v"(gx[m][1]), "+v"(gy[m][1]));
    M7_STAMP(4);                                             // grid expanded

    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1) {
            if (!hasB) break;
            lds_only_barrier();                              // everybody has copied image A out of the staging buffer
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M7_STAMP(5 + 2 * im);
        const int b = b0 + im;
        const float* img = sImg + im * img_elems;
        const float* tp[4][4];
        float tw[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const Taps t = make_taps(gx[m][im], gy[m][im], H, W);
            if constexpr (AUX) {
                const bool st = live && !((m & 1) && xdup);
                if (P.grid && st)
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_float2(gx[m][im], gy[m][im]);
                if (P.idx && st)
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * poff[m]) =
                        make_int2(t.x0, t.y0);
            }
            const bool inxy = t.inx && t.iny;
            tp[m][0] = img + t.o00;
            tp[m][1] = t.inx ? tp[m][0] + 1 : sZero;
            tp[m][2] = t.iny ? img + t.o10 : sZero;
            tp[m][3] = inxy ? img + t.o10 + 1 : sZero;
            tw[m][0] = t.nw; tw[m][1] = t.ne; tw[m][2] = t.sw; tw[m][3] = t.se;
        }
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = tp[m][q][ch * HW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[m][0];
                acc = fmaf(tv[m][ch][1], tw[m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M7_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M7_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M7_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE>
__global__ void __launch_bounds__(1024)
tps_warp_m8_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M8_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M8_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        int nB = 0;
        for (int piece = lw; piece < pieces; piece += NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
            nB += piece >= piecesA;
        }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M8_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M8_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M8_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1) {
            if (!hasB) break;
            lds_only_barrier();                              // everybody has copied image A out of the staging buffer
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M8_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = tv[m][ch][0] * tw[im][m][0];
                acc = fmaf(tv[m][ch][1], tw[im][m][1], acc);
                acc = fmaf(tv[m][ch][2], tw[im][m][2], acc);
                acc = fmaf(tv[m][ch][3], tw[im][m][3], acc);
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = acc;
            }
        lds_only_barrier();                                  // results of image `im` staged
        M8_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M8_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M8_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT>
__global__ void __launch_bounds__(1024)
tps_warp_m9_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M9_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M9_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        int nB = 0, piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        // image B's requests follow when all but AWAIT of this loader's requests for A have been served: every CU's A
        // then travels ahead of every CU's B and lands about a microsecond earlier
        if (AWAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (AWAIT < 13) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT < 13 ? AWAIT : 0) : "memory");
        for (; piece < pieces; piece += NLOAD) { dma(piece); ++nB; }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        unsigned tA = 0, tB = 0;
        if (TRACE) tA = (unsigned)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) { tB = (unsigned)__builtin_amdgcn_s_memtime(); if (lw == 0 && lane == 0 && P.trace) { P.trace[(size_t)blockIdx.x * 16 + 14] = (long long)(tA - ts[0]); P.trace[(size_t)blockIdx.x * 16 + 15] = (long long)(tB - ts[0]); } }
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int qp = live ? tid : nthr - 1;
    const int r = qp / PW, c = qp - r * PW;
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M9_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M9_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M9_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) break;
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M9_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
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
        if (im == 1) lds_only_barrier();                     // everybody has copied image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        M9_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M9_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M9_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT>
__global__ void __launch_bounds__(1024)
tps_warp_m10_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M10_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M10_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        int nB = 0, piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        if (AWAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (AWAIT < 13) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT < 13 ? AWAIT : 0) : "memory");
        for (; piece < pieces; piece += NLOAD) { dma(piece); ++nB; }
        if (TRACE) t_ib = now();
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M10_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M10_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M10_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) break;
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M10_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
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
        if (im == 1) lds_only_barrier();                     // everybody has copied image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        M10_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M10_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M10_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB>
__global__ void __launch_bounds__(1024)
tps_warp_m13_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M13_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M13_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        // Image A first.  Image B's requests start when at most AWAIT of this loader's requests for A are still
        // outstanding (HBM does not serve requests in arrival order: with B's queued behind them A would land together
        // with B), KB of them are issued, then A is complete once at most KB requests remain (vmcnt retires in order):
        // flag A is raised before the bulk of B's requests -- which take ~1 us to issue against HBM back-pressure.
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT) : "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB) : "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (TRACE) t_ib = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M13_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M13_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M13_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) break;
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M13_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
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
        if (im == 1) lds_only_barrier();                     // everybody has copied image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        M13_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M13_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M13_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB>
__global__ void __launch_bounds__(1024)
tps_warp_m14_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M14_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M14_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        // Image A first.  Image B's requests start when at most AWAIT of this loader's requests for A are still
        // outstanding (HBM does not serve requests in arrival order: with B's queued behind them A would land together
        // with B), KB of them are issued, then A is complete once at most KB requests remain (vmcnt retires in order):
        // flag A is raised before the bulk of B's requests -- which take ~1 us to issue against HBM back-pressure.
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT) : "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB) : "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (TRACE) t_ib = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M14_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M14_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M14_STAMP(4);                                             // grid + image A's tap descriptors done

    constexpr int NOUT = (out16 + nct - 1) / nct;            // 16-byte output pieces per thread and image
    v4f ostage[NOUT];                                        // image A's pieces between their LDS read and their store
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) {
            // no image B: image A's pieces leave now
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
            break;
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M14_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // image A's output pieces (read from the staging buffer before this point) are stored while the LDS
            // serves image B's tap reads: the store issue (back-pressured by HBM) is off the critical path
            __builtin_amdgcn_sched_barrier(0);
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
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
        M14_STAMP(6 + 2 * im);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) ostage[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
        }
        if (im == 1) {
            gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(ostage[i]));
        }
    }
    if (TRACE && wv == 0) {
        M14_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M14_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB>
__global__ void __launch_bounds__(1024)
tps_warp_m15_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M15_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M15_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        // Image A first.  Image B's requests start when at most AWAIT of this loader's requests for A are still
        // outstanding (HBM does not serve requests in arrival order: with B's queued behind them A would land together
        // with B), KB of them are issued, then A is complete once at most KB requests remain (vmcnt retires in order):
        // flag A is raised before the bulk of B's requests -- which take ~1 us to issue against HBM back-pressure.
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT) : "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB) : "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (TRACE) t_ib = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M15_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M15_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    describe(std::integral_constant<int, 1>{});              // both before image A lands: the wavefront would idle otherwise
    M15_STAMP(4);                                             // grid + tap descriptors done

    constexpr int NOUT = (out16 + nct - 1) / nct;            // 16-byte output pieces per thread and image
    v4f ostage[NOUT];                                        // image A's pieces between their LDS read and their store
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) {
            // no image B: image A's pieces leave now
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
            break;
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M15_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 1) {
            // image A's output pieces (read from the staging buffer before this point) are stored while the LDS
            // serves image B's tap reads: the store issue (back-pressured by HBM) is off the critical path
            __builtin_amdgcn_sched_barrier(0);
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
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
        M15_STAMP(6 + 2 * im);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) ostage[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
        }
        if (im == 1) {
            gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(ostage[i]));
        }
    }
    if (TRACE && wv == 0) {
        M15_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M15_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB>
__global__ void __launch_bounds__(1024)
tps_warp_m18_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
    constexpr int n = OH * OW;
    static_assert(NW + NLOAD <= 16, "too many wavefronts");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* sT = reinterpret_cast<float4*>(smem);          // K x (TxA, TyA, TxB, TyB)
    float* sInv = smem + 4 * K;                            // 2 x K*K (one copy per solving wavefront)
    float* sFlag = sInv + 2 * ((K * K + 3) & ~3);          // [0] T rows published, [1 + q] loaders done with plane q of the pair
    float* sImg = sFlag + 8;
    float* sZero = sImg + P.zero_off;
    float* sOut = sImg + P.out_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * 2;
    const bool hasB = (b0 + 1) < P.N;
    constexpr int HW = H * W;
    constexpr int img_elems = C * HW;

    unsigned ts[10];
#define M18_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M18_STAMP(0);
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
    if (tid < 8) reinterpret_cast<int*>(sFlag)[tid] = 0;
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fq = (unsigned)(size_t)(sFlag + 1);     // flag operands in registers BEFORE the first DMA (see m13)
        int one = 1;
        asm volatile("" : "+v"(fq), "+v"(one));
        // wait until at most n of this wavefront's requests are outstanding (vmcnt retires in order; literal operands)
        auto wait_le = [&](int n) {
            switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        };
        auto raise = [&](int q) {      // flag word 1 + q: the plane index goes into the instruction's offset field
            if (lane == 0) {
                switch (q) {
                case 0: asm volatile("ds_add_u32 %0, %1 offset:0" ::"v"(fq), "v"(one) : "memory"); break;
                case 1: asm volatile("ds_add_u32 %0, %1 offset:4" ::"v"(fq), "v"(one) : "memory"); break;
                case 2: asm volatile("ds_add_u32 %0, %1 offset:8" ::"v"(fq), "v"(one) : "memory"); break;
                case 3: asm volatile("ds_add_u32 %0, %1 offset:12" ::"v"(fq), "v"(one) : "memory"); break;
                case 4: asm volatile("ds_add_u32 %0, %1 offset:16" ::"v"(fq), "v"(one) : "memory"); break;
                default: asm volatile("ds_add_u32 %0, %1 offset:20" ::"v"(fq), "v"(one) : "memory"); break;
                }
            }
        };
        // pieces of this loader with index <= x
        auto upto = [&](int x) { return x >= lw ? (x - lw) / NLOAD + 1 : 0; };
        constexpr int plane_bytes = HW * 4;
        auto last_piece = [&](int q) { return ((q + 1) * plane_bytes + 1023) / 1024 - 1; };   // plane q complete when pieces <= this landed
        const int nA = upto(piecesA - 1);
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        int issuedB = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            if (q == C - 1) {
                // the last plane of image A: let the first KB requests of image B go out first (cf. m13)
#pragma unroll
                for (int i2 = 0; i2 < KB; ++i2) { if (piece < pieces) { dma(piece); ++issuedB; } piece += NLOAD; }
            }
            wait_le(nA - upto(last_piece(q)) + issuedB);
            raise(q);
        }
        for (; piece < pieces; piece += NLOAD) { dma(piece); ++issuedB; }
        const int firstB = upto(piecesA - 1);     // pieces before image B's first own piece
#pragma unroll
        for (int q = C; q < 2 * C; ++q) {
            const int doneB = upto(last_piece(q) < pieces ? last_piece(q) : pieces - 1) - firstB;   // of this loader's B pieces
            wait_le(issuedB - (doneB > 0 ? doneB : 0));
            raise(q);
        }
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M18_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M18_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    // one k-step of the 8 chains of image `im` (4 mirror pixels x (x, y)); table values from registers
    auto chain_step = [&](auto qc, auto imc) {
        constexpr int q = decltype(qc)::value;
        constexpr int im = decltype(imc)::value;
        const float2 t = reinterpret_cast<const float2*>(sT + q)[im];
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
        for (int m = 0; m < 4; ++m) {
            gx[m][im] = fmaf(val[m], t.x, gx[m][im]);
            gy[m][im] = fmaf(val[m], t.y, gy[m][im]);
        }
    };
    static_for<K>([&](auto qc) { chain_step(qc, std::integral_constant<int, 0>{}); });
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[m][0]), "+v"(gy[m][0]));
    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = n >> 2;                            // 16-byte pieces of one output PLANE
    constexpr int nct = NW * kWave;
    static_assert(out16 <= nct, "one 16-byte piece per thread and plane");
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
    M18_STAMP(4);                                            // image A: grid + tap descriptors done

    // ---- plane pipeline: unit = one channel plane of one image (its taps need that plane only) -------------------
    // plane q = (image q / C, channel q % C) is sampled as soon as the loaders have raised flag q; image B's grid chains
    // (pure VALU) are slotted in behind the tap reads of image A's planes, whose phases are LDS-bound.
    v4f opiece = {0, 0, 0, 0};                               // the 16-byte piece this thread copies out, one plane behind
    auto store_piece = [&](int q) {
        if (tid < out16) store16<STORE>((gchar*)(P.out) + ((size_t)b0 * C + q) * row_bytes, 16u * (unsigned)tid, opiece);
    };
    constexpr int KPART = (K + C - 1) / C;                   // image B's k-steps per plane of image A
    static_for<2 * C>([&](auto qcst) {
        constexpr int q = decltype(qcst)::value;
        constexpr int im = q / C, ch = q % C;
        if (im == 1 && !hasB) return;
        if constexpr (q == C) describe(std::integral_constant<int, 1>{});
        wait_flag(sFlag + 1 + q, NLOAD);
        if constexpr (q == 0) M18_STAMP(5);
        if constexpr (q == C) M18_STAMP(7);
        float tv[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) tv[m][t4] = *((lds_cfloat*)(size_t)(ta[im][m][t4]) + ch * HW);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q > 0) store_piece(q - 1);             // the previous plane's output leaves behind these reads
        if constexpr (im == 0) {                             // a third of image B's chains
            static_for<KPART>([&](auto jc) {
                constexpr int kq = ch * KPART + decltype(jc)::value;
                if constexpr (kq < K) chain_step(std::integral_constant<int, kq>{}, std::integral_constant<int, 1>{});
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        float* stage = sOut + (q & 1) * n;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float acc = tv[m][0] * tw[im][m][0];
            acc = fmaf(tv[m][1], tw[im][m][1], acc);
            acc = fmaf(tv[m][2], tw[im][m][2], acc);
            acc = fmaf(tv[m][3], tw[im][m][3], acc);
            if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(stage) + poff[m]) = acc;
        }
        lds_only_barrier();                                  // plane q staged (double-buffered: one barrier per plane)
        if constexpr (q == C - 1) M18_STAMP(6);
        if constexpr (q == 2 * C - 1) M18_STAMP(8);
        if (tid < out16) opiece = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(stage) + 16 * tid);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(opiece));
    });
    store_piece(hasB ? 2 * C - 1 : C - 1);
    if (TRACE && wv == 0) {
        M18_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M18_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB>
__global__ void __launch_bounds__(1024)
tps_warp_m17_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M17_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M17_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        // Image A first.  Image B's requests start when at most AWAIT of this loader's requests for A are still
        // outstanding (HBM does not serve requests in arrival order: with B's queued behind them A would land together
        // with B), KB of them are issued, then A is complete once at most KB requests remain (vmcnt retires in order):
        // flag A is raised before the bulk of B's requests -- which take ~1 us to issue against HBM back-pressure.
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT) : "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB) : "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        for (; piece < pieces; piece += NLOAD) dma(piece);
        if (TRACE) t_ib = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M17_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M17_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    describe(std::integral_constant<int, 1>{});              // both before image A lands: the wavefront would idle otherwise
    M17_STAMP(4);                                             // grid + tap descriptors done

    constexpr int NOUT = (out16 + nct - 1) / nct;            // 16-byte output pieces per thread and image
    v4f ostage[NOUT];                                        // image A's pieces between their LDS read and their store
    // Staging layout [mirror][channel][thread]: a wavefront's 64 results of one (mirror, channel) are 256 contiguous
    // bytes, written with ds_write_addtid_b32 (address = M0 + offset + 4 * lane: 2 LDS cycles instead of 4).  The
    // 16-byte output piece e = (channel, row, group of 4 columns) is then 4 consecutive threads of one block:
    // (row, 4 g .. 4 g + 3) for the left half-row and the middle group, the x-mirror's threads in reverse order otherwise.
    static_assert(nthr == nct, "thread-ordered staging needs whole wavefronts");
    unsigned oaddr[NOUT];
    bool orev[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        constexpr int G = OW / 4, CGc = PW / 4;
        const int e0 = tid + i * nct, e = e0 < out16 ? e0 : out16 - 1;
        const int ch = e / (OH * G), rem = e - ch * (OH * G), row = rem / G, g = rem - row * G;
        const bool xm = g >= CGc, ym = row >= OH / 2;
        const int rr = ym ? OH - 1 - row : row, c4 = xm ? 4 * (G - 1 - g) : 4 * g;
        const int t = ((rr >> 3) * CGc + (c4 >> 2)) * 32 + (rr & 7) * 4;
        oaddr[i] = (unsigned)(size_t)sOut + 4u * (unsigned)(((2 * (int)ym + (int)xm) * C + ch) * nct + t);
        orev[i] = xm;
    }
    const unsigned m0_stage = (unsigned)(size_t)sOut + (unsigned)wv * 256u;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) {
            // no image B: image A's pieces leave now
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
            break;
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M17_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 1) {
            // image A's output pieces (read from the staging buffer before this point) are stored while the LDS
            // serves image B's tap reads: the store issue (back-pressured by HBM) is off the critical path
            __builtin_amdgcn_sched_barrier(0);
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
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
        static_assert(C == 3, "the staging block below is written out for 3 channels");
        asm volatile("s_mov_b32 m0, %12\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:%13\n\tds_write_addtid_b32 %1 offset:%14\n\tds_write_addtid_b32 %2 offset:%15\n\t"
                     "ds_write_addtid_b32 %3 offset:%16\n\tds_write_addtid_b32 %4 offset:%17\n\tds_write_addtid_b32 %5 offset:%18\n\t"
                     "ds_write_addtid_b32 %6 offset:%19\n\tds_write_addtid_b32 %7 offset:%20\n\tds_write_addtid_b32 %8 offset:%21\n\t"
                     "ds_write_addtid_b32 %9 offset:%22\n\tds_write_addtid_b32 %10 offset:%23\n\tds_write_addtid_b32 %11 offset:%24"
                     ::"v"(res[0][0]), "v"(res[0][1]), "v"(res[0][2]), "v"(res[1][0]), "v"(res[1][1]), "v"(res[1][2]),
                       "v"(res[2][0]), "v"(res[2][1]), "v"(res[2][2]), "v"(res[3][0]), "v"(res[3][1]), "v"(res[3][2]),
                       "s"(m0_stage),
                       "n"(0 * nct * 4), "n"(1 * nct * 4), "n"(2 * nct * 4), "n"(3 * nct * 4), "n"(4 * nct * 4), "n"(5 * nct * 4),
                       "n"(6 * nct * 4), "n"(7 * nct * 4), "n"(8 * nct * 4), "n"(9 * nct * 4), "n"(10 * nct * 4), "n"(11 * nct * 4)
                     : "memory");
        lds_only_barrier();                                  // results of image `im` staged
        M17_STAMP(6 + 2 * im);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            typedef __attribute__((address_space(3))) const v4f lds_cv4;
            const v4f x = *((lds_cv4*)(size_t)oaddr[i]);
            v4f y;
            y[0] = orev[i] ? x[3] : x[0]; y[1] = orev[i] ? x[2] : x[1]; y[2] = orev[i] ? x[1] : x[2]; y[3] = orev[i] ? x[0] : x[3];
            ostage[i] = y;
        }
        if (im == 1) {
            gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(ostage[i]));
        }
    }
    if (TRACE && wv == 0) {
        M17_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M17_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT, int KB, int BMODE>
__global__ void __launch_bounds__(1024)
tps_warp_m16_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M16_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M16_STAMP(0);
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
    if (tid < 4) reinterpret_cast<int*>(sFlag)[tid] = 0;
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        auto now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return (unsigned)t; };
        unsigned t_l0 = 0, t_ia = 0, t_ib = 0, t_a = 0, t_b = 0;
        // flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler wait for vmcnt(0) first -- which would turn "A has landed" into
        // "everything has landed"
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if (TRACE) t_l0 = now();
        // Image A first.  Image B's requests start when at most AWAIT of this loader's requests for A are still
        // outstanding (HBM does not serve requests in arrival order: with B's queued behind them A would land together
        // with B), KB of them are issued, then A is complete once at most KB requests remain (vmcnt retires in order):
        // flag A is raised before the bulk of B's requests -- which take ~1 us to issue against HBM back-pressure.
        int piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        if (TRACE) t_ia = now();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT) : "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) { if (piece < pieces) dma(piece); piece += NLOAD; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB) : "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        if (TRACE) t_a = now();
        if (BMODE == 0) {
            for (; piece < pieces; piece += NLOAD) dma(piece);
            if (TRACE) t_ib = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (BMODE == 2) {
            // diagnostic: image B's requests only after image A has been sampled (flag word 3, raised by wavefront 0)
            {
                const unsigned f3 = (unsigned)(size_t)(sFlag + 3);
                int got = 0;
                while (got < 1) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(f3) : "memory"); __builtin_amdgcn_s_sleep(2); }
            }
            for (; piece < pieces; piece += NLOAD) dma(piece);
            if (TRACE) t_ib = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // image B through registers: 16-byte nt loads, then ds_write_b128 (inline asm: the LDS traffic of an
            // LDS-DMA piece seems to cost the LDS far more cycles than a 16-byte-per-lane write)
            constexpr int MAXB = 16;
            v4f rb[MAXB];
            int nb = 0;
#pragma unroll
            for (int j = 0; j < MAXB; ++j) {
                const int pc = piece + j * NLOAD;
                if (pc < pieces) {
                    int off = pc * 1024 + lane * 16;
                    if (off >= total_bytes) off = 0;
                    const char* g = src + off;
                    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(rb[j]) : "v"(g) : "memory");
                    ++nb;
                }
            }
            if (TRACE) t_ib = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < MAXB; ++j) {
                const int pc = piece + j * NLOAD;
                if (pc < pieces) {
                    const unsigned l = (unsigned)(size_t)sImg + (unsigned)pc * 1024u + (unsigned)lane * 16u;
                    asm volatile("ds_write_b128 %0, %1" ::"v"(l), "v"(rb[j]) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (TRACE) {
            t_b = now();
            if (lw == 0 && lane == 0 && P.trace) {
                long long* t = P.trace + (size_t)blockIdx.x * 16;
                // relative to this loader's first stamp after the entry barrier; slot 11 = that stamp relative to the kernel-entry stamp
                t[11] = (long long)(t_l0 - ts[0]); t[14] = (long long)(t_a - ts[0]); t[15] = (long long)(t_b - ts[0]);
                P.trace2[(size_t)blockIdx.x * 2] = (long long)(t_ia - ts[0]); P.trace2[(size_t)blockIdx.x * 2 + 1] = (long long)(t_ib - ts[0]);
            }
        }
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M16_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M16_STAMP(3);                                             // T ready

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    describe(std::integral_constant<int, 1>{});              // both before image A lands: the wavefront would idle otherwise
    M16_STAMP(4);                                             // grid + tap descriptors done

    constexpr int NOUT = (out16 + nct - 1) / nct;            // 16-byte output pieces per thread and image
    v4f ostage[NOUT];                                        // image A's pieces between their LDS read and their store
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) {
            // no image B: image A's pieces leave now
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
            break;
        }
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M16_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 1) {
            // image A's output pieces (read from the staging buffer before this point) are stored while the LDS
            // serves image B's tap reads: the store issue (back-pressured by HBM) is off the critical path
            __builtin_amdgcn_sched_barrier(0);
            gchar* ob = (gchar*)(P.out) + (size_t)b0 * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
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
        M16_STAMP(6 + 2 * im);
        if (BMODE == 2 && im == 0 && tid == 0) bump_flag(sFlag + 3);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) ostage[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
        }
        if (im == 1) {
            gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) { const int e = tid + i * nct; if (e < out16) store16<STORE>(ob, 16u * (unsigned)e, ostage[i]); }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(ostage[i]));
        }
    }
    if (TRACE && wv == 0) {
        M16_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M16_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT>
__global__ void __launch_bounds__(1024)
tps_warp_m12_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M12_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M12_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        int nB = 0, piece = lw;
        for (; piece < piecesA; piece += NLOAD) dma(piece);
        // image B's requests follow when all but AWAIT of this loader's requests for A have been served: every CU's A
        // then travels ahead of every CU's B and lands about a microsecond earlier
        if (AWAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (AWAIT < 13) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AWAIT < 13 ? AWAIT : 0) : "memory");
        for (; piece < pieces; piece += NLOAD) { dma(piece); ++nB; }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        unsigned tA = 0, tB = 0;
        if (TRACE) tA = (unsigned)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) { tB = (unsigned)__builtin_amdgcn_s_memtime(); if (lw == 0 && lane == 0 && P.trace) { P.trace[(size_t)blockIdx.x * 16 + 14] = (long long)(tA - ts[0]); P.trace[(size_t)blockIdx.x * 16 + 15] = (long long)(tB - ts[0]); } }
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    if (tid < C) sZero[tid * HW] = 0.0f;
    M12_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M12_STAMP(3);                                             // T ready
    // The table rows are requested only now: 80 KB per group from the SAME lines of every XCD's L2 (32 groups per
    // XCD ask for them at the same moment: ~1 us of L2 channel time).  Requested at kernel entry they delay image A's
    // DMA by that microsecond; requested here they overlap the HBM latency of the DMA.
    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            v4f x = {0.1f, 0.2f, 0.3f, 0.4f};
            if (AWAIT != 100) x = pk[j * kWave];
            else asm volatile("" : "+v"(x));
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M12_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) break;
        wait_flag(sFlag + 1 + im, NLOAD);                    // image `im` has landed
        M12_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
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
        if (im == 1) lds_only_barrier();                     // everybody has copied image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        M12_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M12_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M12_STAMP
}

template <int F, int C, int HC, int WC, int OH, int OW, int NLOAD, int STORE, int LDNT, bool AUX, bool TRACE, int AWAIT>
__global__ void __launch_bounds__(1024)
tps_warp_m11_kernel(const M3Params P)
{
    constexpr int K = F + 3;
    constexpr int H = HC, W = WC;
    constexpr int halfW = OW / 2, PW = (halfW + 3) & ~3, nthr = (OH / 2) * PW;
    constexpr int NW = (nthr + kWave - 1) / kWave;
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

    unsigned ts[10];
#define M11_STAMP(i) do { if (TRACE) ts[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
    long long wall0 = 0;
    if (TRACE) { wall0 = (long long)wall_clock64(); }
    M11_STAMP(0);
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
    lds_only_barrier();

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int piecesA = (img_elems * 4 + 1023) >> 10;   // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, LDNT ? 2 : 0);
        };
        // image A's pieces are shared by ALL wavefronts of the group (a wavefront needs ~50 ns per DMA instruction:
        // three loaders alone spend 0.65 us issuing A and 1.7 us issuing the pair); the loaders add image B's
        int nB = 0;
        for (int piece = wv; piece < piecesA; piece += NW + NLOAD) dma(piece);
        int piece = piecesA + lw;
        for (; piece < pieces; piece += NLOAD) { dma(piece); ++nB; }
        // image A: everything but this loader's last nB requests (vmcnt retires in order)
        if (nB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (nB == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        else if (nB == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 1);
        unsigned tA = 0, tB = 0;
        if (TRACE) tA = (unsigned)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE) { tB = (unsigned)__builtin_amdgcn_s_memtime(); if (lw == 0 && lane == 0 && P.trace) { P.trace[(size_t)blockIdx.x * 16 + 14] = (long long)(tA - ts[0]); P.trace[(size_t)blockIdx.x * 16 + 15] = (long long)(tB - ts[0]); } }
        if (lane == 0) bump_flag(sFlag + 2);
        return;
    }

    // ================= compute wavefronts =================
    // their share of image A's DMA, as inline asm: the compiler must not know that this wavefront has LDS-DMA in
    // flight (it would make every LDS access below wait for it).  The table loads that follow are younger, so the
    // compiler's own wait for them covers these requests (vmcnt retires in order); flag A is bumped after that wait.
    {
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int piecesA = (img_elems * 4 + 1023) >> 10;
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        for (int piece = wv; piece < piecesA; piece += NW + NLOAD) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;
            const char* g = src + off;
            const unsigned l = (unsigned)(size_t)sImg + (unsigned)piece * 1024u;
            if (LDNT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(l) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
        }
    }
    const bool live = tid < nthr;
    // Thread -> pixel: a half-wavefront (the 32 lanes one ds_read_b32 cycle serves) owns a block of 4 columns x 8 rows.
    // With a row pitch of W = 100 floats (bank shift 4 per row) such a block touches 32 different LDS banks wherever
    // the warp is locally a translation; 32 consecutive pixels of a row pair would collide where they wrap into the
    // next row (a third of all LDS cycles were bank conflicts with the row-major mapping).
    static_assert(PW % 4 == 0 && (OH / 2) % 8 == 0, "block mapping needs whole 4 x 8 blocks");
    const int qp = live ? tid : nthr - 1;
    const int hw = qp >> 5, l5 = qp & 31;
    constexpr int CG = PW / 4;                               // column groups
    const int rg = hw / CG, cg = hw - rg * CG;
    const int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3);
    const bool xdup = (c & ~3) + 4 > halfW;
    unsigned poff[4];                                        // byte offsets of the 4 pixels in a plane
    poff[0] = 4u * (unsigned)(r * OW + c);
    poff[1] = 4u * (unsigned)(r * OW + (OW - 1 - c));
    poff[2] = 4u * (unsigned)((OH - 1 - r) * OW + c);
    poff[3] = 4u * (unsigned)((OH - 1 - r) * OW + (OW - 1 - c));

    // packed table: [wavefront][K/4 groups][lane] x 16 bytes = this thread's K values (padded to a multiple of 4)
    // as KG coalesced 16-byte loads (P.p_hat_t points at the packed copy)
    constexpr int KG = (K + 3) / 4;
    float v[KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.p_hat_t) + (size_t)wv * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            const v4f x = pk[j * kWave];
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    }
    if (tid < C) sZero[tid * HW] = 0.0f;
    M11_STAMP(1);                                             // loads issued
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * ((KK + 3) & ~3) + e] = invv[i];     // a private copy per solving wavefront
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * ((KK + 3) & ~3) + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) {
            float* dst = reinterpret_cast<float*>(sT + lane) + 2 * wv;
            dst[0] = ax; dst[1] = ay;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump_flag(sFlag + 0);
    }
    wait_flag(sFlag + 0, 2);
    M11_STAMP(3);                                             // T ready
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own share of image A (and the table rows) has landed
    if (lane == 0) bump_flag(sFlag + 1);

    float gx[4][2], gy[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) gx[m][0] = gx[m][1] = gy[m][0] = gy[m][1] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 t = sT[q];
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
    // Tap descriptors (4 LDS byte addresses + 4 weights per mirror pixel) are pure VALU work: image A's are built
    // while A is still in flight, image B's between the issue of A's tap reads and their use.
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
    M11_STAMP(4);                                             // grid + image A's tap descriptors done

#pragma unroll
    for (int im = 0; im < 2; ++im) {
        if (im == 1 && !hasB) break;
        wait_flag(sFlag + 1 + im, im == 0 ? NW + NLOAD : NLOAD);                    // image `im` has landed
        M11_STAMP(5 + 2 * im);
        const int b = b0 + im;
        float tv[4][C][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
#pragma unroll
                for (int q = 0; q < 4; ++q) tv[m][ch][q] = *((lds_cfloat*)(ta[im][m][q]) + ch * HW);
        if (im == 0) {
            __builtin_amdgcn_sched_barrier(0);
            describe(std::integral_constant<int, 1>{});      // overlaps the LDS service time of A's 48 reads
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
        if (im == 1) lds_only_barrier();                     // everybody has copied image A out of the staging buffer
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if (live) *reinterpret_cast<float*>(reinterpret_cast<char*>(sOut) + ch * (int)row_bytes + poff[m]) = res[m][ch];
        lds_only_barrier();                                  // results of image `im` staged
        M11_STAMP(6 + 2 * im);
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;          // wave-uniform
#pragma unroll
        for (int i = 0; i < (out16 + nct - 1) / nct; ++i) {
            const int e = tid + i * nct;
            if (e < out16) {
                const v4f o = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sOut) + 16 * e);
                store16<STORE>(ob, 16u * (unsigned)e, o);
            }
        }
    }
    if (TRACE && wv == 0) {
        M11_STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tend = (unsigned)__builtin_amdgcn_s_memtime();
        const long long wall1 = (long long)wall_clock64();
        if (lane == 0 && P.trace) {
            long long* t = P.trace + (size_t)blockIdx.x * 16;
            ts[2] = ts[1];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (long long)(ts[i] - ts[0]);
            t[10] = (long long)(tend - ts[0]);
            t[12] = wall0; t[13] = wall1;
        }
    }
#undef M11_STAMP
}



inline size_t m5_lds_bytes(int F, int C, int H, int W, int Ho, int Wo, int* zero_off, int* out_off)
{
    const int K = F + 3;
    const int pieces = (2 * C * H * W * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    *out_off = pieces * 256 + (C - 1) * H * W + 4;
    return (size_t)(4 * K + 2 * ((K * K + 3) & ~3) + 4) * sizeof(float) + (size_t)(*out_off) * 4 + (size_t)C * Ho * Wo * 4;
}

inline size_t m4_lds_bytes(int F, int C, int H, int W, int Ho, int Wo, int* zero_off, int* out_off)
{
    const int K = F + 3;
    const int pieces = (2 * C * H * W * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    *out_off = pieces * 256 + (C - 1) * H * W + 4;
    return (size_t)(4 * K + ((K * K + 3) & ~3) + 128 + 4) * sizeof(float) + (size_t)(*out_off) * 4 + (size_t)C * Ho * Wo * 4;
}

inline size_t m_lds_bytes(int F, int C, int H, int W, int* zero_off)
{
    const int K = F + 3;
    const int pieces = (2 * C * H * W * 4 + 1023) / 1024;
    *zero_off = pieces * 256;
    return (size_t)(4 * K + ((K * K + 3) & ~3)) * sizeof(float) + (size_t)pieces * 1024 +
           ((size_t)(C - 1) * H * W + 4) * sizeof(float);
}

// LDS bytes of one workgroup
inline size_t q4_lds_bytes(int F, int C, int H, int W, int imgs, int* zero_off)
{
    const int K = F + 3;
    const int pieces = (imgs * C * H * W * 4 + 1023) / 1024;
    const size_t head = (size_t)(2 * imgs * K + (2 * imgs * K & 2) + ((K * K + 3) & ~3)) * sizeof(float);
    *zero_off = pieces * 256;
    return head + (size_t)pieces * 1024 + ((size_t)(C - 1) * H * W + 4) * sizeof(float);
}

}  // namespace tpspp_q4
