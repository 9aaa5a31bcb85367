// Classic-geometry warp with results staged in place of the consumed input planes (round 3), instantiated per
// geometry: 32x128, 32x160, 48x160, 32x64 (and 32x100, where the image-pair kernel of tpspp_warp_pair.h -- the one
// bench.py measures -- is the production path).  tpspp_warp_geo.h is the same kernel with run-time geometry.
//
// Replaces, bit for bit: preprocessor/tps_preprocessor.py:71-83 + 270-282 (GridGenerator.build_P_prime: two bmm,
// then F.grid_sample bilinear / border / align_corners) for a mirror-symmetric RBF table (see tpspp_warp.hip).
//
// Why another kernel.  The image-pair kernel (tpspp_warp_pair.h) needs 146 KB of LDS and all 2048 wavefront
// registers of a CU: one workgroup per CU, so a launch cannot start before the previous one has drained from its
// CU, and it exists for one geometry.  Here
//   * results are staged IN PLACE of the image they were sampled from (every tap of every channel of that image is
//     in registers before the first result is written: one barrier), so the separate staging buffer disappears and a
//     workgroup is IMGS x ~38 KB of LDS;
//   * the register budget is half a CU (IMGS = 2: one quadrant pixel per thread, 13 + 3 wavefronts x 64 registers;
//     IMGS = 1: two quadrant pixels per thread, 7 + 1 wavefronts x 128 registers): TWO workgroups share a CU, of the
//     same launch or -- this is what pays -- of consecutive launches on different streams, so that one workgroup
//     samples / writes while the other one's image is still landing and HBM never waits for a launch to drain;
//   * a tap descriptor is ONE LDS address + the 2 fractional parts per pixel (+ 2 flag bits): the four taps of a channel
//     are reads at immediate offsets from it, a tap outside the image is read anyway (the word behind it is always
//     inside the workgroup's LDS) and replaced by zero, and the four weights are re-formed when the taps are consumed
//     (the same two subtractions and four products as make_taps: same bits) -- 3 registers per pixel instead of 8;
//   * templated over the geometry: even OW, OH % 16 == 0, H*W % 4 == 0, planes that fit.
// As in the pair kernel: a table row (packed copy, 16-byte pieces in thread order) serves the 4 mirror pixels of all
// IMGS images; LDS-DMA with the nt policy from dedicated loader wavefronts that hold image B's requests back behind
// image A's; flags instead of barriers wherever a wavefront would wait for an event it does not need; flat 16-byte nt
// stores of the staged image.
// The arithmetic is unchanged: T rows and grid coordinates are the k-ascending fp32 FMA chains from zero, taps and
// weights as in tpspp_warp_dev.h.  Compiled with -ffp-contract=off.
#pragma once
#include "tpspp_warp_pair.h"

namespace tpspp_img {

using namespace tpspp_dev;
using tpspp_pair::gchar;
using tpspp_pair::perm_x;
using tpspp_pair::perm_y;
using tpspp_pair::store16_nt;
using tpspp_pair::v4f;
using tpspp_pair::wait_flag;
using tpspp_pair::wait_flag_lds;
using tpspp_pair::make_taps_lite;
using tpspp_pair::TapsLite;

struct ImgParams {
    const float* in; const float* ctrl; const float* inv_delta_c;
    const float* packed;   // pack_img_table_kernel layout (same QP as the kernel)
    int N;
    float* out; float* grid; int32_t* idx;
    int late_from;         // workgroups >= late_from request their images only after their grid is expanded (lab knob)
    long long* trace;      // optional: 8 words per workgroup (tpspp_warp_set_trace; layout at the end of the kernel)
};

// geometry of the thread -> pixel mapping (shared with the table packing).  The quadrant (upper left quarter of the
// output) is tiled by blocks of BW columns x BH rows = 32 pixels, one per half-wavefront, shaped so that a block's rows
// start on different LDS banks at a row pitch of OW floats (input and output sizes are equal): 4 x 8 for OW = 100
// (rows 4 banks apart), 32 x 1 where OW is a multiple of 32 (every row starts on the same bank: a 4 x 8 block would
// be an 8-way conflict on every tap read -- 32x128 measured 21 us per 512 images that way, slower than the round-1
// kernel).  A thread owns QP of the (OH/2)/BH row groups.
// Round 5: where OW is a multiple of 32 but the half-row is not (OW = 160: 80 columns), 32-column blocks would compute
// ceil(80 / 32) * 32 = 96 columns -- a fifth of every wavefront's lanes on pixels other lanes own; 16 x 2 blocks cover
// the half-row exactly.  Their two rows start on the same bank (a 2-way conflict on the tap reads), which costs less
// than the lanes: these kernels are bound by vector-ALU time, not by the LDS.
__host__ __device__ constexpr int img_block_w(int OW)
{
    return (OW % 32) == 0 ? (((OW / 2) % 32) == 0 || ((OW / 2) % 16) != 0 ? 32 : 16) : ((OW % 32) & -(OW % 32));
}
template <int OH, int OW, int QP>
struct ImgGeo {
    static constexpr int halfW = OW / 2;
    static constexpr int BW = img_block_w(OW), BH = 32 / BW;
    static constexpr int CG = (halfW + BW - 1) / BW;         // column groups per half-row
    static constexpr int RG = (OH / 2) / BH;                 // row groups of the upper half
    static_assert(OW % 4 == 0 && OH % 16 == 0 && CG * BW <= OW, "needs whole pixel blocks");
    static_assert(RG % QP == 0, "quadrant pixels per thread must divide the row groups");
    static constexpr int nthr = CG * (RG / QP) * 32;         // compute threads
    static constexpr int NW = (nthr + kWave - 1) / kWave;    // compute wavefronts
};
// thread t, its quadrant pixel j -> (r, c)
__host__ __device__ inline void img_thread_pixel(int t, int j, int CG, int QP, int BW, int* r, int* c)
{
    const int hw = t >> 5, l5 = t & 31, rgb = hw / CG, cg = hw - rgb * CG;
    *r = (rgb * QP + j) * (32 / BW) + l5 / BW;
    *c = cg * BW + l5 % BW;
}

constexpr int kImgAwait = 6;   // image B's requests start when <= 6 of a loader's requests for A are outstanding,
constexpr int kImgKB = 4;      // flag A is raised after 4 of them have been issued (see the pair kernel)

// LDS layout (floats): T (2 K IMGS, padded) | inv_delta_C (IMGS copies, padded) | flags (4) | the IMGS images, contiguous
// as in HBM, rounded up to whole 1-KB DMA pieces | W + 4 floats that out-of-image taps of the last row may read
template <int F, int C, int H, int W, int OH, int OW, int IMGS>
struct ImgLds {
    static constexpr int K = F + 3;
    static constexpr int inv1 = (K * K + 3) & ~3;
    static constexpr int t_off = 0;
    static constexpr int inv_off = (2 * K * IMGS + 3) & ~3;
    static constexpr int flag_off = inv_off + IMGS * inv1;
    static constexpr int img_off = flag_off + 4;
    static constexpr int img_elems = C * H * W;              // floats per image (input and output: OH * OW <= H * W)
    static constexpr int pieces = (IMGS * img_elems * 4 + 1023) / 1024;
    static constexpr size_t bytes = (size_t)(img_off + pieces * 256 + W + 4) * 4;
    static_assert(OH * OW <= H * W, "results are staged in place of the image: the output must not be larger");
};

// WPC: workgroups that must fit a CU side by side (sets the register budget).
// (Measured and dropped: the two taps of a row as one 8-byte access -- ds_read2_b32 is no faster than two ds_read_b32,
// ds_read_b64 at 4-byte alignment is 2.5x slower for the whole kernel.)
template <int F, int C, int H, int W, int OH, int OW, int IMGS, int QP, int NLOAD, int WPC, bool AUX, bool TRACE>
__global__ void __launch_bounds__((ImgGeo<OH, OW, QP>::NW + NLOAD) * kWave, (WPC * (ImgGeo<OH, OW, QP>::NW + NLOAD) + 3) / 4)
tps_warp_img_kernel(const ImgParams P)
{
    using Geo = ImgGeo<OH, OW, QP>;
    using L = ImgLds<F, C, H, W, OH, OW, IMGS>;
    constexpr int K = F + 3;
    constexpr int nthr = Geo::nthr, NW = Geo::NW;
    constexpr int n = OH * OW, HW = H * W, img_elems = C * HW;
    static_assert(HW % 4 == 0 && (C * n) % 4 == 0, "planes and output images are moved in 16-byte pieces");
    static_assert(IMGS == 1 || IMGS == 2, "one image or an image pair per workgroup");
    static_assert(IMGS == 1 || ((C * H * W * 4) >> 10) / NLOAD >= kImgKB, "too few pieces per loader for the A / B hand-over");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* sT = reinterpret_cast<float2*>(smem + L::t_off);      // [K][IMGS]
    float* sInv = smem + L::inv_off;
    float* sFlag = smem + L::flag_off;     // [0] T rows published, [1] / [2] loaders done with image A / B, [3] grids expanded
    float* sImg = smem + L::img_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b0 = blockIdx.x * IMGS;
    const bool hasB = IMGS == 2 && (b0 + 1) < P.N;

    long long ts[6];
#define IMG_STAMP(i) do { if (TRACE) ts[i] = (long long)wall_clock64(); } while (0)
    IMG_STAMP(5);                                            // kernel entry

    // T-solve inputs first (wavefront g -> image b0 + g; lane i keeps control point i), ahead of this CU's image traffic
    constexpr int KK = K * K;
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;
    if (wv < IMGS) {
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
    lds_only_barrier();                                      // the only barrier every wavefront takes part in

    if (wv >= NW) {
        // ================= loader wavefronts =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        constexpr int PA = (img_elems * 4 + 1023) >> 10;     // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;                 // tail lanes re-read a valid address (their 16 bytes land behind the images)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, 2 /* nt */);
        };
        // The flag operands live in registers BEFORE the first DMA: a VGPR written after it had been an operand of an
        // LDS-DMA instruction makes the compiler insert s_waitcnt vmcnt(0) first.  The flag updates are inline asm for
        // the same reason (a visible LDS access is ordered behind ALL outstanding LDS-DMA of the wavefront).
        unsigned fa = (unsigned)(size_t)(sFlag + 1), fb = (unsigned)(size_t)(sFlag + 2);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(fb), "+v"(one));
        if ((int)blockIdx.x >= P.late_from) wait_flag(sFlag + 3, NW);
        int piece = lw;
        for (; piece < PA; piece += NLOAD) dma(piece);
        if (hasB) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kImgAwait) : "memory");
#pragma unroll
            for (int i = 0; i < kImgKB; ++i) { dma(piece); piece += NLOAD; }
            // vmcnt retires in order: once at most kImgKB requests are outstanding and kImgKB of image B's have been
            // issued behind image A's, A is complete
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kImgKB) : "memory");
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
            for (; piece < pieces; piece += NLOAD) dma(piece);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fb), "v"(one) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fa), "v"(one) : "memory");
        }
        return;
    }

    // ================= compute wavefronts =================
    const bool live = tid < nthr;
    const int tt = live ? tid : nthr - 1;
    // byte offsets of the 4 mirror pixels of quadrant pixel j in a plane: derived from the thread index where they are
    // needed (the staging writes, the optional grid / index stores) instead of living in 4 QP registers across the kernel
    bool xdup;                                               // middle group: its x-mirror is another thread's pixel
    {
        int r, c;
        img_thread_pixel(tt, 0, Geo::CG, QP, Geo::BW, &r, &c);
        xdup = (c & ~(Geo::BW - 1)) + Geo::BW > Geo::halfW;
    }
    auto pixel_off = [&](int t_, int j, int m) -> unsigned {
        int r, c;
        img_thread_pixel(t_, j, Geo::CG, QP, Geo::BW, &r, &c);
        const int rr = (m & 2) ? OH - 1 - r : r, cc = (m & 1) ? OW - 1 - c : c;
        return 4u * (unsigned)(rr * OW + cc);
    };

    // packed table: [wavefront][QP][KG][lane] x 16 bytes = this thread's K values per quadrant pixel
    constexpr int KG = (K + 3) / 4;
    float v[QP][KG * 4];
    {
        const v4f* pk = reinterpret_cast<const v4f*>(P.packed) + (size_t)wv * QP * KG * kWave + lane;
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const v4f x = pk[(j * KG + g) * kWave];
                v[j][4 * g] = x[0]; v[j][4 * g + 1] = x[1]; v[j][4 * g + 2] = x[2]; v[j][4 * g + 3] = x[3];
            }
    }
    if (wv < IMGS) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[wv * L::inv1 + e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes visible to own reads
        const float* hrow = sInv + wv * L::inv1 + (lane < K ? lane : K - 1) * K;   // stride K is odd: no conflicts
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {                        // ordered broadcast: the sum is the reference's FMA chain
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) sT[lane * IMGS + wv] = make_float2(ax, ay);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(sFlag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wait_flag_lds(sFlag + 0, IMGS);                          // (LDS-instruction poll: tpspp_warp_pair.h)
    IMG_STAMP(0);                                            // T ready

    // ---- 8 QP IMGS FMA chains: QP quadrant pixels x 4 mirror pixels x IMGS images x (x, y), each k-ascending from zero ----
    float gx[IMGS][QP][4], gy[IMGS][QP][4];
#pragma unroll
    for (int im = 0; im < IMGS; ++im)
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) gx[im][j][m] = gy[im][j][m] = 0.0f;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        float2 t[IMGS];
#pragma unroll
        for (int im = 0; im < IMGS; ++im) t[im] = sT[q * IMGS + im];
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            float val[4];
            if constexpr (q == 0) {
                val[0] = val[1] = val[2] = val[3] = v[j][0];
            } else if constexpr (q == 1) {                    // P.x flips under the x-mirror
                val[0] = v[j][1]; val[1] = -v[j][1]; val[2] = v[j][1]; val[3] = -v[j][1];
            } else if constexpr (q == 2) {                    // P.y flips under the y-mirror
                val[0] = v[j][2]; val[1] = v[j][2]; val[2] = -v[j][2]; val[3] = -v[j][2];
            } else {
                constexpr int k = q - 3;
                val[0] = v[j][3 + k];
                val[1] = v[j][3 + perm_x<F>(k)];
                val[2] = v[j][3 + perm_y<F>(k)];
                val[3] = v[j][3 + perm_x<F>(perm_y<F>(k))];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int im = 0; im < IMGS; ++im) {
                    gx[im][j][m] = fmaf(val[m], t[im].x, gx[im][j][m]);
                    gy[im][j][m] = fmaf(val[m], t[im].y, gy[im][j][m]);
                }
        }
    });
#pragma unroll
    for (int im = 0; im < IMGS; ++im)
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(gx[im][j][m]), "+v"(gy[im][j][m]));
    if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(sFlag) + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    typedef __attribute__((address_space(3))) const float lds_cfloat;
    constexpr unsigned row_bytes = n * 4u;
    constexpr int out16 = (C * n) >> 2;                      // 16-byte pieces of one output image
    constexpr int nct = NW * kWave;
    constexpr int NOUT = (out16 + nct - 1) / nct;

    // ---- tap descriptors of all images: pure VALU work, done while image A is in flight.  One LDS address + two
    // fractions per pixel, two flag bits per pixel in one register per image ----
    unsigned ta[IMGS][QP][4];                                // LDS byte address of the north-west tap in plane 0
    float tf[IMGS][QP][4][2];                                // fractional parts
    unsigned oob[IMGS];                                      // bit 2 p: east column outside, bit 2 p + 1: south row outside (pixel p = 4 j + m)
    static_for<IMGS>([&](auto imc) {
        constexpr int im = decltype(imc)::value;
        const int b = b0 + im;
        const unsigned img_lds = (unsigned)(size_t)(sImg + im * img_elems);
        oob[im] = 0;
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const TapsLite t = make_taps_lite(gx[im][j][m], gy[im][j][m], H, W);   // same bits, fewer VALU instructions
                if constexpr (AUX) {
                    const bool st = live && !((m & 1) && xdup) && (im == 0 || hasB);
                    if (P.grid && st)
                        *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * pixel_off(tt, j, m)) =
                            make_float2(gx[im][j][m], gy[im][j][m]);
                    if (P.idx && st)
                        *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * pixel_off(tt, j, m)) =
                            make_int2(t.x0, t.y0);
                }
                ta[im][j][m] = img_lds + 4u * (unsigned)t.o00;
                tf[im][j][m][0] = t.wx; tf[im][j][m][1] = t.wy;
                oob[im] |= (t.inx ? 0u : 1u) << (2 * (4 * j + m));
                oob[im] |= (t.iny ? 0u : 2u) << (2 * (4 * j + m));
            }
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(tf[im][j][m][0]), "+v"(tf[im][j][m][1]), "+v"(ta[im][j][m]));
        asm volatile("" : "+v"(oob[im]));
    });
    IMG_STAMP(1);                                            // grid + tap descriptors done

    v4f o[NOUT];                                             // an image's 16-byte output pieces between LDS and HBM
    auto store_image = [&](int b) {                          // flat copy out: 16 bytes per lane, 1 KB per wavefront store, nt
        gchar* ob = (gchar*)(P.out) + (size_t)b * C * row_bytes;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) store16_nt(ob + 16u * (unsigned)e, o[i]);
        }
    };
    static_for<IMGS>([&](auto imc) {
        constexpr int im = decltype(imc)::value;
        if (im == 1 && !hasB) { store_image(b0); return; }   // odd batch: the last workgroup has no image B
        wait_flag_lds(sFlag + 1 + im, NLOAD);                // image `im` has landed
        if (im == 0) IMG_STAMP(2);
        // the four taps of every channel at immediate offsets from the pixel's address; a tap outside the image is read
        // anyway (the word exists: next row, next plane, next image or the pad behind the last one)
        float res[QP][4][C];
        // wavefronts whose pixels all have their four taps inside the image (most of them) skip the zero selects
        const bool any_oob = __builtin_amdgcn_ballot_w64(oob[im] != 0u) != 0;
        // MB mirror pixels' taps in flight at a time: 4 where the register budget is a whole CU per workgroup, 2 where two
        // workgroups share it
        constexpr int MB = WPC >= 2 ? 2 : 4;
        static_for<QP * (4 / MB)>([&](auto jc) {
            constexpr int j = decltype(jc)::value / (4 / MB), mb0 = (decltype(jc)::value % (4 / MB)) * MB;
            float tv[MB][C][4];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                lds_cfloat* a = (lds_cfloat*)(size_t)(ta[im][j][mb0 + mm]);
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    tv[mm][ch][0] = a[ch * HW];
                    tv[mm][ch][1] = a[ch * HW + 1];
                    tv[mm][ch][2] = a[ch * HW + W];
                    tv[mm][ch][3] = a[ch * HW + W + 1];
                }
            }
            if (im == 1 && decltype(jc)::value == 0) {
                // image A's output pieces leave while the LDS serves image B's tap reads: their issue is back-pressured
                // by HBM and would otherwise sit on the critical path between the two images
                __builtin_amdgcn_sched_barrier(0);
                store_image(b0);
                __builtin_amdgcn_sched_barrier(0);
            }
            auto combine = [&](auto oobc) {
                constexpr bool OOB = decltype(oobc)::value;
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
                    constexpr int dummy = 0; (void)dummy;
                    const int m = mb0 + mm;
                    const unsigned fl = oob[im] >> (2 * (4 * j + m));
                    const bool inx = !(fl & 1u), iny = !(fl & 2u), inxy = !(fl & 3u);
                    const float w = tf[im][j][m][0], nn = tf[im][j][m][1];
                    const float e = 1.0f - w, s = 1.0f - nn;
                    const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float v01 = (!OOB || inx) ? tv[mm][ch][1] : 0.0f;
                        const float v10 = (!OOB || iny) ? tv[mm][ch][2] : 0.0f;
                        const float v11 = (!OOB || inxy) ? tv[mm][ch][3] : 0.0f;
                        float acc = tv[mm][ch][0] * nw;
                        acc = fmaf(v01, ne, acc);
                        acc = fmaf(v10, sw, acc);
                        acc = fmaf(v11, se, acc);
                        res[j][m][ch] = acc;
                    }
                }
            };
            if (any_oob) combine(std::true_type{}); else combine(std::false_type{});
        });
        lds_only_barrier();                                  // every tap of the image is in registers: its planes are free
        // results in place of the image, in the output's own layout (C, OH, OW)
        char* stage = reinterpret_cast<char*>(sImg + im * img_elems);
        int tq = tt;
        asm volatile("" : "+v"(tq));                         // (keeps the offsets' arithmetic here, not hoisted to the top)
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned po = pixel_off(tq, j, m);
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    if (live) *reinterpret_cast<float*>(stage + ch * (int)row_bytes + po) = res[j][m][ch];
            }
        lds_only_barrier();                                  // results staged
        if (im == 0) IMG_STAMP(3);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int e = tid + i * nct;
            if (e < out16) o[i] = *reinterpret_cast<const v4f*>(stage + 16 * e);
        }
        if (im == IMGS - 1) {
            store_image(b0 + im);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NOUT; ++i) asm volatile("" : "+v"(o[i]));
        }
    });
    if (TRACE && wv == 0) {
        IMG_STAMP(4);                                        // all stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's stores retired
        if (lane == 0 && P.trace) {
            // chip-wide 100 MHz clock: 0 T ready, 1 grid + descriptors (image A), 2 image A landed, 3 image A staged,
            // 4 stores issued, 5 where it ran (xcc << 8 | se / sh / cu), 6 retired, 7 kernel entry
            long long* t = P.trace + (size_t)blockIdx.x * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = ts[i];
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            t[5] = (long long)(((xcc & 0xf) << 8) | ((hw >> 8) & 0xff));
            t[6] = (long long)wall_clock64();
            t[7] = ts[5];
        }
    }
#undef IMG_STAMP
}

// [wavefront][QP][KG][lane][4]: the table values of thread (wavefront, lane)'s quadrant pixel j, q = 4 g .. 4 g + 3
static __global__ void __launch_bounds__(256)
pack_img_table_kernel(const float* __restrict__ p_hat, int p_hat_ld, int OW, int CG, int QP, int BW, int nthr, int K,
                      float* __restrict__ packed)
{
    const int KG = (K + 3) / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // index into packed
    const int NW = (nthr + kWave - 1) / kWave;
    const int total = NW * QP * KG * kWave * 4;
    if (i >= total) return;
    const int comp = i & 3, l = (i >> 2) & (kWave - 1);
    int rest = i >> 8;
    const int g = rest % KG; rest /= KG;
    const int j = rest % QP; const int w = rest / QP;
    const int t = w * kWave + l, q = 4 * g + comp;
    float val = 0.0f;
    if (t < nthr && q < K) {
        int r, c;
        img_thread_pixel(t, j, CG, QP, BW, &r, &c);
        val = p_hat[(size_t)(r * OW + c) * p_hat_ld + q];
    }
    packed[i] = val;
}

}  // namespace tpspp_img
