// Classic-geometry warp for ANY image size that fits the LDS (round 4): the in-place kernel of tpspp_warp_img.h with the
// geometry as run-time arguments.
//
// Replaces, bit for bit: preprocessor/tps_preprocessor.py:71-83 + 270-282 (GridGenerator.build_P_prime: two bmm,
// then F.grid_sample bilinear / border / align_corners) for a mirror-symmetric RBF table and img_size ==
// rectified_img_size (tps_preprocessor.py:39-58 takes any; the reference's own configs and tests use 32x100, 32x128,
// 32x160, 48x160, 64x256).
//
// tpspp_warp_img.h / tpspp_warp_pair.h are instantiated per geometry (H, W are template arguments: every plane
// offset is an immediate).  Here only (F, C, QP) are compile-time -- they size register arrays -- and H = OH, W = OW,
// the thread -> pixel mapping and the LDS layout are computed at launch.  What that costs: two address additions per
// (pixel, channel) in front of the four tap reads, and run-time loop bounds in the copy-out.  What is kept: one image
// per workgroup staged once in LDS by LDS-DMA (nt), results staged IN PLACE of the image (every tap of the image is in
// registers before the first result is written), a tap descriptor of one LDS address + two fractions per pixel, the
// packed table (16-byte pieces in thread order), flat 16-byte nt stores, flags polled with LDS instructions, the lean
// tap arithmetic of make_taps_lite().
// Large images: a thread owns at most 4 quadrant pixels (register budget), a workgroup at most 13 compute wavefronts;
// when that does not cover the quadrant, `bands` workgroups share an image: each stages the WHOLE image (taps may
// fall anywhere) but expands, samples and writes only its own row groups (and their mirror rows).
// The arithmetic is unchanged: T rows and grid coordinates are the k-ascending fp32 FMA chains from zero.  Compiled
// with -ffp-contract=off.
#pragma once
#include "tpspp_warp_img.h"

namespace tpspp_geo {

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

struct GeoParams {
    const float* in; const float* ctrl; const float* inv_delta_c;
    const float* packed;   // pack_img_table_kernel layout for (QP, BW, CG) below, thread order of the WHOLE quadrant
    int N;
    float* out; float* grid; int32_t* idx;
    int H, W;              // input = output size
    int BW, CG;            // pixel block width (32 / BW rows), column groups per half-row
    int RGB;               // row-group batches of the quadrant (a thread owns QP row groups of one batch)
    int bands;             // workgroups per image; each owns RGB / bands batches
    int nthr, NW;          // compute threads / wavefronts per workgroup
    int img_off;           // float offset of the staged image in LDS (behind T and the flags)
};

// LDS floats in front of the images: T (2 K per image, padded to 4) + 4 flag words
__host__ __device__ constexpr int geo_img_off(int K, int IMGS) { return ((2 * K * IMGS + 3) & ~3) + 4; }
inline size_t geo_lds_bytes(int K, int C, int H, int W, int IMGS)
{
    const int pieces = (IMGS * C * H * W * 4 + 1023) / 1024;
    return (size_t)(geo_img_off(K, IMGS) + pieces * 256 + W + 4) * 4;   // + what out-of-image taps of the last row may read
}

constexpr int kGeoAwait = 6;   // image B's requests start when <= 6 of a loader's requests for A are outstanding,
constexpr int kGeoKB = 4;      // flag A is raised after 4 of them have been issued (the launcher guarantees that many)

// register budget: up to 4 quadrant pixels x C channels of results per thread; from QP = 3 on the launcher keeps a workgroup
// at <= 12 wavefronts (more bands), i.e. 3 per SIMD = 168 registers per lane
template <int QP> constexpr int geo_max_threads() { return QP >= 3 ? 768 : 1024; }

// IMGS = 2 (round 4): an image PAIR per workgroup where two images fit the LDS and one workgroup covers the quadrant --
// image B lands while image A is sampled, one image's T-solve latency and tails overlap the other's ALU work (32x160:
// 15.1 against 17.9 us per 512 images in the instantiated kernel).  The loaders hold B's requests back behind A's as in
// tpspp_warp_pair.h; a table value serves the 4 mirror pixels of both images.
template <int F, int C, int QP, int IMGS, bool AUX>
__global__ void __launch_bounds__(geo_max_threads<QP>())
tps_warp_geo_kernel(const GeoParams P)
{
    constexpr int K = F + 3;
    static_assert(IMGS == 1 || IMGS == 2, "one image or an image pair per workgroup");
    const int H = P.H, W = P.W, HW = H * W, img_elems = C * HW;
    const int BW = P.BW, BH = 32 / BW, CG = P.CG, NW = P.NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* sT = reinterpret_cast<float2*>(smem);           // [K][IMGS]
    float* sFlag = smem + ((2 * K * IMGS + 3) & ~3);        // [0] T rows published, [1] / [2] loaders done with image A / B
    float* sImg = smem + P.img_off;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int grp = blockIdx.x / P.bands, band = blockIdx.x - grp * P.bands;    // (IMGS == 2: bands == 1)
    const int b0 = grp * IMGS;
    const bool hasB = IMGS == 2 && (b0 + 1) < P.N;
    const int NLOAD = (int)(blockDim.x / kWave) - NW;
    if (NW < IMGS) __builtin_trap();                         // wavefront g < IMGS publishes image g's T: the launcher guarantees NW >= IMGS

    // T-solve inputs first (wavefront g -> image b0 + g; lane i keeps control point i and row i of inv_delta_C, 16 bytes
    // at a time: the last piece starts at column K - 4 so that the last row does not read past the matrix)
    constexpr int KGI = (K + 3) / 4;
    float hrowv[KGI * 4];
    float cx = 0.0f, cy = 0.0f;
    if (wv < IMGS) {
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
    if (tid < 4) reinterpret_cast<int*>(sFlag)[tid] = 0;
    lds_only_barrier();                                      // the only barrier every wavefront takes part in

    if (wv >= NW) {
        // ================= loader wavefronts: 1 KB per instruction =================
        const int lw = wv - NW;
        const int total_bytes = (hasB ? 2 : 1) * img_elems * 4;
        const int pieces = (total_bytes + 1023) >> 10;
        const int PA = (img_elems * 4 + 1023) >> 10;         // pieces that hold bytes of image A
        const char* src = reinterpret_cast<const char*>(P.in + (size_t)b0 * img_elems);
        auto dma = [&](int piece) {
            int off = piece * 1024 + lane * 16;
            if (off >= total_bytes) off = 0;                 // tail lanes re-read a valid address (their bytes land in the pad)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off),
                (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + piece * 1024),
                16, 0, 2 /* nt */);
        };
        // the flag operands live in registers BEFORE the first DMA, the updates are inline asm (tpspp_warp_pair.h)
        unsigned fa = (unsigned)(size_t)(sFlag + 1);
        int one = 1;
        asm volatile("" : "+v"(fa), "+v"(one));
        int piece = lw;
        for (; piece < PA; piece += NLOAD) dma(piece);
        if (hasB) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGeoAwait) : "memory");
#pragma unroll
            for (int i = 0; i < kGeoKB; ++i) { dma(piece); piece += NLOAD; }
            // vmcnt retires in order: once at most kGeoKB requests are outstanding and kGeoKB of image B's have been
            // issued behind image A's, A is complete
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGeoKB) : "memory");
            if (lane == 0) tpspp_pair::flag_add<0>(fa, one);
            for (; piece < pieces; piece += NLOAD) dma(piece);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) tpspp_pair::flag_add<4>(fa, one);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) tpspp_pair::flag_add<0>(fa, one);
        }
        return;
    }

    // ================= compute wavefronts =================
    // thread -> its QP quadrant pixels: half-wavefront hw = block of BW columns x BH rows; row-group batch rgb (global
    // over the bands) holds QP consecutive row groups
    const bool live = tid < P.nthr;                          // (the last wavefront may be half empty)
    const int tt = live ? tid : P.nthr - 1;
    const int hw = tt >> 5, l5 = tt & 31;
    const int rgb_l = hw / CG, cg = hw - rgb_l * CG;
    const int rgb = band * (P.RGB / P.bands) + rgb_l;
    const int c = cg * BW + (l5 & (BW - 1));                 // BW is a power of two; c < W (CG BW <= W)
    const int r0 = rgb * QP * BH + l5 / BW;                  // quadrant pixel j sits BH rows further per j
    // (column groups may reach past the centre: such a lane's pixels are other lanes' mirror pixels, computed twice
    // with the same bits -- the table is mirror-symmetric -- and written twice with the same value)
    auto pixel_off = [&](int j, int m) -> unsigned {
        const int r = r0 + j * BH;
        const int rr = (m & 2) ? H - 1 - r : r, cc = (m & 1) ? W - 1 - c : c;
        return 4u * (unsigned)(rr * W + cc);
    };

    // packed table: [wavefront][QP][KG][lane] x 16 bytes in the thread order of the whole quadrant
    constexpr int KG = (K + 3) / 4;
    float v[QP][KG * 4];
    {
        const int t_glob = (rgb * CG + cg) * 32 + l5;
        const v4f* pk = reinterpret_cast<const v4f*>(P.packed) + (size_t)(t_glob >> 6) * QP * KG * kWave + (t_glob & (kWave - 1));
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const v4f x = pk[(j * KG + g) * kWave];
                v[j][4 * g] = x[0]; v[j][4 * g + 1] = x[1]; v[j][4 * g + 2] = x[2]; v[j][4 * g + 3] = x[3];
            }
    }
    if (wv < IMGS) {
        float ax = 0.0f, ay = 0.0f;
        static_for<K>([&](auto qc) {                         // ordered broadcast: the sum is the reference's FMA chain
            constexpr int q = decltype(qc)::value;
            constexpr int idx = (q / 4 < KGI - 1) ? q : 4 * (KGI - 1) + (q - (K - 4));
            ax = fmaf(hrowv[idx], readlane_f(cx, q), ax);
            ay = fmaf(hrowv[idx], readlane_f(cy, q), ay);
        });
        if (lane < K) sT[lane * IMGS + wv] = make_float2(ax, ay);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(sFlag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wait_flag_lds(sFlag + 0, IMGS);
    asm volatile("" ::"v"(v[QP - 1][KG * 4 - 1]));           // (keeps the table's padding register from being recycled early)

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

    typedef __attribute__((address_space(3))) const float lds_cfloat;
    const unsigned row_bytes = (unsigned)HW * 4u;
    // ---- tap descriptors of all images: one LDS address + two fractions per pixel, two flag bits per pixel ----
    unsigned ta[IMGS][QP][4];
    float tf[IMGS][QP][4][2];
    unsigned oob[IMGS];                                      // bit 2 p: east column outside, bit 2 p + 1: south row outside (p = 4 j + m)
    static_for<IMGS>([&](auto imc) {
        constexpr int im = decltype(imc)::value;
        const int b = b0 + im;
        const unsigned img_lds = (unsigned)(size_t)(sImg + im * img_elems);
        oob[im] = 0;
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const TapsLite t = make_taps_lite(gx[im][j][m], gy[im][j][m], H, W);
                if constexpr (AUX) {
                    const bool st = live && (im == 0 || hasB);
                    if (P.grid && st)
                        *reinterpret_cast<float2*>(reinterpret_cast<char*>(P.grid) + (size_t)b * 2 * row_bytes + 2u * pixel_off(j, m)) =
                            make_float2(gx[im][j][m], gy[im][j][m]);
                    if (P.idx && st)
                        *reinterpret_cast<int2*>(reinterpret_cast<char*>(P.idx) + (size_t)b * 2 * row_bytes + 2u * pixel_off(j, m)) =
                            make_int2(t.x0, t.y0);
                }
                ta[im][j][m] = img_lds + 4u * (unsigned)t.o00;
                tf[im][j][m][0] = t.wx; tf[im][j][m][1] = t.wy;      // the four weights are re-formed from these at the taps
                oob[im] |= (t.inx ? 0u : 1u) << (2 * (4 * j + m));
                oob[im] |= (t.iny ? 0u : 2u) << (2 * (4 * j + m));
            }
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(tf[im][j][m][0]), "+v"(tf[im][j][m][1]), "+v"(ta[im][j][m]));
        asm volatile("" : "+v"(oob[im]));
    });

    // copy-out geometry: this band's rows [ra, ra + rows) and their mirror rows of every channel, 16 bytes per lane
    const int rows = (P.RGB / P.bands) * QP * BH;
    const int ra = band * rows;
    const int seg16 = (rows * W) >> 2;                       // 16-byte pieces of one row range of one plane
    const int nct = NW * kWave;
    auto copy_out = [&](int im) {
        const char* stage = reinterpret_cast<const char*>(sImg + im * img_elems);
        gchar* ob = (gchar*)(P.out) + (size_t)(b0 + im) * C * row_bytes;
        for (int e = tid; e < 2 * C * seg16; e += nct) {
            const int sg = e / seg16, i = e - sg * seg16;    // segment = (channel, upper / lower range)
            const int ch = sg >> 1;
            const int row_lo = (sg & 1) ? H - ra - rows : ra;
            const unsigned off = (unsigned)ch * row_bytes + 4u * (unsigned)(row_lo * W) + 16u * (unsigned)i;
            const v4f x = *reinterpret_cast<const v4f*>(stage + off);
            store16_nt(ob + off, x);
        }
    };

    static_for<IMGS>([&](auto imc) {
        constexpr int im = decltype(imc)::value;
        if (im == 1 && !hasB) return;                        // odd batch: the last workgroup has no image B
        wait_flag_lds(sFlag + 1 + im, NLOAD);                // image `im` has landed
        // the four taps of every channel: row 0 at a, row 1 at a + 4 W, channel planes 4 HW apart; a tap outside the
        // image is read anyway (the word exists: next row, next plane, next image or the pad) and replaced by zero
        float res[QP][4][C];
        const bool any_oob = __builtin_amdgcn_ballot_w64(oob[im] != 0u) != 0;
        constexpr int MB = (QP * C >= 12) ? 1 : 2;           // mirror pixels whose taps are in flight together
        static_for<QP * (4 / MB)>([&](auto jc) {
            constexpr int j = decltype(jc)::value / (4 / MB), mb0 = (decltype(jc)::value % (4 / MB)) * MB;
            float tv[MB][C][4];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                unsigned a0 = ta[im][j][mb0 + mm];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    lds_cfloat* p0 = (lds_cfloat*)(size_t)a0;
                    lds_cfloat* p1 = (lds_cfloat*)(size_t)(a0 + 4u * (unsigned)W);
                    tv[mm][ch][0] = p0[0];
                    tv[mm][ch][1] = p0[1];
                    tv[mm][ch][2] = p1[0];
                    tv[mm][ch][3] = p1[1];
                    a0 += row_bytes;
                }
            }
            auto combine = [&](auto oobc) {
                constexpr bool OOB = decltype(oobc)::value;
#pragma unroll
                for (int mm = 0; mm < MB; ++mm) {
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
        // results in place of the image, in the output's own layout (C, H, W)
        char* stage = reinterpret_cast<char*>(sImg + im * img_elems);
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned po = pixel_off(j, m);
                if (live) {
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) *reinterpret_cast<float*>(stage + ch * row_bytes + po) = res[j][m][ch];
                }
            }
        lds_only_barrier();                                  // results staged
        copy_out(im);
    });
}

}  // namespace tpspp_geo
