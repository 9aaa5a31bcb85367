/*
 * oracle/tps_oracle.c -- CPU restatement of the reference's TPS grid generator + bilinear sampler.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the reported CPU baseline.
 * The product path (tps_pp_amd/) never links, imports or calls it.
 *
 * Parity pin: the reference holds no numeric golden vector for this path,
 * so this restatement is pinned against outputs of the reference itself, run in the build
 * container by tests/golden/make_golden.py and committed under tests/golden/ (*.npz);
 * tests/test_oracle_golden.py replays them: the grid must match the reference's
 * torch.bmm result BIT FOR BIT, and so must the warped image match F.grid_sample (weight_form 2
 * below restates the FMA-contracted AVX512 kernel the reference's CPU run really executes).
 *
 * Functions follow (paths relative to /root/reference):
 *   tps_oracle_solve_T        mmocr/models/textrecog/preprocessor/tps_preprocessor.py:273-280
 *                             mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:484,489-494
 *   tps_oracle_grid           tps_preprocessor.py:281 (P_hat @ T)
 *                             tps_pp.py:467-479 (P_hat * (score*0.5 + 1), cat[1, P, .]) and :495
 *   tps_oracle_grid_sample    F.grid_sample(mode='bilinear', padding_mode='border',
 *                             align_corners=True) as called at tps_preprocessor.py:79-83 and
 *                             tps_pp.py:606-615; arithmetic = torch ATen/native/GridSampler.h
 *                             (grid_sampler_unnormalize, clip_coordinates, within_bounds_2d) and
 *                             the scalar bilinear body of GridSampler.cpp / GridSampler.cu.
 *   tps_oracle_warp           the three chained, the grid kept in a scratch buffer.
 *
 * Summation order.  torch.bmm on the CPU (MKL sgemm, K = F+3 <= 35, N = 2) was measured to be
 * bit-identical to a zero-initialised, k-ascending fp32 FMA chain for both products
 * (make_golden.py re-checks that on every regeneration); that chain is what is written here with
 * explicit fmaf().  Compile with -ffp-contract=off so nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* T[b] = inv_delta_C (F+3 x F+3) @ [ctrl[b] (F x 2) ; 0 (3 x 2)]  ->  (F+3) x 2, (x, y) interleaved */
ORACLE_API void tps_oracle_solve_T(const float* inv_delta_c, const float* ctrl, int N, int F,
                                   float* T)
{
    const int K = F + 3;
    for (int b = 0; b < N; ++b) {
        const float* c = ctrl + (size_t)b * F * 2;
        float* t = T + (size_t)b * K * 2;
        for (int i = 0; i < K; ++i) {
            float ax = 0.0f, ay = 0.0f;
            for (int q = 0; q < K; ++q) {
                const float cx = q < F ? c[2 * q + 0] : 0.0f; /* the three appended zero rows */
                const float cy = q < F ? c[2 * q + 1] : 0.0f;
                const float h = inv_delta_c[(size_t)i * K + q];
                ax = fmaf(h, cx, ax);
                ay = fmaf(h, cy, ay);
            }
            t[2 * i + 0] = ax;
            t[2 * i + 1] = ay;
        }
    }
}

/*
 * grid[b, p, :] = row(b, p) @ T[b],  row = [1, P.x, P.y, m_0 .. m_{F-1}]
 *   p_xy == NULL : classic layout, p_hat is (n, F+3) with leading dimension p_hat_ld and already
 *                  holds [1, P.x, P.y, rbf...]                       (tps_preprocessor.py:255-268)
 *   p_xy != NULL : TPS_PP layout, p_hat is (n, F) rbf only, p_xy is (n, 2)   (tps_pp.py:452-479)
 *   score != NULL: m_k = p_hat[p,k] * (score[b,p,k] * 0.5f + 1.0f) -- mul, add, mul, three
 *                  separately rounded fp32 ops (tps_pp.py:474); else m_k = rbf value.
 */
ORACLE_API void tps_oracle_grid(const float* p_hat, int p_hat_ld, const float* p_xy,
                                const float* score, const float* T, int N, int n, int F,
                                float* grid)
{
    const int K = F + 3;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < N; ++b) {
        const float* t = T + (size_t)b * K * 2;
        for (int p = 0; p < n; ++p) {
            const float* ph = p_hat + (size_t)p * p_hat_ld;
            float r0, r1, r2;
            const float* rbf;
            if (p_xy) { r0 = 1.0f; r1 = p_xy[2 * p]; r2 = p_xy[2 * p + 1]; rbf = ph; }
            else      { r0 = ph[0]; r1 = ph[1]; r2 = ph[2]; rbf = ph + 3; }
            float ax = 0.0f, ay = 0.0f;
            ax = fmaf(r0, t[0], ax); ay = fmaf(r0, t[1], ay);
            ax = fmaf(r1, t[2], ax); ay = fmaf(r1, t[3], ay);
            ax = fmaf(r2, t[4], ax); ay = fmaf(r2, t[5], ay);
            const float* s = score ? score + ((size_t)b * n + p) * F : NULL;
            for (int k = 0; k < F; ++k) {
                float m = rbf[k];
                if (s) {
                    float g = s[k] * 0.5f;
                    g = g + 1.0f;
                    m = m * g;
                }
                ax = fmaf(m, t[2 * (3 + k) + 0], ax);
                ay = fmaf(m, t[2 * (3 + k) + 1], ay);
            }
            grid[((size_t)b * n + p) * 2 + 0] = ax;
            grid[((size_t)b * n + p) * 2 + 1] = ay;
        }
    }
}

static inline float unnormalize_ac(float coord, int size)
{
    /* GridSampler.h grid_sampler_unnormalize, align_corners=true: ((coord + 1) / 2) * (size - 1) */
    return ((coord + 1.0f) / 2.0f) * (float)(size - 1);
}

static inline float clip_border(float in, int clip_limit)
{
    /* GridSampler.h clip_coordinates: std::min(limit - 1, std::max(in, 0)) with std:: semantics */
    const float lim = (float)(clip_limit - 1);
    float m = (in > 0.0f) ? in : 0.0f; /* max(in, 0); a NaN coordinate maps to 0 (the reference
                                          leaves NaN undefined: it would index with a cast NaN) */
    return (m < lim) ? m : lim;        /* min(lim, m) */
}

/*
 * out[b,c,i,j] = bilinear(in[b,c], grid[b,i,j]) with border padding, align_corners=True.
 * idx (optional): int32 (N, Ho*Wo, 2) = (ix_nw, iy_nw) of every output pixel -- the "sampling-grid
 * indices" the north star wants bit-exact.
 * weight_form 0: GridSampler.h / GridSampler.cu scalar form: nw = (ix_se - ix) * (iy_se - iy) ...,
 *                out = ((v_nw*nw + v_ne*ne) + v_sw*sw) + v_se*se, nothing fused.
 * weight_form 1: GridSamplerKernel.cpp weights (w = ix - ix_nw, e = 1 - w, n = iy - iy_nw,
 *                s = 1 - n; nw = s*e, ne = s*w, sw = n*e, se = n*w), accumulation as form 0.
 * weight_form 2: (DEFAULT of tps_oracle_warp) form-1 weights and the accumulation the reference's
 *                CPU kernel really performs, torch 2.10 AVX512 build with FMA contraction:
 *                out = fma(v_se, se, fma(v_sw, sw, fma(v_ne, ne, v_nw * nw))).
 *                Measured BIT-IDENTICAL to F.grid_sample on the CPU (make_golden.py re-checks).
 * The unnormalise step of GridSamplerKernel.cpp is (g + 1) * ((size-1)/2); it equals the header's
 * ((g + 1) / 2) * (size - 1) bit for bit (a division by two is exact), so one form serves all.
 */
ORACLE_API void tps_oracle_grid_sample(const float* in, const float* grid, int N, int C, int H,
                                       int W, int Ho, int Wo, float* out, int32_t* idx,
                                       int weight_form)
{
    const int n = Ho * Wo;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < N; ++b) {
        const float* g = grid + (size_t)b * n * 2;
        for (int p = 0; p < n; ++p) {
            float ix = unnormalize_ac(g[2 * p + 0], W);
            float iy = unnormalize_ac(g[2 * p + 1], H);
            ix = clip_border(ix, W);
            iy = clip_border(iy, H);
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const int x1 = x0 + 1, y1 = y0 + 1;
            float nw, ne, sw, se;
            if (weight_form == 0) {
                const float xe = (float)x1, ys = (float)y1;
                nw = (xe - ix) * (ys - iy);
                ne = (ix - fx) * (ys - iy);
                sw = (xe - ix) * (iy - fy);
                se = (ix - fx) * (iy - fy);
            } else {
                const float w_ = ix - fx, e_ = 1.0f - w_, n_ = iy - fy, s_ = 1.0f - n_;
                nw = s_ * e_; ne = s_ * w_; sw = n_ * e_; se = n_ * w_;
            }
            const int fused = (weight_form == 2);
            if (idx) {
                idx[((size_t)b * n + p) * 2 + 0] = x0;
                idx[((size_t)b * n + p) * 2 + 1] = y0;
            }
            const int in00 = (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W);
            const int in01 = (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W);
            const int in10 = (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W);
            const int in11 = (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W);
            for (int c = 0; c < C; ++c) {
                const float* pl = in + ((size_t)b * C + c) * H * W;
                /* an out-of-range corner reads as 0 (mask_gather with src = 0 / the skipped
                 * `if (within_bounds_2d)` of the scalar kernel) */
                const float v00 = in00 ? pl[(size_t)y0 * W + x0] : 0.0f;
                const float v01 = in01 ? pl[(size_t)y0 * W + x1] : 0.0f;
                const float v10 = in10 ? pl[(size_t)y1 * W + x0] : 0.0f;
                const float v11 = in11 ? pl[(size_t)y1 * W + x1] : 0.0f;
                float acc;
                if (fused) {
                    acc = v00 * nw;
                    acc = fmaf(v01, ne, acc);
                    acc = fmaf(v10, sw, acc);
                    acc = fmaf(v11, se, acc);
                } else {
                    float t;
                    acc = 0.0f;
                    if (in00) { t = v00 * nw; acc = acc + t; }
                    if (in01) { t = v01 * ne; acc = acc + t; }
                    if (in10) { t = v10 * sw; acc = acc + t; }
                    if (in11) { t = v11 * se; acc = acc + t; }
                }
                out[((size_t)b * C + c) * n + p] = acc;
            }
        }
    }
}

/*
 * The whole hot path for one batch: T-solve -> grid -> sample in0 (and in1 when given).
 * grid_out / idx0_out may be NULL.  Returns 0, or -1 on allocation failure.
 */
ORACLE_API int tps_oracle_warp(const float* in0, int C0, int H0, int W0,
                               const float* in1, int C1, int H1, int W1,
                               const float* ctrl, const float* score,
                               const float* inv_delta_c, const float* p_hat, int p_hat_ld,
                               const float* p_xy, int N, int F, int Ho, int Wo,
                               float* out0, float* out1, float* grid_out, int32_t* idx0_out)
{
    const int n = Ho * Wo, K = F + 3;
    float* T = (float*)malloc((size_t)N * K * 2 * sizeof(float));
    float* grid = grid_out ? grid_out : (float*)malloc((size_t)N * n * 2 * sizeof(float));
    if (!T || !grid) { free(T); if (!grid_out) free(grid); return -1; }
    tps_oracle_solve_T(inv_delta_c, ctrl, N, F, T);
    tps_oracle_grid(p_hat, p_hat_ld, p_xy, score, T, N, n, F, grid);
    tps_oracle_grid_sample(in0, grid, N, C0, H0, W0, Ho, Wo, out0, idx0_out, 2);
    if (in1) tps_oracle_grid_sample(in1, grid, N, C1, H1, W1, Ho, Wo, out1, NULL, 2);
    free(T);
    if (!grid_out) free(grid);
    return 0;
}

ORACLE_API int tps_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void tps_oracle_set_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
