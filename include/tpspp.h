/*
 * tpspp.h -- C ABI of libtpspp_hip.so: the TPS++ rectification hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference has no native code and no FFI:
 * its "operator API" for this path is a handful of PyTorch calls inside two nn.Modules.  Each entry
 * point below replaces the call sites cited next to it (paths relative to the reference tree,
 * mmocr/models/textrecog/...); tps_pp_amd/ binds them with ctypes and INTEGRATION.md shows the
 * few lines a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to densely packed row-major fp32 (int32 for indices); tensors
 *     are NCHW like the reference's.  No allocation (one exception: tpspp_warp_plan_create's small host
 *     structure, freed by tpspp_warp_plan_destroy), no ownership transfer, no host synchronisation:
 *     work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     the call returns immediately.  Re-entrant per stream.
 *   - return value: 0 on success, a negative TPSPP_E* code otherwise; tpspp_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - arithmetic contract (what makes the result equal the reference's, not merely close):
 *       T-solve and grid:  zero-initialised, k-ascending fp32 FMA chains -- bit-identical to the
 *                          reference's torch.bmm on the CPU;
 *       sampler:           ATen bilinear / border / align_corners=True with the CPU kernel's weight
 *                          form and FMA order -- bit-identical to F.grid_sample on the CPU.
 */
#ifndef TPSPP_H_
#define TPSPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPSPP_ABI_VERSION 5   /* 2 (round 4): tpspp_warp_bwd and tpspp_nrtr_decoder_fwd carry the sizes of their workspace / pointer table;
                               3: tpspp_down_fused_bf16_fwd / _x3_fwd / _f32_fwd, tpspp_token_gemm_bf16_fwd, tpspp_front_fwd and tpspp_front_bf16_fwd takes feat0 = feat1 = NULL;
                               4 (round 6): tpspp_nrtr_decoder_fwd takes status_out, tpspp_resize_normalize_fwd takes interpolation;
                               5 (round 6): tpspp_warp_plan_create / _run / _run_on / _destroy */

#define TPSPP_OK        0
#define TPSPP_EINVAL  (-22)  /* bad argument (null pointer, non-positive size, unsupported shape) */
#define TPSPP_ENODEV  (-19)  /* no HIP device / wrong architecture */
#define TPSPP_EIO      (-5)  /* the HIP runtime reported a launch error */

typedef void* tpspp_stream_t; /* hipStream_t */

/* ABI version of the loaded library (== TPSPP_ABI_VERSION it was built with). */
int tpspp_abi_version(void);

/* Message for the last non-zero return on this thread ("" if none). */
const char* tpspp_last_error(void);

/*
 * T[b] = inv_delta_c (F+3 x F+3) @ [ctrl[b] (F x 2); 0 (3 x 2)]          -> T (N, F+3, 2)
 * replaces: torch.bmm(batch_inv_delta_C, batch_C_prime_with_zeros)
 *           preprocessor/tps_preprocessor.py:273-280, backbones/tps_pp/tps_pp.py:484,489-494
 */
int tpspp_solve_T(const float* inv_delta_c, const float* ctrl, int N, int F, float* T,
                  tpspp_stream_t stream);

/*
 * grid[b,p,:] = [1, P.x, P.y, m_0..m_{F-1}] @ T[b]                         -> grid (N, n, 2)
 *   p_xy == NULL: p_hat is (n, F+3), leading dimension p_hat_ld, rows [1, P.x, P.y, rbf_0..]
 *                 (GridGenerator.P_hat, tps_preprocessor.py:255-268)
 *   p_xy != NULL: p_hat is (n, F) rbf only (Attention_Enhanced_TPS.P_hat) and p_xy is (n, 2)
 *   score (N, n, F) or NULL: m_k = rbf_k * (score_k * 0.5 + 1)   (tps_pp.py:474, thela = 0.5)
 * replaces: torch.bmm(batch_P_hat, batch_T)   tps_preprocessor.py:281, tps_pp.py:467-479,495
 */
int tpspp_build_grid(const float* p_hat, int p_hat_ld, const float* p_xy, const float* score,
                     const float* T, int N, int n, int F, float* grid, tpspp_stream_t stream);

/*
 * out = grid_sample(in, grid, mode='bilinear', padding_mode='border', align_corners=True)
 *   in (N,C,H,W), grid (N,Ho,Wo,2) -> out (N,C,Ho,Wo); idx_or_null (N,Ho*Wo,2) int32 receives the
 *   north-west corner (ix_nw, iy_nw) of every output pixel.
 * replaces: F.grid_sample   tps_preprocessor.py:79-83, tps_pp.py:606-615
 */
int tpspp_grid_sample(const float* in, const float* grid, int N, int C, int H, int W, int Ho,
                      int Wo, float* out, int32_t* idx_or_null, tpspp_stream_t stream);

/*
 * p_hat_t[c, p] = p_hat[p, c]: the batch-shared RBF table transposed to (cols, n) so that a wavefront
 * whose lanes own consecutive output pixels reads it fully coalesced.  One-off preparation (the
 * table is a module buffer); cols = F+3 (classic layout) or F (TPS_PP layout).
 */
int tpspp_transpose_p_hat(const float* p_hat, int p_hat_ld, int n, int cols, float* p_hat_t,
                          tpspp_stream_t stream);

/*
 * Does the HOST copy of a classic-layout table (n = Ho*Wo rows of [1, P.x, P.y, rbf_0..rbf_{F-1}])
 * have the exact (bitwise) 4-fold mirror symmetry of the reference's constants?  Returns 1 / 0.
 * When 1, pass TPSPP_TABLE_MIRROR4 to tpspp_warp_fwd: one table row then serves four output pixels.
 * (GridGenerator's table always has it: fiducials on the top/bottom edges at mirrored abscissae,
 * tps_preprocessor.py:197-211, pixel centres mirrored about the image centre, :242-253.)
 */
int tpspp_table_mirror_symmetry(const float* p_hat_host, int p_hat_ld, int Ho, int Wo, int F);

/*
 * The prepared form of a mirror-symmetric classic table that the image-pair and in-place kernels read: the (F+3, n)
 * transposed table of tpspp_transpose_p_hat followed by a packed copy in those kernels' thread order (a thread's F+3
 * values per quadrant pixel as 16-byte pieces, [wavefront][quadrant pixels per thread][(F+3+3)/4][lane][4]; thread ->
 * pixel: a half-wavefront owns a block of 32 pixels -- 4 columns x 8 rows, 8 x 4 or 32 x 1 by the row pitch -- of the left
 * half of the upper half-image, a thread 1 - 4 such row groups depending on the geometry); for the large geometries (more
 * than two row groups per thread in that copy) a third section follows: the same packing with ONE row group per thread,
 * read by the span-staging kernel (tpspp_warp_span.h).  tpspp_prepared_table_floats:
 * buffer size in floats, 0 when the geometry has no prepared form (needs Ho % 16 == 0, Wo % 4 == 0).  One-off
 * preparation, like the transposition.  Pass the buffer
 * as p_hat_t together with TPSPP_TABLE_PACKED | TPSPP_TABLE_SPAN (and TPSPP_TABLE_MIRROR4 once the symmetry has been verified).
 * The buffer's size grew in round 5 (the third section): size it with THIS function, never with a remembered formula.
 * replaces nothing in the reference (its table is a module buffer, tps_preprocessor.py:187-188); see tpspp_warp_fwd.
 */
size_t tpspp_prepared_table_floats(int Ho, int Wo, int F);
int tpspp_prepare_mirror_table(const float* p_hat, int p_hat_ld, int Ho, int Wo, int F, float* prepared,
                               tpspp_stream_t stream);

#define TPSPP_TABLE_MIRROR4 1      /* table_flags bit: symmetry verified by the caller */
#define TPSPP_TABLE_PACKED 8       /* table_flags bit: p_hat_t is a tpspp_prepare_mirror_table buffer (the packed copy
                                      follows the transposed table).  Never changes results. */
#define TPSPP_TABLE_SPAN 32        /* table_flags bit, with TPSPP_TABLE_PACKED: the buffer was sized by THIS library's
                                      tpspp_prepared_table_floats and filled by its tpspp_prepare_mirror_table, i.e. it carries the
                                      third section (round 5) that the span-staging kernel of the large geometries reads.  A
                                      caller still holding a two-section buffer of an earlier build omits the bit and gets the
                                      kernels that do not read behind the second section.  Never changes results. */
#define TPSPP_SCORE_TRANSPOSED 2   /* table_flags bit: `score` is laid out (N, F, n) instead of the
                                      reference's (N, n, F): lanes that own consecutive pixels then read
                                      it coalesced.  Same values, same results. */
#define TPSPP_BWD_FIXED_POINT 16   /* table_flags bit of tpspp_warp_bwd only: accumulate dL/d input in 64-bit fixed point (bitwise
                                    * reproducible from run to run) instead of the default fp64 LDS atomics; per call, any stream */
#define TPSPP_BWD_TWO_KERNELS 64   /* table_flags bit of tpspp_warp_bwd only: never the one-launch form of the classic rectifier's call
                                    * (A/B runs, tests); per call.  The environment variable of the same name, read once when
                                    * the first backward runs, does the same for a whole process. */
#define TPSPP_IO_BF16 4            /* table_flags bit: in0 / in1 / out0 / out1 hold bfloat16 (the pointers are
                                      reinterpreted; everything else stays fp32).  The bf16 configuration
                                      (BASELINE.json configs[2]): T, grid and interpolation in fp32 exactly as
                                      without the flag, one round-to-nearest-even at the store.  Shapes the
                                      plane-streaming kernel takes only (TPS_PP geometry); else TPSPP_EINVAL. */

/*
 * The fused hot path: T-solve -> grid -> bilinear warp of in0 (and in1 when non-NULL) in ONE kernel;
 * T lives in LDS and the grid in registers, neither touches HBM unless grid_or_null is given.
 *   in0 (N,C0,H0,W0) -> out0 (N,C0,Ho,Wo);  in1 (N,C1,H1,W1) -> out1 (N,C1,Ho,Wo)  [optional]
 *   ctrl (N,F,2); score (N,Ho*Wo,F) or NULL; inv_delta_c (F+3,F+3); p_hat / p_hat_ld / p_xy as in
 *   tpspp_build_grid; p_hat_t_or_null = tpspp_transpose_p_hat(p_hat) (same values, enables the
 *   coalesced / LDS-staged fast kernels; NULL selects the generic kernel -- identical results);
 *   table_flags: OR of TPSPP_TABLE_MIRROR4, TPSPP_TABLE_PACKED, TPSPP_TABLE_SPAN (only meaningful with p_hat_t),
 *   TPSPP_SCORE_TRANSPOSED (none of them changes results) and TPSPP_IO_BF16 (bf16 images in and out);
 *   grid_or_null (N,Ho*Wo,2); idx_or_null (N,Ho*Wo,2) int32 = NW corner in in0.
 * replaces: GridGenerator.build_P_prime + F.grid_sample   tps_preprocessor.py:71-83
 *           Attention_Enhanced_TPS.build_P_prime + 2x F.grid_sample   tps_pp.py:597-615
 */
int tpspp_warp_fwd(const float* in0, int C0, int H0, int W0,
                   const float* in1, int C1, int H1, int W1,
                   const float* ctrl, const float* score,
                   const float* inv_delta_c, const float* p_hat, int p_hat_ld, const float* p_xy,
                   const float* p_hat_t_or_null, int table_flags, int N, int F, int Ho, int Wo,
                   float* out0, float* out1, float* grid_or_null, int32_t* idx_or_null,
                   tpspp_stream_t stream);

/*
 * Prepared calls of tpspp_warp_fwd (round 6; ABI 5).  A caller that rectifies batch after batch into the same buffers -- a
 * serving loop, bench.py -- hands the 25 arguments over ONCE: tpspp_warp_plan_run(plan) is tpspp_warp_fwd with the stored
 * arguments (same checks, same dispatch, same kernel, same results), tpspp_warp_plan_run_on the same on another stream.  What it
 * saves is the caller's foreign-function marshalling per launch (ctypes: 1.6 of ~4.3 us), which matters when the device needs
 * ~8 us per batch and, at the start of a burst, waits for the host.  The plan holds POINTERS, not copies: the buffers must
 * outlive it.  tpspp_warp_plan_create allocates a small host structure (the only entry point that allocates);
 * tpspp_warp_plan_destroy(NULL) is a no-op.  Thread safety: a plan is immutable after creation.
 * replaces: the same call sites as tpspp_warp_fwd (tps_preprocessor.py:71-83, tps_pp.py:597-615).
 */
typedef struct tpspp_warp_plan tpspp_warp_plan_t;
int tpspp_warp_plan_create(const float* in0, int C0, int H0, int W0,
                           const float* in1, int C1, int H1, int W1,
                           const float* ctrl, const float* score,
                           const float* inv_delta_c, const float* p_hat, int p_hat_ld, const float* p_xy,
                           const float* p_hat_t_or_null, int table_flags, int N, int F, int Ho, int Wo,
                           float* out0, float* out1, float* grid_or_null, int32_t* idx_or_null,
                           tpspp_stream_t stream, tpspp_warp_plan_t** plan_out);
int tpspp_warp_plan_run(const tpspp_warp_plan_t* plan);
int tpspp_warp_plan_run_on(const tpspp_warp_plan_t* plan, tpspp_stream_t stream);
void tpspp_warp_plan_destroy(tpspp_warp_plan_t* plan);

/*
 * Backward of tpspp_warp_fwd: given dL/d out0 [, dL/d out1] returns
 *   g_in0 (N, C0, H0, W0), g_in1 (N, C1, H1, W1)  dL/d input (zeroed here, then accumulated with float
 *                                                  atomics); either may be NULL (not wanted)
 *   g_ctrl (N, F, 2)                                dL/d control points
 *   g_score                                         dL/d score in the layout of `score` ((N, n, F), or
 *                                                  (N, F, n) with TPSPP_SCORE_TRANSPOSED), or NULL
 *   g_grid_ws, g_grid_ws_floats                     scratch and its size in floats, at least
 *                                                  tpspp_warp_bwd_workspace_floats(N, Ho, Wo) (checked: TPSPP_EINVAL
 *                                                  otherwise; the size grew in round 3); on return its first
 *                                                  N * Ho*Wo * 2 floats hold dL/d grid
 * grid (N, Ho*Wo, 2) is the sampling grid the forward produced (its grid_or_null output) and
 * T (N, F+3, 2) = tpspp_solve_T(inv_delta_c, ctrl) (only dL/d score reads it: may be NULL when score is NULL); the tables
 * are the forward's.  Gradients follow
 * ATen's CPU grid_sampler_2d_backward (bilinear, border, align_corners=True: zero coordinate gradient
 * where the coordinate was clamped) and the transposes of the two products of build_P_prime.
 * The classic rectifier's call (one input of <= 3 channels, no score, transposed table given, 1024 < Ho*Wo <= 4096, the
 * fp64 accumulator) is ONE launch; every other call a sampling + a parameter kernel.  Same sampling arithmetic either way;
 * dL/d control points of the one-launch form sums 8-term fp32 chains in fp64 (the two-kernel form: ~12-term chains), the
 * two agree within 5e-5 of the largest entry.  Which form runs depends on these arguments only: in1 == NULL, score == NULL,
 * p_xy == NULL (p_xy means the TPS_PP table layout, whose p_hat lacks the [1, x, y] columns the one-launch form reads from the
 * transposed classic table), p_hat_t given and 16-byte aligned, C0 <= 3, F <= 21, Ho*Wo in (1024, 4096] and a multiple of 4,
 * the fp64 accumulator, and the TPSPP_BWD_TWO_KERNELS flag bit / environment variable unset.
 * replaces: autograd through backbones/tps_pp/tps_pp.py:467-496,597-615;
 *           preprocessor/tps_preprocessor.py:71-83,270-282
 */
size_t tpspp_warp_bwd_workspace_floats(int N, int Ho, int Wo);
/* How dL/d input is accumulated in LDS when whole planes fit.  Default = fp64 LDS atomics: every term's fp32 bits are kept
 * whatever the spread of magnitudes (a non-finite gradient poisons the four taps it touches, as ATen does), but the fp64 sum
 * depends on the order in which the atomics arrive (2^-53 relative per add), so g_in0 / g_in1 are NOT bitwise reproducible
 * from run to run: after the rounding to fp32 two runs differ by at most one fp32 ulp, and only where the fp64 sum lies
 * within ~2^-50 of an fp32 rounding tie (tests/test_gpu_backward.py bounds it).  TPSPP_BWD_FIXED_POINT in `table_flags`
 * selects, for that call only, round 3's 64-bit fixed point: order-independent, hence bitwise reproducible, exact within
 * 2^-50 of a pass's largest |g| (a non-finite gradient turns the whole plane NaN).  g_ctrl / g_score are computed in a fixed
 * order in either mode.  tpspp_warp_bwd_set_accumulator(1) makes the fixed point the process-wide default of calls that do
 * not pass the flag (kept for scripts/bench_backward.py; prefer the flag: it is per call and per stream). */
int tpspp_warp_bwd_set_accumulator(int fixed_point);
int tpspp_warp_bwd(const float* g_out0, const float* in0, int C0, int H0, int W0,
                   const float* g_out1, const float* in1, int C1, int H1, int W1,
                   const float* grid, const float* T, const float* inv_delta_c,
                   const float* p_hat, int p_hat_ld, const float* p_xy, const float* score,
                   const float* p_hat_t_or_null, int table_flags, int N, int F, int Ho, int Wo,
                   float* g_in0, float* g_in1, float* g_ctrl, float* g_score,
                   float* g_grid_ws, size_t g_grid_ws_floats, tpspp_stream_t stream);

/*
 * out = act(conv2d(cat_c(up(src_0), up(src_1), up(src_2)), W) + bias [+ residual]) [+ residual]
 *       [* post_scale + post_shift]
 * fp32, NCHW, 1x1 or 3x3 kernel with "same" padding ((K-1)/2), stride (sh, sw), on the fp32 matrix
 * cores (exact fp32 products, fp32 accumulation).
 *   src_ptrs[i]   (N, C_i, H_i, W_i); src_dims + 5*i = {C_i, H_i, W_i, uh_i, uw_i}: source i is
 *                 nearest-upsampled by (uh_i, uw_i) on the fly; all sources share the logical size;
 *                 with nsrc > 1 every C_i must be a multiple of 32 (1x1) / 8 (3x3)
 *   weight_t      (Cin*KH*KW, Cout) = the PyTorch weight (Cout, Cin, KH, KW) flattened and
 *                 transposed once by the caller (BatchNorm, if any, folded in): generic kernel
 *   weight_tiled  the same values arranged (chunk, tap, channel-in-chunk, cout) with
 *                 KC = tpspp_conv_chunk_channels(K) channels per chunk, zero-padded to whole
 *                 chunks: [ceil(Cin/KC)][KH*KW][KC][Cout]; enables the tiled kernel (either pointer
 *                 may be NULL; results agree to fp32 rounding, the summation order differs)
 *   bias          (Cout) or NULL;  residual (N, Cout, Ho, Wo) or NULL
 *   res_mode      0 none, 1 act(conv + bias) + residual, 2 act(conv + bias + residual)
 *   relu          activation code: 0 none, 1 ReLU, 2 GELU in its exact erf form (nn.GELU / mmcv.GELU,
 *                 common/modules/transformer_module.py:116,121)
 *   post_scale/post_shift (Cout) or NULL: per-channel affine applied last (an eval-mode BatchNorm that
 *                 FOLLOWS the activation: backbones/nrtr_modality_transformer.py:42-48)
 * replaces: mmcv ConvModule / nn.Conv2d (+ nn.Upsample, torch.cat, skip additions)
 *           backbones/tps_pp/tps_pp.py:126-131,149-154,156-169,538-552,560-562;
 *           preprocessor/tps_preprocessor.py:101-128; backbones/resnet_v2_large.py:131-135;
 *           layers/conv_layer.py:12-33; backbones/nrtr_modality_transformer.py:19-48
 */
int tpspp_conv2d_fwd(const float* const* src_ptrs, const int* src_dims, int nsrc,
                     const float* weight_t, const float* weight_tiled, const float* bias,
                     const float* residual, const float* post_scale, const float* post_shift,
                     int res_mode, int relu, int N, int Cout, int KH, int KW, int sh, int sw,
                     float* out, int Ho, int Wo, tpspp_stream_t stream);

/*
 * DGAB block of the TPS++ regressor on a (N, C, 16, 64) feature map x with the point features
 * y (N, C, 32) [= en_feat (N, C, 2, 16) viewed per channel]:
 *     xn = LayerNorm_(16,64)(x);  A = gated attention(xn, y);  x1 = x + proj(A);
 *     out = x1 + fc2(gelu(fc1(LayerNorm_(16,64)(x1))))          (proj / fc1 / fc2 act along W)
 *   ln*_w/b (16,64); mlp_w_t (96,65) and mlp_h_t (48,17) = the bias-free Linear weights transposed;
 *   proj_slab [64][64], fc1_slab [4][64][64], fc2_slab [4][64][64]: Linear weights permuted into the
 *   MFMA k-slot order (slot 2*ks+half holds input feature 32*(ks>>4)+(ks&3)+8*((ks&15)>>2)+4*half);
 *   fc1 split into 4 blocks of 64 hidden units; biases in natural order; scratch (N,C,16,64).
 * replaces: backbones/tps_pp/DGAB.py:25-77 as called at backbones/tps_pp/tps_pp.py:318-319
 */
int tpspp_dgab_fwd(const float* x, const float* y, const float* ln1_w, const float* ln1_b,
                   const float* mlp_w_t, const float* mlp_h_t, const float* proj_slab,
                   const float* proj_b, const float* ln2_w, const float* ln2_b,
                   const float* fc1_slab, const float* fc1_b, const float* fc2_slab,
                   const float* fc2_b, float* scratch, float* out, int N, int C,
                   tpspp_stream_t stream);

/*
 * tpspp_dgab_fwd with the three Linear layers of the chain on the bf16 matrix cores (the bf16 configuration):
 * x, LayerNorms, gates, softmaxes, GELU and both residual sums in fp32; the gated map, the normalised x1 and the
 * GELU output are rounded to bf16 once each as matrix operands; erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7).
 *   proj_slab [4 k-steps][2][64 out][8] bf16 = Wp[out][16 j + 8 h + e];
 *   fc1_slab  [4 blocks][4][2][64][8] = W1[64 b + m][16 j + perm[8 h + e]];
 *   fc2_slab  [4 blocks][4][2][64][8] = W2[m][64 b + 16 j + perm[8 h + e]];  perm as in tpspp_front_bf16_fwd;
 *   scratch   (N,C,16,64) bf16;  everything else as tpspp_dgab_fwd.
 *   split3: the three-term "bf16x3" split (see tpspp_conv2d_bf16_fwd): scratch is then fp32 and every slab holds its hi
 *   and lo halves back to back ([..][hi|lo][4 k-steps][2][64][8]); results within ~1e-5 of tpspp_dgab_fwd.
 * replaces: backbones/tps_pp/DGAB.py:25-77
 */
int tpspp_dgab_bf16_fwd(const float* x, const float* y, const float* ln1_w, const float* ln1_b,
                        const float* mlp_w_t, const float* mlp_h_t, const void* proj_slab,
                        const float* proj_b, const float* ln2_w, const float* ln2_b,
                        const void* fc1_slab, const float* fc1_b, const void* fc2_slab,
                        const float* fc2_b, void* scratch, float* out, int N, int C, int split3,
                        tpspp_stream_t stream);

/*
 * The pointwise front of TPS_PP (ResNet45v2 wiring) fused: feat0 = relu(W0 outs0 + b0),
 * feat1 = relu(W1 outs1 + b1) at (H, W); feat2 = relu(W2 x + b2) at (H/2, W/2);
 * feat_grid = relu(Wg cat(feat0, feat1, Upsample2(feat2)) + bg).
 *   outs0/outs1 (N,32,H,W), x (N,64,H/2,W/2); w0/w1_slab [32][64], w2_slab [64][64] = the 1x1 weights
 *   transposed; wg_slab [3][64 k-slots (MFMA order, see tpspp_dgab_fwd)][64]; outputs (N,64,...).
 * replaces: backbones/tps_pp/tps_pp.py:560-562,581-585 (down0, down1, down2, up_sample + cat + down_feat)
 */
int tpspp_front_fwd(const float* outs0, const float* outs1, const float* x,
                    const float* w0_slab, const float* b0, const float* w1_slab, const float* b1,
                    const float* w2_slab, const float* b2, const float* wg_slab, const float* bg,
                    float* feat0, float* feat1, float* feat2, float* feat_grid,
                    int N, int H, int W, tpspp_stream_t stream);

/*
 * Attention score: score_t[b, pt, px] = tanh(scale * sum_j f[b, j, px] * p[b, pt, j]) with
 *   f = W2 (W1 de_feat[b, :, px] + b1) + b2   (feat_linear: Linear 64->32, Linear 32->128, no activation)
 *   de_feat (N, 64, n); w1_slab [64][32] = W1 transposed; w2_slab [32 k-slots][128] = W2 columns in
 *   the MFMA order (slot 2*ks+half <- input feature (ks&3) + 8*(ks>>2) + 4*half), transposed;
 *   p (N, 32, 128) = p_linear(point features); score_t (N, 32, n) -- the (N, F, n) layout that
 *   tpspp_warp_fwd takes with TPSPP_SCORE_TRANSPOSED.
 * replaces: backbones/tps_pp/tps_pp.py:293-312 (get_score / atten_score)
 */
int tpspp_score_fwd(const float* de_feat, const float* w1_slab, const float* b1, const float* w2_slab,
                    const float* b2, const float* p, float scale, float* score_t, int N, int n,
                    tpspp_stream_t stream);

/*
 * tpspp_score_fwd with the three-term "bf16x3" split (see tpspp_conv2d_bf16_fwd) in its three matrix products: fp32
 * tensors in and out, ~5e-6 of scale before the tanh.
 *   w1_slab [hi|lo][4 k-steps][2][32 out][8] bf16 = W1[out][16 j + 8 h + e];
 *   w2_slab [hi|lo][2 k-steps][2][128 out][8]     = W2[out][16 j + perm[8 h + e]]  (perm as in tpspp_front_bf16_fwd)
 * replaces: backbones/tps_pp/tps_pp.py:293-312
 */
int tpspp_score_x3_fwd(const float* de_feat, const void* w1_slab, const float* b1,
                       const void* w2_slab, const float* b2, const float* p, float scale,
                       float* score_t, int N, int n, tpspp_stream_t stream);

/*
 * CBAM(64, ratio 16) on the (N, 64, 2, 16) bottleneck map: mlp0_w (4,64), mlp2_w (64,4) = the
 * bias-free shared 1x1 MLP; sp_w (1,2,3,3), sp_b (1) = the spatial-attention conv.
 * replaces: backbones/tps_pp/tps_pp.py:27-82 as called at :163
 */
int tpspp_cbam_fwd(const float* x, const float* mlp0_w, const float* mlp2_w, const float* sp_w,
                   const float* sp_b, float* out, int N, tpspp_stream_t stream);

/*
 * Per-point stages on en_feat (N, 64, 2, 16) [point t = flattened (2,16) index]:
 *   ctrl (N,32,2) = fc2( relu(fc1b( relu(fc1a(en[:, t])) )) flattened )     fc1a (256,64) fc1b (2,256) fc2 (64,64)
 *   p    (N,32,128) = pl1( pl0(en[:, t]) )                                   pl0 (32,64)   pl1 (128,32)
 * weights / biases in nn.Linear layout.
 * replaces: backbones/tps_pp/tps_pp.py:321-323 (localization_fc1/fc2) and :305 (p_linear)
 */
int tpspp_tpe_points_fwd(const float* en_feat, const float* fc1a_w, const float* fc1a_b,
                         const float* fc1b_w, const float* fc1b_b, const float* fc2_w,
                         const float* fc2_b, const float* pl0_w, const float* pl0_b,
                         const float* pl1_w, const float* pl1_b, float* ctrl, float* p, int N,
                         tpspp_stream_t stream);

/*
 * out (N, C, H/2, W/2) = MaxPool2d(kernel 2, stride 2)(in);  out (N, C) = AdaptiveAvgPool2d(1)(in)
 * replaces: preprocessor/tps_preprocessor.py:110,114,118,126 (LocalizationNetwork.conv)
 */
int tpspp_maxpool2x2_fwd(const float* in, int N, int C, int H, int W, float* out, tpspp_stream_t stream);
int tpspp_global_avgpool_fwd(const float* in, int N, int C, int H, int W, float* out,
                             tpspp_stream_t stream);

/* Channels per K-chunk of the tiled conv kernel for a 1x1 / 3x3 kernel (layout of weight_tiled). */
int tpspp_conv_chunk_channels(int kernel_size);

/*
 * tpspp_front_fwd on the bf16 matrix cores (the bf16 configuration): inputs and feat0 / feat1 / feat2 are bf16,
 * feat_grid bf16 or fp32 (feat_grid_f32 bit 0); accumulation, bias and ReLU in fp32; feat_grid is computed from the
 * bf16-rounded feat0 / feat1 / feat2 (what the separate convolutions would read back).  feat_grid_f32 bit 1 (value 2):
 * feat0 / feat1 / feat2 are written in the BLOCKED layout (N, 8, H, W, 8) -- the eight channels of a group next to
 * each other per pixel -- that tpspp_conv2d_bf16_fwd takes with layout code 2 (same values; feat_grid stays NCHW).
 *   w0 / w1  [2 k-steps][2 halves][64 cout][8] bf16 with value W[cout][16 j + 8 h + e];  w2 the same with 4 k-steps;
 *   wg       [12][2][64][8] with value Wg[cout][16 j + perm[8 h + e]], perm = {0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15}
 *            (the order in which the matrix core's result registers come back as the next operand);
 *   b0 / b1 / b2 / bg (64) fp32.  Needs H even and W a multiple of 32.
 *   split3: the three-term "bf16x3" split on fp32 tensors -- inputs and all four outputs fp32, every slab holds its hi
 *   and lo halves ([hi|lo][k-steps][2][64][8]), feat0 / feat1 / feat2 are chained without an intermediate rounding;
 *   feat_grid_f32 bit 1 then means the fp32 blocked layout (N, 8, H, W, 8) (tpspp_conv2d_bf16_fwd's code 3).
 * Non-finite values: the ReLU of the bf16 form is a signed 16-bit max on the rounded pair (identical bits for every finite
 * value and for +-inf; a NaN with the sign bit clear propagates like torch.relu's, a NaN with the sign bit set becomes +0);
 * the fp32 / three-term forms and tpspp_down_fused_* use fmaxf(x, 0), which sends every NaN to 0.  NaN inputs are outside
 * the parity contract of all of them (tests/test_gpu_conv_bf16.py pins this behaviour so that a change is noticed).
 * replaces: backbones/tps_pp/tps_pp.py:560-562,581-585
 */
int tpspp_front_bf16_fwd(const void* outs0, const void* outs1, const void* x,
                         const void* w0, const float* b0, const void* w1, const float* b1,
                         const void* w2, const float* b2, const void* wg, const float* bg,
                         void* feat0, void* feat1, void* feat2, void* feat_grid, int feat_grid_f32,
                         int N, int H, int W, int split3, tpspp_stream_t stream);

/*
 * A Linear layer over channel-major tokens on the bf16 matrix cores:  out (Co, M) = act(W^T X + bias) [+ res]
 *   X (K, M) fp32 (rounded to bf16 -- split3: split into hi + lo -- as it is staged), M = images x tokens;
 *   w_arranged: the (Co, K, 1, 1) weight arranged as tpspp_conv2d_bf16_fwd takes a 1x1 kernel (split3: with hi and lo slabs);
 *   bias (Co) or NULL; res (Co, M) fp32 or NULL (added after the activation); act 0 none, 2 GELU (erf);
 *   out (Co, M) fp32 (out_f32 != 0) or bf16.  K % 32 == 0, Co % 128 == 0, M % 4 == 0; fp32 accumulation over k ascending.
 * The recogniser head's encoder projections and the decoder's one-off key / value projections in the bf16 / bf16x3
 * configurations go through this kernel (tpspp_nrtr_encoder_fwd / tpspp_nrtr_decoder_fwd with TPSPP_HEAD_BF16 / _BF16X3).
 * replaces: nn.Linear inside MultiHeadAttention / PositionwiseFeedForward, common/layers/transformer_layers.py:36-75
 */
int tpspp_token_gemm_bf16_fwd(const float* X, const void* w_arranged, const float* bias, const float* res,
                              void* out, int out_f32, int K, int Co, int M, int act, int split3,
                              tpspp_stream_t stream);

/*
 * down0 + down0_1 (or down1 + down1_1) of the ResNet45v2 wiring in one kernel (bf16 configuration):
 *     out = bf16(relu(conv3x3 stride 2 pad 1 (bf16(relu(W0 in + b0))) + bd))
 * without the intermediate (N, 64, H, W) map in HBM; bit for bit what tpspp_front_bf16_fwd's feat0 / feat1 followed by
 * tpspp_conv2d_bf16_fwd (3x3, stride 2, blocked output) give.
 *   in (N, 32, H, W) bf16, W = 128, H even;  w0 / b0 as tpspp_front_bf16_fwd takes them;  wd = the (64, 64, 3, 3) weight
 *   arranged as tpspp_conv2d_bf16_fwd takes it ([4 chunks][9 taps][2][64 cout][8] bf16), bd (64) fp32;
 *   out: the blocked layout (N, 8, H/2, 64, 8) bf16 (tpspp_conv2d_bf16_fwd's layout code 2).
 * tpspp_front_bf16_fwd may then be called with feat0 = feat1 = NULL (blocked mode): it keeps computing them as operands of
 * feat_grid and stores neither.
 * replaces: backbones/tps_pp/tps_pp.py:560-563 (self.down0_1(self.down0(outs[0])), self.down1_1(self.down1(outs[1])))
 */
int tpspp_down_fused_bf16_fwd(const void* in, const void* w0, const float* b0, const void* wd, const float* bd,
                              void* out, int N, int H, int W, int relu, tpspp_stream_t stream);

/*
 * The same for the three-term "bf16x3" split: in (N, 32, H, 128) fp32, w0 / wd with their hi and lo slabs (as
 * tpspp_front_bf16_fwd / tpspp_conv2d_bf16_fwd take them with split3), out the fp32 blocked layout (N, 8, H/2, 64, 8)
 * (layout code 3); bit for bit tpspp_front_bf16_fwd(split3)'s feat0 / feat1 followed by the three-term 3x3 stride-2
 * convolution.  tpspp_front_bf16_fwd(split3, blocked) accepts feat0 = feat1 = NULL as well.
 * replaces: backbones/tps_pp/tps_pp.py:560-563
 */
int tpspp_down_fused_x3_fwd(const float* in, const void* w0, const float* b0, const void* wd, const float* bd,
                            float* out, int N, int H, int W, int relu, tpspp_stream_t stream);

/*
 * The same for the exact-fp32 configuration: in (N, 32, H, 128) fp32, w0_slab [32][64] (tpspp_front_fwd's), wd_tiled the
 * (64, 64, 3, 3) weight as tpspp_conv2d_fwd's weight_tiled ([16 chunks][9 taps][4][64]), out (N, 64, H/2, 64) fp32 NCHW; bit for
 * bit tpspp_front_fwd's feat0 / feat1 followed by tpspp_conv2d_fwd (3x3, stride 2).  tpspp_front_fwd accepts
 * feat0 = feat1 = NULL.
 * replaces: backbones/tps_pp/tps_pp.py:560-563
 */
int tpspp_down_fused_f32_fwd(const float* in, const float* w0_slab, const float* b0, const float* wd_tiled,
                             const float* bd, float* out, int N, int H, int W, int relu, tpspp_stream_t stream);

/*
 * The same fused convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: bf16 operands, fp32
 * accumulation; bias / residual / activation / affine in fp32) for the bf16 configurations
 * (BASELINE.json configs[2], configs[4]).  Every tensor is independently bf16 NCHW (layout code 0), fp32 NCHW (1) or
 * bf16 BLOCKED (2): (N, C/8, H, W, 8), the eight channels of a group next to each other per pixel -- a 16-byte unit
 * of that layout IS a unit of the kernel's channel-innermost LDS patch, so a blocked source is staged with 16-byte
 * loads and no transposition, and a blocked output leaves as 8-byte pieces straight from the result registers
 * (layers that only feed other convolutions use it; C a multiple of 8; not with split3); with split3 there is code 3,
 * the same blocked shape in fp32 (sources, residual, output: a patch position's 8 channels are two 16-byte loads).
 * Same values in every layout.
 *   src_ptrs[i]      (N, C_i, H_i, W_i); src_dims + 6*i = {C_i, H_i, W_i, uh_i, uw_i, layout code}; uh, uw in
 *                    {1, 2, 4}; with nsrc > 1 every C_i must be a multiple of
 *                    KC = tpspp_conv_bf16_chunk_channels(K)
 *   weight_arranged  bf16, the PyTorch weight (Cout, Cin, KH, KW) (BatchNorm folded) zero-padded to
 *                    Cout % 64 == 0 and Cin % KC == 0 and laid out
 *                    [Cout/64][Cin/KC][KH*KW][KC/8][64 cout][8 cin]  (the LDS image of each K-chunk)
 *   bias, post_scale, post_shift   (Cout) fp32 or NULL
 *   residual         (N, Cout, Ho, Wo), layout code residual_f32, or NULL; res_mode as above
 *   relu             0 none, 1 ReLU, 2 GELU (exact erf form)
 *   out              (N, Cout, Ho, Wo), layout code out_f32 (bf16: round to nearest even)
 *   split3           "bf16x3": every fp32 operand is split into bf16 halves (hi = bf16(x), lo = bf16(x - hi)) and a
 *                    product is hi*hi + hi*lo + lo*hi in fp32: ~5e-6 relative error per layer (fp32: 3e-7, bf16:
 *                    2.5e-3) at three bf16 matrix instructions.  weight_arranged then holds TWO slabs per chunk,
 *                    [Cout/64][Cin/KC][hi|lo][KH*KW][KC/8][64][8]; fp32 sources are split as they are staged
 * replaces: the call sites listed for tpspp_conv2d_fwd when the module runs in bf16.
 */
int tpspp_conv2d_bf16_fwd(const void* const* src_ptrs, const int* src_dims, int nsrc,
                          const void* weight_arranged, const float* bias,
                          const void* residual, int residual_f32,
                          const float* post_scale, const float* post_shift,
                          int res_mode, int relu, int N, int Cout, int KH, int KW, int sh, int sw,
                          void* out, int out_f32, int Ho, int Wo, int split3, tpspp_stream_t stream);

/* Channels per K-chunk of the bf16 conv kernel for a 1x1 / 3x3 kernel (layout of weight_arranged). */
int tpspp_conv_bf16_chunk_channels(int kernel_size);

/* Tuning / testing (results do not depend on it).  Bit 0: tpspp_conv2d_fwd uses its generic kernel even when
 * weight_tiled is given.  Bit 1: tpspp_conv2d_bf16_fwd does not use the persistent kernel for blocked 3x3 layers.
 * Bit 2: ... nor the wide-tile kernel (tpspp_conv3_wide.hip) for the blocked 3x3 layers with >= 128 output channels. */
int tpspp_conv_set_tuning(int flags);

/*
 * Launch-shape override for tpspp_warp_fwd (tuning / benchmarking only; results do not depend on
 * it): images per workgroup and threads per workgroup of the gather kernel (0 = heuristic), and
 * kernel choice: 0 = automatic, 1 = force the gather kernel, 2 = require the LDS-staged kernel
 * (TPSPP_EINVAL if the shape does not qualify), 3 = as 2 but ignore TPSPP_TABLE_MIRROR4,
 * 4 = require the plane-streaming kernel, 5 = require the image-pair kernel (classic 32x100 geometry with a
 * tpspp_prepare_mirror_table buffer), 6 = require the in-place kernel (32x100, 32x128, 48x160, 32x64 with C = 1 or 3,
 * same buffer), 7 = require a run-time-geometry kernel (C = 1, 3 or 4, F = 20, same buffer): the in-place kernel where one
 * workgroup covers the image, else row bands with span staging, 8 = require the span-staging kernel (geometries whose
 * prepared table has the third section: tpspp_prepared_table_floats), 9 = require the in-place kernel with run-time
 * geometry in its banded form as well (every band stages the whole image: round 4's kernel, kept for comparisons);
 * bands = workgroups per image pair in the LDS-staged kernel, 0..8 (0 = heuristic); with kernel_choice 7 / 9: bits 0-2 =
 * workgroups per image (0 = heuristic), bit 3 = never an image pair per workgroup; with kernel_choice 8: bits 0-5 =
 * workgroups per image (0 = heuristic), bit 6 = every workgroup on its global-memory path, bit 7 = measure each band's
 * row span before staging it (the first form) instead of requesting the band's rows +- 2 at launch, bits 8-15 = LDS budget
 * per workgroup in KB (0 = 38 / 40: four workgroups per CU).
 */
int tpspp_warp_set_tuning(int images_per_group, int threads_per_group, int kernel_choice, int bands);

/*
 * Diagnostics: when device_buf is non-NULL the LDS-staged kernel writes 8 int64 shader-clock stamps
 * per workgroup (start, T ready, grid expanded, images in LDS, stores retired, DMA issued, DMA
 * landed, unused) into device_buf[workgroup * 8 + i]; the image-pair kernel writes ticks since kernel entry
 * of (T ready, tap descriptors done, image A landed, A staged, image B landed, B staged, stores retired) and
 * the 100 MHz wall clock at entry.  NULL (default) disables it.
 */
int tpspp_warp_set_trace(long long* device_buf);
/* Diagnostics of the persistent decoder step (tpspp_nrtr_decoder_fwd, reduced-precision heads): when non-NULL every workgroup
 * writes the 100 MHz wall clock at the end of each of its phases, device_buf[(step * workgroups + workgroup) * 64 + phase]
 * (phase 0 = entry; 8 phases per layer, then the classifier); the buffer must hold steps x workgroups x 64 int64.  NULL
 * (default) disables it. */
int tpspp_head_set_trace(long long* device_buf);
/* Lab / test hook: occupies the device for a while -- `workgroups` workgroups of one wavefront, each holding `lds_bytes` of LDS
 * (>= 64 KB keeps a workgroup of the persistent decoder, ~103 KB, off that CU), spin for `milliseconds` of wall-clock time
 * (<= 2000) on `stream`.  tests/test_gpu_head.py uses it to decode on a partly occupied device. */
int tpspp_lab_occupy(int workgroups, int lds_bytes, int milliseconds, tpspp_stream_t stream);

/* ===== Recogniser head after TPS++: NRTR encoder / decoder ==========
 *
 * Layout convention of the head: activations are CHANNEL-MAJOR matrices X[c][m] (row = feature,
 * column = token; m = image * T + token), fp32, dense.  Linear weights are passed K-MAJOR: the
 * PyTorch weight (out_features, in_features) transposed once by the caller to (in_features,
 * out_features).  d_k = d_v = 64 (n_head = d_model / 64), dropout is the identity (inference).
 */

/* out (cols, rows) = in (rows, cols) transposed. */
int tpspp_transpose2d(const float* in, int rows, int cols, float* out, tpspp_stream_t stream);

/*
 * y[c][m] = (x[c][m] - mean_m) / sqrt(var_m + eps) * gamma[c] + beta[c], statistics over the C rows of
 * column m (biased variance): nn.LayerNorm(C) applied to every token of a channel-major (C, M) matrix.
 * replaces: common/layers/transformer_layers.py:44,46,103-105; encoders/nrtr_encoder.py:49;
 *           decoders/nrtr_decoder.py:77
 */
int tpspp_layernorm_cm_fwd(const float* x, const float* gamma, const float* beta, int C, int M, float eps,
                           float* y, tpspp_stream_t stream);

/*
 * Multi-head self-attention of the encoder on already projected q/k/v:
 *   qkv (3C, N*T) channel-major: rows [0,C) = q, [C,2C) = k, [2C,3C) = v; head h = rows [64h, 64h+64)
 *   out (C, N*T):  softmax_j(q_i . k_j / 8  masked to j < valid_len[b]) . v_j   per image b and head
 *   valid_len (N) device int32 or NULL (no mask); T <= 256.
 * replaces: common/modules/transformer_module.py:24-33,82-93 (ScaledDotProductAttention inside
 *           MultiHeadAttention) with the key mask of encoders/nrtr_encoder.py:51-65
 */
int tpspp_attn_enc_fwd(const float* qkv, int N, int C, int T, const int* valid_len, float* out,
                       tpspp_stream_t stream);

/*
 * Fused LayerNorm -> Linear on a channel-major activation x (K, M), made for the few hundred columns of one
 * decoder step (any size is accepted; large M is better served by the two separate calls):
 *   token_major == 0:  out (Cout, M) = act(W^T LN(x) + bias) [+ residual]     act: 0 none, 1 ReLU, 2 GELU(erf)
 *   token_major != 0:  out (M, Cout) = LN(x)^T W + bias                       (no activation / residual)
 * The LayerNorm's affine is folded into the weight by the caller, once:
 *   w_gamma (K, Cout) = diag(gamma) W_kmajor,   w_colsum (Cout) = column sums of w_gamma,
 *   bias_eff (Cout)   = beta^T W_kmajor (+ bias)  or NULL when that is zero,
 * so that  W^T LN(x)[:, m] = rstd_m (w_gamma^T x[:, m] - mean_m w_colsum) + bias_eff  and the kernel multiplies
 * the raw activation (mean / rstd of every column are accumulated from the same loads, biased variance, eps).
 * replaces: `norm` followed by a projection, common/layers/transformer_layers.py:150-163
 */
int tpspp_linear_ln_fwd(const float* x, int K, int M, float eps, const float* w_gamma,
                        const float* w_colsum, int Cout, const float* bias_eff, int act,
                        const float* residual, int token_major, float* out, tpspp_stream_t stream);

/*
 * ResizeOCR + ToTensorOCR + NormalizeOCR on the GPU: N uint8 HWC crops of different
 * sizes, packed in one device buffer, -> out (N, C, H, W) fp32.  Image n (src_h[n] x src_w[n] x C at
 * src_packed + src_offsets[n]) is resized to H x resize_w[n], columns >= resize_w[n] hold pad_value, and every
 * byte v of channel c becomes lut[c*256 + v] (the caller tabulates (v/255 - mean[c]) / std[c] in fp32).
 * The widths come from the host logic of ResizeOCR.__call__ (tps_pp_amd/ocr_transforms.py).
 *   interpolation   ResizeOCR's `backend` (ocr_transforms.py:34-36,46,65 -> mmcv.imresize(..., backend=)):
 *     TPSPP_RESIZE_CV2 (0)     backend None / 'cv2': OpenCV's 8-bit INTER_LINEAR arithmetic (11-bit fixed-point weights;
 *                              INTER_AREA for an exact 2x2 shrink).  PARITY UNPINNED against OpenCV (absent at build time);
 *                              bit-exact against oracle/resize_oracle.py.
 *     TPSPP_RESIZE_PILLOW (1)  backend 'pillow': Image.resize(size, Image.BILINEAR) on uint8 -- Pillow's Resample.c (double
 *                              coefficients of the support-scaled triangle filter, 22-bit fixed point, horizontal pass into
 *                              uint8, then the vertical pass).  PINNED: bit-exact against the installed Pillow's outputs
 *                              (tests/golden/resize_pillow.npz, written by tests/golden/make_resize_golden.py).
 * replaces: mmocr/datasets/pipelines/ocr_transforms.py:67-156 (mmcv.imresize + mmcv.impad, TF.to_tensor, TF.normalize)
 */
#define TPSPP_RESIZE_CV2    0
#define TPSPP_RESIZE_PILLOW 1
int tpspp_resize_normalize_fwd(const unsigned char* src_packed, const long long* src_offsets,
                               const int* src_h, const int* src_w, const int* resize_w,
                               const float* lut, int pad_value, int N, int C, int H, int W,
                               float* out, int interpolation, tpspp_stream_t stream);

/*
 * flags bit of tpspp_nrtr_encoder_fwd / tpspp_nrtr_decoder_fwd (the bf16 configuration, BASELINE.json configs[4]):
 * the wide projections run on the bf16 matrix cores -- encoder: wqkv / fc / w1 / w2; decoder: the one-off key / value
 * projections of the encoder output -- and their entries of layer_ptrs then point to bf16 weights arranged as
 * tpspp_conv2d_bf16_fwd wants a 1x1 kernel; the decoder keeps the encoder keys / values as bf16 (they are the only
 * HBM-bound operand of a decoding step).  Residual stream, LayerNorms, softmaxes, the per-step projections and the
 * classifier stay fp32.
 */
#define TPSPP_HEAD_BF16 1
/* The same projections with the "bf16x3" split of tpspp_conv2d_bf16_fwd (fp32 tensors, ~5e-6 per layer): encoder wqkv / fc /
 * w1 / w2 and the decoder's key projection of the encoder output; their layer_ptrs entries point to hi+lo arranged weights.
 * Keys / values stay fp32. */
#define TPSPP_HEAD_BF16X3 2

/* Scratch sizes (bytes) for the two calls below; 0 on bad arguments. */
size_t tpspp_nrtr_encoder_workspace(int N, int C, int T, int d_inner);
size_t tpspp_nrtr_decoder_workspace(int N, int C, int T, int d_inner, int n_layers, int max_seq_len,
                                    int num_out);

/*
 * NRTREncoder.forward: feat (N, C, H, W) viewed as (N, C, T = H*W) -> LayerNorm(layer_stack(tokens)).
 *   layer_ptrs   n_layers x 12 device pointers, per layer in this order (K-major weights, NULL = absent bias):
 *                norm1.weight, norm1.bias, [linear_q|linear_k|linear_v].weight^T concatenated (C, 3C),
 *                its bias (3C) | NULL, fc.weight^T (C, C), fc.bias | NULL, norm2.weight, norm2.bias,
 *                mlp.w_1.weight^T (C, d_inner), mlp.w_1.bias, mlp.w_2.weight^T (d_inner, C), mlp.w_2.bias
 *   ln_g, ln_b   the encoder's final layer_norm
 *   valid_len    (N) int32 device or NULL: number of valid tokens per image = min(T, ceil(T * valid_ratio))
 *   out_cm       (C, N*T) channel-major result or NULL;  out_ntc (N, T, C) (the reference's return) or NULL
 * Layer order ('norm','self_attn','norm','ffn'), activation GELU (erf).
 * replaces: textrecog/encoders/nrtr_encoder.py:67-87, common/layers/transformer_layers.py:57-75
 */
int tpspp_nrtr_encoder_fwd(const float* feat, int N, int C, int T, int d_inner, int n_layers,
                           const float* const* layer_ptrs, const float* ln_g, const float* ln_b,
                           const int* valid_len, void* workspace, size_t workspace_bytes,
                           float* out_cm, float* out_ntc, int flags, tpspp_stream_t stream);

/*
 * NRTRDecoder.forward_test (greedy, forced_tokens == NULL) / forward_train (teacher forcing):
 *   enc_cm       (C, N*T) channel-major encoder output (tpspp_nrtr_encoder_fwd's out_cm)
 *   layer_ptrs, layer_ptrs_len   the table and its length in pointers, which must be n_layers x 24 + 1 (checked: a
 *                table in the 12- or 18-pointer layout of earlier rounds is refused, not read out of bounds);
 *                per layer (every LayerNorm folded into the projection that
 *                follows it, see tpspp_linear_ln_fwd: w_gamma = diag(norm.weight) W^T-k-major, colsum its
 *                column sums, bias_eff = norm.bias^T W (+ bias)):
 *                  self_attn q|k|v fused (C, 3C): w_gamma, colsum (3C), bias_eff (3C);
 *                  self_attn.fc.weight^T (C, C), its bias | NULL;
 *                  enc_attn.linear_q (C, C): w_gamma, colsum (C), bias_eff (C);
 *                  enc_attn.linear_k.weight^T, its bias | NULL;  enc_attn.linear_v.weight^T (no bias);
 *                  enc_attn.fc.weight^T, its bias | NULL;
 *                  mlp.w_1 (C, d_inner): w_gamma, colsum (d_inner), bias_eff (d_inner);
 *                  mlp.w_2.weight^T (d_inner, C), mlp.w_2.bias;
 *                  then six entries (or NULL: the round-2 channel-major step kernels are used) with the six per-step
 *                  projections (q|k|v with norm1 folded, self fc, enc q with norm2 folded, enc fc, w_1 with norm3 folded,
 *                  w_2) arranged in MFMA fragment order for the token-major step GEMM, Co zero-padded to a multiple of 32:
 *                  exact fp32 (no flag):  [Co/32][K/8][2 k halves][32 outputs][4 k] fp32, k = 8 u + 4 half + e;
 *                  TPSPP_HEAD_BF16 / _BF16X3: hi = bf16(w), lo = bf16(w - hi),
 *                  [Co/32][K/16][hi|lo][2 k halves][32 outputs][8 k] bf16 (products hi*hi + hi*lo + lo*hi, fp32
 *                  accumulation: inside the fp32 tolerance);
 *                one more pointer follows the last layer: the classifier (final layer_norm folded) arranged the same way
 *   emb (num_classes, C) trg_word_emb.weight;  pos_table (n_position, C) position_enc.position_table;
 *   w_cls (C, num_out), cls_colsum (num_out), b_cls (num_out): the classifier with the final layer_norm
 *                (eps 1e-6) folded in the same way;  max_seq_len <= 64;  valid_len as above (cross-attention
 *                key mask) or NULL
 *   forced_tokens NULL: greedy decoding from start_idx; out (N, max_seq_len, num_out) = per-step softmax
 *                scores, tokens_out (N, max_seq_len + 1) int32 (or NULL) = <BOS> followed by the arg-max
 *                of every step.
 *                non-NULL (N, max_seq_len) int32 padded targets: out = raw logits of every position
 *                (keys holding padding_idx are masked out of the self-attention).
 * One position per step against cached keys/values: same results as the reference's full re-run of the
 * padded sequence at every step, up to fp32 summation order.  Nothing synchronises with the host.
 *
 * How the steps run, and what that asks of the caller.  With d_model 512, 8 heads, num_out <= 128, <= 8 layers and the
 * arranged per-step weights present, the max_seq_len steps of up to 512 images run as ONE persistent launch: 16 workgroups
 * (one per CU, 512 threads, ~103 KB of LDS) per 32 images synchronise among themselves through device counters
 * (tpspp_head_persist.h).  That form has two requirements, both ENFORCED by this entry point rather than left to the caller:
 *   (1) co-residency: a group of 8 such clusters (128 workgroups) must be resident together.  The call queries the device's
 *       CU count and the kernel's occupancy; a device (or CU-masked / CPX partition) that cannot hold 128 workgroups gets the
 *       launch-per-phase pipeline instead (same scores: bit-identical for the exact-fp32 head, within 2e-5 with identical
 *       tokens for the reduced-precision heads), one that holds 128..255 gets 256 images per launch.  Under stream capture
 *       the launch pipeline is used as well.
 *   (2) one persistent decode in flight per device and process: every such call makes `stream` wait (hipStreamWaitEvent, no
 *       host synchronisation) for the previous persistent decode issued by this process on the same device, whatever stream
 *       that one ran on.  Decodes issued from several streams / threads are therefore safe and simply run one after the
 *       other; kernels of other streams may overlap a decode (its clusters wait for their CUs).  Decodes issued by ANOTHER
 *       PROCESS on the same device are not seen by this guard: one process per GPU, as everywhere in this library.
 * Failure is loud, never silent: if a cluster's barrier is not completed within TPSPP_HEAD_TIMEOUT_MS (environment,
 * wall-clock milliseconds, default 4000; only possible when (2) is violated from outside the process, or the device is held
 * by another kernel for that long), the scores of that cluster's images are NaN for EVERY step from the failing one to
 * max_seq_len - 1, and *status_out = 1; no step of `out` is left uninitialised and the call itself has long returned
 * TPSPP_OK (it is asynchronous).
 *   status_out   device int32 (or NULL): written on `stream` behind the decode -- 0 = every step completed, 1 = a barrier
 *                timed out (scores NaN as described).  Read it with the copy that fetches the scores / tokens; the Python
 *                wrapper (tps_pp_amd.nrtr_head) does and raises TpsppError.  TPSPP_HEAD_NO_PERSIST=1 (environment) forces
 *                the launch pipeline, which has no such failure mode (status 0).
 * replaces: textrecog/decoders/nrtr_decoder.py:95-113,131-177, common/layers/transformer_layers.py:133-163
 */
int tpspp_nrtr_decoder_fwd(const float* enc_cm, int N, int C, int T, int d_inner, int n_layers,
                           const float* const* layer_ptrs, int layer_ptrs_len,
                           const float* emb, const float* pos_table, int n_position,
                           const float* w_cls, const float* cls_colsum, const float* b_cls, int num_out,
                           int max_seq_len,
                           int start_idx, int padding_idx, const int* valid_len,
                           const int* forced_tokens, void* workspace, size_t workspace_bytes,
                           float* out, int* tokens_out, int* status_out, int flags, tpspp_stream_t stream);

/*
 * Layout change between bf16 convolutions and the sampler: in_blocked (N, C/8, HW, 8) bf16 (tpspp_conv2d_bf16_fwd's layout
 * code 2) -> out_nchw (N, C, HW) bf16.  C % 8 == 0, HW % 64 == 0, 16-byte aligned tensors.  Same values.
 * (The bf16 backbone's second stage ends on the persistent blocked kernel; the warp reads channel planes.)
 * replaces nothing in the reference (its maps are NCHW throughout).
 */
int tpspp_blocked_to_nchw_bf16(const void* in_blocked, int N, int C, int HW, void* out_nchw, tpspp_stream_t stream);

/*
 * AttnConvertor.tensor2idx on the device: scores (N, L, C) fp32 (the decoder's per-step soft-max scores, or any tensor of
 * that shape) -> per position the maximum (val_out (N, L)) and its first index, torch.max's tie and NaN rules; idx_out (N, L)
 * int32 holds that index where the reference's scan KEEPS the character -- position before the image's first end_idx and
 * index != padding_idx -- and -1 elsewhere.  The host then needs one copy of 2 x N x L words per batch.
 * replaces: textrecog/convertors/attn.py:124-140 (torch.max per image, two device->host copies per image, the Python scan)
 */
int tpspp_attn_tensor2idx_fwd(const float* scores, int N, int L, int C, int end_idx, int padding_idx,
                              int* idx_out, float* val_out, tpspp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TPSPP_H_ */
