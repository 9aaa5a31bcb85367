// bf16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16: bf16 operands,
// fp32 accumulation) for BASELINE.json configs[2] / [4] (the bf16 variants of the TPS++ regressor and the
// conv stem).  Same fusion surface as the fp32 kernel (tpspp_conv.hip): 1x1 / 3x3, strides 1 / 2 / (2,1),
// "same" padding, up to three channel-concatenated sources each nearest-upsampled by a power of two on the
// fly, bias + ReLU, residual before or after the activation, optional per-channel affine last.  Every
// source / the residual / the output is independently bf16 or fp32 in memory (NCHW), so the layers around
// the convolutions that must stay fp32 (control points, TPS solve, grid) exchange tensors without a
// conversion pass.
//
// GEMM view:  D[cout][pixel] = W[cout][k] X[k][pixel];  weights are the A operand, the image the B operand,
// so a half-wavefront holds 32 consecutive pixels of one output channel (coalesced NCHW rows).
//
// The 16x faster matrix pipe moves the bottleneck to operand delivery, hence three differences from the
// fp32 kernel:
//  * workgroup tile = 256 pixels x 64 channels, wavefront = 64 pixels x 64 channels (four 32x32
//    accumulators): one LDS fragment read per MFMA instead of 1.5;
//  * the LDS patch is CHANNEL-INNERMOST, [channel group of 8][position][8 channels]: the B fragment of a
//    lane (pixel p, k-half h) for one tap is ONE ds_read_b128 at lane_base + immediate (8 consecutive k =
//    8 channels of the tap), conflict-free because consecutive pixels are 16 B apart.  The NCHW -> channel-
//    innermost transposition happens in the staging registers: a thread owns patch positions, loads the
//    chunk's channels of that position (coalesced 2-B loads along the row, packed in pairs) and writes 16 B;
//  * weights are pre-arranged on the host as [cout tile][chunk][tap][k group][64 cout][8 k] bf16, i.e. the
//    global image of a chunk's slab IS its LDS image and an A fragment is one ds_read_b128.
// Next chunk's patch and slab are prefetched into registers during the multiply (one barrier pair per chunk).
//
// Replaces (reference, mmocr/models/textrecog/): the same call sites as tpspp_conv2d_fwd --
// backbones/tps_pp/tps_pp.py:126-131,149-154,156-169,538-552,560-562; backbones/resnet_v2_large.py:131-135;
// layers/conv_layer.py:12-33 -- when the module runs in bf16.
// Bound: operand delivery (LDS bandwidth) below the bf16 MFMA peak (2.5 PFLOP/s); 1x1 layers HBM.
#include "tpspp_conv_bf16_impl.h"

TPSPP_EXPORT int tpspp_conv_bf16_chunk_channels(int kernel_size)
{
    return kernel_size == 1 ? kKC1 : kKC3;
}

TPSPP_EXPORT int tpspp_conv2d_bf16_fwd(const void* const* src_ptrs, const int* src_dims, int nsrc,
                                       const void* weight_arranged, const float* bias,
                                       const void* residual, int residual_f32,
                                       const float* post_scale, const float* post_shift,
                                       int res_mode, int relu, int N, int Cout, int KH, int KW, int sh, int sw,
                                       void* out, int out_f32, int Ho, int Wo, int split3, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(src_ptrs && src_dims && weight_arranged && out, "tpspp_conv2d_bf16_fwd: null pointer");
    TPSPP_REQUIRE(nsrc >= 1 && nsrc <= 3, "tpspp_conv2d_bf16_fwd: 1..3 sources");
    TPSPP_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3), "tpspp_conv2d_bf16_fwd: kernel must be 1x1 or 3x3");
    TPSPP_REQUIRE(N >= 0 && Cout > 0 && sh >= 1 && sw >= 1 && Ho > 0 && Wo > 0, "tpspp_conv2d_bf16_fwd: bad sizes");
    TPSPP_REQUIRE(res_mode >= 0 && res_mode <= 2 && (res_mode == 0) == (residual == nullptr),
                  "tpspp_conv2d_bf16_fwd: residual / res_mode mismatch");
    TPSPP_REQUIRE(relu >= 0 && relu <= 2, "tpspp_conv2d_bf16_fwd: activation code must be 0 (none), 1 (ReLU) or 2 (GELU)");
    TPSPP_REQUIRE((post_scale == nullptr) == (post_shift == nullptr),
                  "tpspp_conv2d_bf16_fwd: post_scale/post_shift come together");
    BParams P;
    P.nsrc = nsrc;
    const int KC = (KH == 1) ? kKC1 : kKC3;
    int cin = 0, Hi = -1, Wi = -1;
    for (int i = 0; i < nsrc; ++i) {
        const int* d = src_dims + 6 * i;                      // C, H, W, uh, uw, 0 bf16 NCHW / 1 fp32 NCHW / 2 bf16 blocked
        TPSPP_REQUIRE(src_ptrs[i] && d[0] > 0 && d[1] > 0 && d[2] > 0, "tpspp_conv2d_bf16_fwd: bad source %d", i);
        TPSPP_REQUIRE((d[3] == 1 || d[3] == 2 || d[3] == 4) && (d[4] == 1 || d[4] == 2 || d[4] == 4),
                      "tpspp_conv2d_bf16_fwd: upsampling factors must be 1, 2 or 4");
        TPSPP_REQUIRE(nsrc == 1 || d[0] % KC == 0,
                      "tpspp_conv2d_bf16_fwd: concatenated sources need channel counts that are multiples of %d", KC);
        P.src[i].p = src_ptrs[i];
        P.src[i].C = d[0]; P.src[i].H = d[1]; P.src[i].W = d[2];
        P.src[i].lh = d[3] >> 1; P.src[i].lw = d[4] >> 1;      // 1,2,4 -> 0,1,2
        TPSPP_REQUIRE(d[5] >= 0 && d[5] <= 3 && (d[5] != 2 || (d[0] % 8 == 0 && !split3)) && (d[5] != 3 || (d[0] % 8 == 0 && split3)),
                      "tpspp_conv2d_bf16_fwd: source %d: layout code must be 0 / 1 / 2 / 3 (blocked: channels a multiple of 8; 2 bf16 "
                      "not with split3, 3 fp32 only with split3)", i);
        P.src[i].f32 = d[5];
        const int lh = d[1] * d[3], lw = d[2] * d[4];
        TPSPP_REQUIRE(Hi < 0 || (Hi == lh && Wi == lw), "tpspp_conv2d_bf16_fwd: sources disagree on the logical size");
        Hi = lh; Wi = lw;
        cin += d[0];
    }
    for (int i = nsrc; i < 3; ++i) P.src[i] = P.src[nsrc - 1];
    P.Cin = cin; P.Cout = Cout; P.N = N; P.Hi = Hi; P.Wi = Wi; P.Ho = Ho; P.Wo = Wo;
    P.ph = (KH - 1) / 2; P.pw = (KW - 1) / 2;
    TPSPP_REQUIRE(Ho == (Hi + 2 * P.ph - KH) / sh + 1 && Wo == (Wi + 2 * P.pw - KW) / sw + 1,
                  "tpspp_conv2d_bf16_fwd: output size does not match input size / stride ('same' padding)");
    P.wt = reinterpret_cast<const u32x4*>(weight_arranged);
    TPSPP_REQUIRE(residual_f32 >= 0 && residual_f32 <= 3 && out_f32 >= 0 && out_f32 <= 3 &&
                      ((residual_f32 != 2 && out_f32 != 2) || (Cout % 8 == 0 && !split3)) &&
                      ((residual_f32 != 3 && out_f32 != 3) || (Cout % 8 == 0 && split3)),
                  "tpspp_conv2d_bf16_fwd: residual / output layout code must be 0 / 1 / 2 / 3 (blocked: Cout a multiple of 8; 2 bf16 "
                  "not with split3, 3 fp32 only with split3)");
    P.bias = bias; P.res = residual; P.res_f32 = residual_f32; P.out = out; P.out_f32 = out_f32;
    P.post_scale = post_scale; P.post_shift = post_shift;
    P.relu = relu; P.res_mode = res_mode;
    P.nchunks = (cin + KC - 1) / KC;
    if (N == 0) return TPSPP_OK;
    TPSPP_REQUIRE((long)N * ((Cout + BN - 1) / BN) <= 65535, "tpspp_conv2d_bf16_fwd: grid too large");
    hipStream_t st = tpspp::as_stream(stream);
    bool ok = false;
    // the wide blocked 3x3 layers of the backbone (>= 128 output channels on 8x32 / 4x16 maps): 128 x 64 wavefront tiles, weights
    // streamed into registers (tpspp_conv3_wide.hip); bit-identical results
    if (!split3 && KH == 3 && sh == 1 && sw == 1 && !tpspp::g_conv_bf16_no_wide && tpspp::conv3_wide_launch(P, st))
        return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
    // the backbone's stem (3 fp32 channels -> 32 bf16 NCHW): tpspp_conv_stem.hip; bit-identical results
    if (!split3 && KH == 3 && sh == 1 && sw == 1 && !tpspp::g_conv_bf16_no_wide && tpspp::conv_stem_launch(P, st))
        return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
    // the big blocked 3x3 layers: persistent, LDS-DMA-fed kernel (tpspp_conv_bf16_persist.hip); bit-identical results
    if (!split3 && KH == 3 && !tpspp::g_conv_bf16_no_persist && tpspp::conv_bf16_persist_launch(P, sh, sw, st))
        return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
    // 1x1 layers with >= 256 input channels between blocked maps: the wide-tile kernel's 1x1 form (tpspp_conv3_wide.hip)
    if (!split3 && KH == 1 && sh == 1 && sw == 1 && !tpspp::g_conv_bf16_no_wide && tpspp::conv1x1_wide_launch(P, st))
        return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
    // 1x1 layers between blocked maps (the backbone's BasicBlocks): every activation read once, the weight in LDS
    // (tpspp_conv1x1_blk.hip); bit-identical results
    if (!split3 && KH == 1 && sh == 1 && sw == 1 && !tpspp::g_conv_bf16_no_persist && tpspp::conv1x1_blk_launch(P, st))
        return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
    if (split3) {
        ok = tpspp::conv_bf16x3_launch(P, KH, sh, sw, st);
    } else {
        if (KH == 1 && sh == 1 && sw == 1)      ok = launch_by_shape<1, 1, 1, kKC1, false>(P, st);
        else if (KH == 1 && sh == 2 && sw == 2) ok = launch_by_shape<1, 2, 2, kKC1, false>(P, st);
        else if (KH == 3 && sh == 1 && sw == 1) ok = launch_by_shape<3, 1, 1, kKC3, false>(P, st);
        else if (KH == 3 && sh == 2 && sw == 2) ok = launch_by_shape<3, 2, 2, kKC3, false>(P, st);
        else if (KH == 3 && sh == 2 && sw == 1) ok = launch_by_shape<3, 2, 1, kKC3, false>(P, st);
    }
    TPSPP_REQUIRE(ok, "tpspp_conv2d_bf16_fwd: no kernel for a %dx%d kernel with stride (%d,%d)", KH, KW, sh, sw);
    return tpspp::check_launch("tpspp_conv2d_bf16_fwd");
}
