// GPU-side ResizeOCR + ToTensorOCR + NormalizeOCR: a batch of uint8 HWC crops of
// different sizes -> one (N, C, H, W_max) fp32 tensor, resized (bilinear), right-padded and normalised, so that the
// recogniser's input no longer passes through a CPU data loader.
//
// Reference: mmocr/datasets/pipelines/ocr_transforms.py:67-156 (ResizeOCR.__call__, ToTensorOCR, NormalizeOCR) as
// configured by configs/_base_/recog_pipelines/crnn_pp_pipeline.py:85-95.  The resize there is mmcv.imresize ->
// cv::resize(INTER_LINEAR) on uint8; this kernel follows OpenCV's published 8-bit algorithm (resize.cpp: source
// coordinate from a double scale cast to float, 11-bit fixed-point weights rounded to nearest-even, horizontal pass in
// int32, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; INTER_AREA for an exact 2x2 shrink).
// PARITY UNPINNED: OpenCV is not installed where this was built, so the integer arithmetic is checked bit-for-bit
// against oracle/resize_oracle.py (the same restatement) and not against OpenCV itself.  "/255, -mean, /std" is a
// 256-entry table per channel computed by the caller with torch's own fp32 arithmetic, hence exact.
//
// Round 6 -- interpolation 1 = backend='pillow' (ocr_transforms.py:46,65,99-101 forward `backend` to mmcv.imresize, whose
// pillow branch is Image.fromarray(img).resize(size, Image.BILINEAR)): resize_norm_pillow_kernel restates Pillow's
// src/libImaging/Resample.c -- precompute_coeffs in double (support = max(scale, 1), window [int(center - support + 0.5),
// int(center + support + 0.5)) clipped to the image, triangle weights normalised by their ascending sum),
// normalize_coeffs_8bpc (int(0.5 + k 2^22)), horizontal pass into uint8, then the vertical pass, each
// clip8((2^21 + sum(pixel * k)) >> 22).  IEEE double +, -, *, / with -ffp-contract=off are the same operations on the device as
// in Pillow's C on the host, so the coefficients are computed where they are used.  PINNED: tests/golden/resize_pillow.npz
// holds the installed Pillow's own outputs (tests/golden/make_resize_golden.py) and the -m gpu test compares bit for bit.
//
// Bound: HBM, trivially (one thread per output pixel: 4 taps x C bytes in, C floats out; 49 KB per 32x128 image).
#include "tpspp_common.h"

namespace {

struct ResizeParams {
    const unsigned char* src;      // packed HWC images
    const long long* off;          // (N) byte offset of each image
    const int* sh; const int* sw;  // (N) source height / width
    const int* dw;                 // (N) resized width (<= W); columns >= dw[n] are padding
    const float* lut;              // (C, 256): value -> normalised float
    float* out;                    // (N, C, H, W)
    int N, C, H, W, pad_value;
};

__device__ __forceinline__ void coeff(int d, int src, int dst, int& s, float& f)
{
    const double scale = 1.0 / ((double)dst / (double)src);
    f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f = f - (float)s;
}

__device__ __forceinline__ int sat_short(float v)
{
    const float r = rintf(v);                                  // cvRound: nearest, ties to even
    return (int)fminf(fmaxf(r, -32768.0f), 32767.0f);
}

__global__ void __launch_bounds__(256)
resize_norm_kernel(const ResizeParams P)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.H * P.W) return;
    const int y = p / P.W, x = p - y * P.W;
    const int SH = P.sh[n], SW = P.sw[n], DW = P.dw[n], C = P.C;
    const unsigned char* img = P.src + P.off[n];
    float* o = P.out + ((size_t)n * C * P.H + y) * P.W + x;
    const size_t plane = (size_t)P.H * P.W;
    if (x >= DW) {                                             // mmcv.impad: constant padding on the right
        for (int c = 0; c < C; ++c) o[c * plane] = P.lut[c * 256 + P.pad_value];
        return;
    }
    if (SH == 2 * P.H && SW == 2 * DW) {                       // INTER_AREA, exact 2x2
        const unsigned char* r0 = img + ((size_t)(2 * y) * SW + 2 * x) * C;
        const unsigned char* r1 = r0 + (size_t)SW * C;
        for (int c = 0; c < C; ++c) {
            const int v = ((int)r0[c] + (int)r0[C + c] + (int)r1[c] + (int)r1[C + c] + 2) >> 2;
            o[c * plane] = P.lut[c * 256 + v];
        }
        return;
    }
    int sx, sy;
    float fx, fy;
    coeff(x, SW, DW, sx, fx);
    coeff(y, SH, P.H, sy, fy);
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= SW - 1) { fx = 0.0f; sx = SW - 1; }
    const int a0 = sat_short((1.0f - fx) * 2048.0f), a1 = sat_short(fx * 2048.0f);
    const int b0 = sat_short((1.0f - fy) * 2048.0f), b1 = sat_short(fy * 2048.0f);
    const int sx1 = min(sx + 1, SW - 1);
    const int y0 = min(max(sy, 0), SH - 1), y1 = min(max(sy + 1, 0), SH - 1);
    const unsigned char* r0 = img + (size_t)y0 * SW * C;
    const unsigned char* r1 = img + (size_t)y1 * SW * C;
    for (int c = 0; c < C; ++c) {
        const int S0 = (int)r0[sx * C + c] * a0 + (int)r0[sx1 * C + c] * a1;
        const int S1 = (int)r1[sx * C + c] * a0 + (int)r1[sx1 * C + c] * a1;
        int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        o[c * plane] = P.lut[c * 256 + v];
    }
}

// ---- Pillow's BILINEAR (Resample.c) --------------------------------------------------------------------------------------
struct PilAxis { int lo, n; double center, ss, ww; };

__device__ __forceinline__ double pil_triangle(double x)
{
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// precompute_coeffs for ONE output index d of an axis resampled in_size -> out_size (box = the whole image)
__device__ __forceinline__ PilAxis pil_axis(int d, int in_size, int out_size)
{
    PilAxis A;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                 // bilinear: support 1.0
    A.ss = 1.0 / filterscale;
    A.center = ((double)d + 0.5) * scale;
    int lo = (int)(A.center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(A.center + support + 0.5);
    if (hi > in_size) hi = in_size;
    A.lo = lo;
    A.n = hi - lo;
    double ww = 0.0;
    for (int x = 0; x < A.n; ++x) ww += pil_triangle(((double)(x + lo) - A.center + 0.5) * A.ss);
    A.ww = ww;
    return A;
}

// k[x] / ww, then normalize_coeffs_8bpc
__device__ __forceinline__ int pil_coef(const PilAxis& A, int x)
{
    double w = pil_triangle(((double)(x + A.lo) - A.center + 0.5) * A.ss);
    if (A.ww != 0.0) w = w / A.ww;
    return w < 0.0 ? (int)(-0.5 + w * 4194304.0) : (int)(0.5 + w * 4194304.0);
}

__device__ __forceinline__ int pil_clip8(int v)
{
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <int C>
__global__ void __launch_bounds__(256)
resize_norm_pillow_kernel(const ResizeParams P)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.H * P.W) return;
    const int y = p / P.W, x = p - y * P.W;
    const int SH = P.sh[n], SW = P.sw[n], DW = P.dw[n];
    const unsigned char* img = P.src + P.off[n];
    float* o = P.out + ((size_t)n * C * P.H + y) * P.W + x;
    const size_t plane = (size_t)P.H * P.W;
    if (x >= DW) {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c * plane] = P.lut[c * 256 + P.pad_value];
        return;
    }
    // (an axis that is not resized has the coefficients {2^22} / {2^22, 0}: the pass Pillow skips is an exact identity here)
    const PilAxis AX = pil_axis(x, SW, DW), AY = pil_axis(y, SH, P.H);
    constexpr int KC = 8;                                      // horizontal coefficients kept in registers (scale <= 3.5)
    int kx[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) kx[j] = j < AX.n ? pil_coef(AX, j) : 0;
    int vacc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) vacc[c] = 1 << 21;
    for (int i = 0; i < AY.n; ++i) {
        const int ky = pil_coef(AY, i);
        const unsigned char* row = img + ((size_t)(AY.lo + i) * SW + AX.lo) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << 21;
        if (AX.n <= KC) {
#pragma unroll
            for (int j = 0; j < KC; ++j)
                if (j < AX.n) {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += (int)row[j * C + c] * kx[j];
                }
        } else {
            for (int j = 0; j < AX.n; ++j) {
                const int k = pil_coef(AX, j);
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += (int)row[j * C + c] * k;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) vacc[c] += pil_clip8(acc[c]) * ky;          // the horizontal pass's uint8 image, times ky
    }
#pragma unroll
    for (int c = 0; c < C; ++c) o[c * plane] = P.lut[c * 256 + pil_clip8(vacc[c])];
}

}  // namespace

TPSPP_EXPORT int tpspp_resize_normalize_fwd(const unsigned char* src_packed, const long long* src_offsets,
                                            const int* src_h, const int* src_w, const int* resize_w,
                                            const float* lut, int pad_value, int N, int C, int H, int W,
                                            float* out, int interpolation, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(interpolation == TPSPP_RESIZE_CV2 || interpolation == TPSPP_RESIZE_PILLOW,
                  "tpspp_resize_normalize_fwd: interpolation must be TPSPP_RESIZE_CV2 (0) or TPSPP_RESIZE_PILLOW (1)");
    TPSPP_REQUIRE(src_packed && src_offsets && src_h && src_w && resize_w && lut && out,
                  "tpspp_resize_normalize_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && pad_value >= 0 && pad_value <= 255,
                  "tpspp_resize_normalize_fwd: bad sizes (1..4 channels, pad value 0..255)");
    TPSPP_REQUIRE(N <= 65535, "tpspp_resize_normalize_fwd: at most 65535 images per call");
    if (N == 0) return TPSPP_OK;
    ResizeParams P;
    P.src = src_packed; P.off = src_offsets; P.sh = src_h; P.sw = src_w; P.dw = resize_w; P.lut = lut; P.out = out;
    P.N = N; P.C = C; P.H = H; P.W = W; P.pad_value = pad_value;
    const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)N);
    hipStream_t st = tpspp::as_stream(stream);
    if (interpolation == TPSPP_RESIZE_PILLOW) {
        switch (C) {
        case 1: hipLaunchKernelGGL(resize_norm_pillow_kernel<1>, grid, dim3(256), 0, st, P); break;
        case 2: hipLaunchKernelGGL(resize_norm_pillow_kernel<2>, grid, dim3(256), 0, st, P); break;
        case 3: hipLaunchKernelGGL(resize_norm_pillow_kernel<3>, grid, dim3(256), 0, st, P); break;
        default: hipLaunchKernelGGL(resize_norm_pillow_kernel<4>, grid, dim3(256), 0, st, P); break;
        }
    } else {
        hipLaunchKernelGGL(resize_norm_kernel, grid, dim3(256), 0, st, P);
    }
    return tpspp::check_launch("tpspp_resize_normalize_fwd");
}
