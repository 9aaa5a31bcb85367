// Pooling for the localisation network (classic TPS-STN): MaxPool2d(2,2) and AdaptiveAvgPool2d(1).
// Pure bandwidth; one thread per output element / one wavefront per plane.
// Replaces: nn.MaxPool2d(2, 2) and nn.AdaptiveAvgPool2d(1) in
// mmocr/models/textrecog/preprocessor/tps_preprocessor.py:110-126 (reference).
#include "tpspp_common.h"

namespace {

__global__ void __launch_bounds__(256)
maxpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W, int Ho, int Wo)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)planes * Ho * Wo;
    if (i >= total) return;
    const int ox = (int)(i % Wo);
    const long long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* p = in + (pl * H + 2 * oy) * W + 2 * ox;     // floor mode: 2*oy+1 < H, 2*ox+1 < W
    const float a = p[0], b = p[1], c = p[W], d = p[W + 1];
    // torch's max propagates NaN; fmaxf would drop it
    float m = a;
    m = (b > m || b != b) ? b : m;
    m = (c > m || c != c) ? c : m;
    m = (d > m || d != d) ? d : m;
    out[i] = m;
}

// one wavefront per (n, c) plane; sum in a fixed order (lane-strided partial sums, then a butterfly)
__global__ void __launch_bounds__(256)
global_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int HW)
{
    const int lane = threadIdx.x & 63;
    const int pl = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (pl >= planes) return;
    const float* p = in + (size_t)pl * HW;
    float s = 0.0f;
    for (int i = lane; i < HW; i += 64) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[pl] = s / (float)HW;
}

}  // namespace

TPSPP_EXPORT int tpspp_maxpool2x2_fwd(const float* in, int N, int C, int H, int W, float* out,
                                      tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && out, "tpspp_maxpool2x2_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C > 0 && H >= 2 && W >= 2, "tpspp_maxpool2x2_fwd: bad sizes");
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * C * Ho * Wo;
    if (total == 0) return TPSPP_OK;
    TPSPP_REQUIRE((total + 255) / 256 < (1LL << 31), "tpspp_maxpool2x2_fwd: too large");
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       tpspp::as_stream(stream), in, out, N * C, H, W, Ho, Wo);
    return tpspp::check_launch("tpspp_maxpool2x2_fwd");
}

TPSPP_EXPORT int tpspp_global_avgpool_fwd(const float* in, int N, int C, int H, int W, float* out,
                                          tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && out, "tpspp_global_avgpool_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0, "tpspp_global_avgpool_fwd: bad sizes");
    const int planes = N * C;
    if (planes == 0) return TPSPP_OK;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0,
                       tpspp::as_stream(stream), in, out, planes, H * W);
    return tpspp::check_launch("tpspp_global_avgpool_fwd");
}
