// Device-side helpers shared by the warp kernels (tpspp_warp.hip, tpspp_warp_stream.hip).
// Arithmetic contract: see include/tpspp.h; everything here must stay bit-identical to
// oracle/tps_oracle.c (weight_form 2).  Compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace tpspp_dev {

constexpr int kWave = 64;
constexpr int kMaxK = 64;  // F + 3 <= 64: one lane per row of T in the wave-level solve

struct Taps {
    int o00, o01, o10, o11;   // offsets inside one H x W plane (clamped: always readable)
    float nw, ne, sw, se;
    float wx, wy;             // fractional parts: nw = (1-wy)(1-wx), ne = (1-wy) wx, sw = wy (1-wx), se = wy wx
    bool inx, iny;            // is the east column / south row inside the plane
    int x0, y0;
};

// ATen bilinear, padding_mode='border', align_corners=True; weight form and rounding of the CPU
// vector kernel (oracle/tps_oracle.c, weight_form 2).
__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W)
{
    Taps t;
    float ix = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
    float iy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    const float limx = (float)(W - 1), limy = (float)(H - 1);
    ix = (ix > 0.0f) ? ix : 0.0f;   // NaN -> 0
    iy = (iy > 0.0f) ? iy : 0.0f;
    ix = (ix < limx) ? ix : limx;
    iy = (iy < limy) ? iy : limy;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float w = ix - fx, e = 1.0f - w, nn = iy - fy, s = 1.0f - nn;
    t.nw = s * e; t.ne = s * w; t.sw = nn * e; t.se = nn * w;
    t.wx = w; t.wy = nn;
    t.inx = (x0 + 1) < W;
    t.iny = (y0 + 1) < H;
    const int x1 = t.inx ? x0 + 1 : x0;
    const int y1 = t.iny ? y0 + 1 : y0;
    // 24-bit multiplies are full rate (rows and widths are far below 2^24)
    const int r0 = __mul24(y0, W), r1 = __mul24(y1, W);
    t.o00 = r0 + x0; t.o01 = r0 + x1;
    t.o10 = r1 + x0; t.o11 = r1 + x1;
    t.x0 = x0; t.y0 = y0;
    return t;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ pl, const Taps& t)
{
    float v00 = pl[t.o00];
    float v01 = pl[t.o01];
    float v10 = pl[t.o10];
    float v11 = pl[t.o11];
    v01 = t.inx ? v01 : 0.0f;
    v10 = t.iny ? v10 : 0.0f;
    v11 = (t.inx && t.iny) ? v11 : 0.0f;
    float acc = v00 * t.nw;
    acc = fmaf(v01, t.ne, acc);
    acc = fmaf(v10, t.sw, acc);
    acc = fmaf(v11, t.se, acc);
    return acc;
}

__device__ __forceinline__ void sample_planes(const float* __restrict__ in, float* __restrict__ out,
                                              int C, int HW, int n, const Taps& t)
{
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        const float r0 = bilerp(in + (size_t)(c + 0) * HW, t);
        const float r1 = bilerp(in + (size_t)(c + 1) * HW, t);
        const float r2 = bilerp(in + (size_t)(c + 2) * HW, t);
        const float r3 = bilerp(in + (size_t)(c + 3) * HW, t);
        out[(size_t)(c + 0) * n] = r0;
        out[(size_t)(c + 1) * n] = r1;
        out[(size_t)(c + 2) * n] = r2;
        out[(size_t)(c + 3) * n] = r3;
    }
    if (c + 3 == C) {   // the 3-channel image case: keep all 12 taps in flight
        const float r0 = bilerp(in + (size_t)(c + 0) * HW, t);
        const float r1 = bilerp(in + (size_t)(c + 1) * HW, t);
        const float r2 = bilerp(in + (size_t)(c + 2) * HW, t);
        out[(size_t)(c + 0) * n] = r0;
        out[(size_t)(c + 1) * n] = r1;
        out[(size_t)(c + 2) * n] = r2;
        return;
    }
    for (; c < C; ++c) out[(size_t)c * n] = bilerp(in + (size_t)c * HW, t);
}

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One wavefront: T[i] = sum_q inv[i][q] * Cz[q], q ascending, FMA chain from 0.  Lane i owns row i.
// `inv` may be LDS or global.  Returns (Tx, Ty) of row `lane` (garbage for lane >= K).
__device__ __forceinline__ float2 wave_solve_T(const float* inv, const float* __restrict__ ctrl_b,
                                               int F, int K, int lane)
{
    float cx = 0.0f, cy = 0.0f;             // rows F..F+2 of [C';0] are the appended zeros
    if (lane < F) {
        const float2 c = reinterpret_cast<const float2*>(ctrl_b)[lane];
        cx = c.x; cy = c.y;
    }
    const int row = lane < K ? lane : K - 1;
    const float* h = inv + row * K;
    float ax = 0.0f, ay = 0.0f;
    for (int q = 0; q < K; ++q) {
        const float hv = h[q];
        const float bx = readlane_f(cx, q);
        const float by = readlane_f(cy, q);
        ax = fmaf(hv, bx, ax);
        ay = fmaf(hv, by, ay);
    }
    return make_float2(ax, ay);
}

// compile-time loop: every index is a constant, so register arrays stay in registers
template <int N, int I = 0, class Fn>
__device__ __forceinline__ void static_for(Fn&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// Workgroup barrier that orders LDS traffic only: outstanding global loads and LDS-DMA keep flying
// (a __syncthreads() would also drain vmcnt).
__device__ __forceinline__ void lds_only_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// optional diagnostics: 8 stamps per workgroup; slot 7 = chip-wide 100 MHz clock at slot 0
__device__ __forceinline__ void stamp(long long* trace, int slot)
{
    if (trace && (threadIdx.x & (kWave - 1)) == 0) {
        trace[(size_t)blockIdx.x * 8 + slot] = (long long)__builtin_amdgcn_s_memtime();
        if (slot == 0) trace[(size_t)blockIdx.x * 8 + 7] = (long long)wall_clock64();
    }
}

}  // namespace tpspp_dev
