// Kernel lab for the classic warp (BASELINE configs[1]: 512 x 3x32x100 fp32, F = 20).
// Times every variant of the q4 kernel over rotating buffers with HIP events, checks each one bit for
// bit against the library's production path (tpspp_warp_fwd), prints per-workgroup phase stamps.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I tps_pp_amd/csrc -I include \
//         -I scripts/ubench scripts/ubench/warp_lab.hip -o scripts/ubench/warp_lab -ldl
//   scripts/ubench/warp_lab scripts/ubench/warp_lab_consts.bin tps_pp_amd/libtpspp_hip.so
// (consts file: scripts/ubench/make_lab_consts.py)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "warp_lab_kernels.h"

using namespace tpspp_q4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int (*warp_fwd_t)(const float*, int, int, int, const float*, int, int, int, const float*, const float*,
                          const float*, const float*, int, const float*, const float*, int, int, int, int, int,
                          float*, float*, float*, int32_t*, void*);

static const int F = 20, K = 23, C = 3, H = 32, W = 100, n = H * W;
static int N = 512;
static const int SETS = 14;

struct Bufs {
    float* in[SETS]; float* ctrl[SETS]; float* out[SETS];
    float* inv; float* p_hat; float* p_hat_t; float* p_hat_pk; float* p_hat_pk2;
    float* ref[2]; float* refgrid; int32_t* refidx; float* grid; int32_t* idx;
    long long* trace; long long* trace2;
};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

// ---- micro kernels ---------------------------------------------------------------------------
__global__ void spread_k(long long* t)
{
    extern __shared__ float sm[];
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = (long long)wall_clock64(); }
    if (threadIdx.x == 4096) sm[0] = 1.f;
}
template <int MODE>
__global__ void __launch_bounds__(256) copy_k(const v4f* __restrict__ src, char* dst, int n4)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        v4f x = src[i];
        store16<MODE>((gchar*)dst, (unsigned)i * 16u, x);
    }
}

template <int MODE, int LDMODE>
__global__ void __launch_bounds__(1024) copy_blk_k(const v4f* __restrict__ src, char* dst, int per_block4, long long* tr)
{
    if (threadIdx.x == 0 && tr) tr[blockIdx.x * 2] = (long long)wall_clock64();
    const v4f* s = src + (size_t)blockIdx.x * per_block4;
    const unsigned base = (unsigned)blockIdx.x * (unsigned)per_block4 * 16u;
    for (int i = threadIdx.x; i < per_block4; i += blockDim.x) {
        v4f x;
        if (LDMODE) x = __builtin_nontemporal_load(s + i); else x = s[i];
        store16<MODE>((gchar*)dst, base + (unsigned)i * 16u, x);
    }
    if (threadIdx.x == 0 && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[blockIdx.x * 2 + 1] = (long long)wall_clock64(); }
}

// DMA latency probe: `waves` wavefronts per block, each streams `per_wave` 1-KB pieces of its block's 76.8-KB chunk into LDS
template <int NT>
__global__ void __launch_bounds__(1024) dma_probe_k(const float* in, int per_wave, int chunk_bytes, long long* tr, int regs)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    const char* src = reinterpret_cast<const char*>(in) + (size_t)blockIdx.x * chunk_bytes;
    v4f acc = {0, 0, 0, 0};
    for (int i = 0; i < per_wave; ++i) {
        const int piece = wv + i * nw;
        if (regs) {
            acc += NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + piece * 1024 + lane * 16)) : *reinterpret_cast<const v4f*>(src + piece * 1024 + lane * 16);
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sm) + piece * 1024), 16, 0, NT ? 2 : 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (regs) asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc)); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
    if (lane == 0) { tr[(blockIdx.x * 16 + wv) * 2] = (long long)(t1 - t0); tr[(blockIdx.x * 16 + wv) * 2 + 1] = (long long)(t2 - t0); }
    if (regs && acc[0] == 1234.5f) tr[0] = 0;
}

template <class L> static float period_us(L launch, int iters, int warm = 30)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / iters;
}

template <int IMGS, int NW, int NLOAD, int FIRST, int STORE, int LDNT, bool AUX>
static void launch_q4(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    Q4Params P;
    P.in = B.in[set]; P.C = C; P.H = H; P.W = W;
    P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx;
    const int G = W / 4, U = (G + 1) / 2;
    P.units = (H / 2) * U;
    P.trace = trace;
    const size_t lds = q4_lds_bytes(F, C, H, W, IMGS, &P.zero_off);
    auto kern = tps_warp_q4_kernel<F, C, H, W, IMGS, NW, NLOAD, FIRST, STORE, LDNT, AUX>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
    hipLaunchKernelGGL(kern, dim3((N + IMGS - 1) / IMGS), dim3((NW + NLOAD) * 64), lds, st, P);
}

template <int NW, int NLOAD, int FIRST, int STORE, int LDNT, int DBG>
static void launch_m(const Bufs& B, int set, float* out, float*, int32_t*, long long* trace, hipStream_t st)
{
    Q4Params P;
    P.in = B.in[set]; P.C = C; P.H = H; P.W = W;
    P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = nullptr; P.idx = nullptr; P.units = 0;
    P.trace = trace;
    const size_t lds = m_lds_bytes(F, C, H, W, &P.zero_off);
    auto kern = tps_warp_m_kernel<F, C, H, W, NW, NLOAD, FIRST, STORE, LDNT, DBG>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
    hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3((NW + NLOAD) * 64), lds, st, P);
}

template <int NW, int NLOAD, int FIRST, int STORE, int LDNT, int DBG>
static void launch_m2(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    Q4Params P;
    P.in = B.in[set]; P.C = C; P.H = H; P.W = W;
    P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.units = 0;
    P.trace = trace;
    const size_t lds = m_lds_bytes(F, C, H, W, &P.zero_off);
    if (grid || idx) {
        auto kern = tps_warp_m2_kernel<F, C, H, W, NW, NLOAD, FIRST, STORE, LDNT, DBG, true>;
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3((NW + NLOAD) * 64), lds, st, P);
    } else {
        auto kern = tps_warp_m2_kernel<F, C, H, W, NW, NLOAD, FIRST, STORE, LDNT, DBG, false>;
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3((NW + NLOAD) * 64), lds, st, P);
    }
}

template <int NLOAD, int FIRST, int STORE, int LDNT>
static void launch_m3(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m3_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m3_kernel<F, C, H, W, NLOAD, FIRST, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m3_kernel<F, C, H, W, NLOAD, FIRST, STORE, LDNT, true, false>);
    else go(tps_warp_m3_kernel<F, C, H, W, NLOAD, FIRST, STORE, LDNT, false, false>);
}

template <int NLOAD, int STORE, int LDNT>
static void launch_m4(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m4_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD + 1) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m4_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m4_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false>);
    else go(tps_warp_m4_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false>);
}

template <int NLOAD, int STORE, int LDNT>
static void launch_m5(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m5_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m5_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false>);
    else go(tps_warp_m5_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false>);
}

template <int NLOAD, int STORE, int LDNT>
static void launch_m7(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m7_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m7_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false>);
    else go(tps_warp_m7_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false>);
}

template <int NLOAD, int STORE, int LDNT>
static void launch_m8(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m8_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m8_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false>);
    else go(tps_warp_m8_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT>
static void launch_m9(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m9_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT>);
    else if (grid || idx) go(tps_warp_m9_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT>);
    else go(tps_warp_m9_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT>
static void launch_m10(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m10_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT>);
    else if (grid || idx) go(tps_warp_m10_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT>);
    else go(tps_warp_m10_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB>
static void launch_m13(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m13_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB>);
    else if (grid || idx) go(tps_warp_m13_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB>);
    else go(tps_warp_m13_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB>
static void launch_m14(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m14_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB>);
    else if (grid || idx) go(tps_warp_m14_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB>);
    else go(tps_warp_m14_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB>
static void launch_m15(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m15_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB>);
    else if (grid || idx) go(tps_warp_m15_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB>);
    else go(tps_warp_m15_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB>
static void launch_m17(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off) + 2048;
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m17_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB>);
    else if (grid || idx) go(tps_warp_m17_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB>);
    else go(tps_warp_m17_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB>
static void launch_m18(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m18_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB>);
    else if (grid || idx) go(tps_warp_m18_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB>);
    else go(tps_warp_m18_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT, int KB, int BMODE>
static void launch_m16(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m16_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT, KB, BMODE>);
    else if (grid || idx) go(tps_warp_m16_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT, KB, BMODE>);
    else go(tps_warp_m16_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT, KB, BMODE>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT>
static void launch_m12(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m12_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT>);
    else if (grid || idx) go(tps_warp_m12_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT>);
    else go(tps_warp_m12_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT>);
}

template <int NLOAD, int STORE, int LDNT, int AWAIT>
static void launch_m11(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_pk2;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m11_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true, AWAIT>);
    else if (grid || idx) go(tps_warp_m11_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false, AWAIT>);
    else go(tps_warp_m11_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false, AWAIT>);
}

template <int NLOAD, int STORE, int LDNT>
static void launch_m6(const Bufs& B, int set, float* out, float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    M3Params P;
    P.in = B.in[set]; P.ctrl = B.ctrl[set]; P.inv_delta_c = B.inv; P.p_hat_t = B.p_hat_t;
    P.N = N; P.n = n; P.Ho = H; P.Wo = W;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace; P.trace2 = B.trace2;
    const size_t lds = m5_lds_bytes(F, C, H, W, H, W, &P.zero_off, &P.out_off);
    const int threads = (13 + NLOAD) * 64;
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
        hipLaunchKernelGGL(kern, dim3((N + 1) / 2), dim3(threads), lds, st, P);
    };
    if (trace) go(tps_warp_m6_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, true>);
    else if (grid || idx) go(tps_warp_m6_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, true, false>);
    else go(tps_warp_m6_kernel<F, C, H, W, H, W, NLOAD, STORE, LDNT, false, false>);
}

struct Variant { std::string name; std::function<void(const Bufs&, int, float*, float*, int32_t*, long long*, hipStream_t)> run; };

#define VAR(IMGS, NW, NLOAD, FIRST, STORE, LDNT) \
    Variant{ "q4 imgs=" #IMGS " nw=" #NW " nload=" #NLOAD " first=" #FIRST " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          if (g || ix) launch_q4<IMGS, NW, NLOAD, FIRST, STORE, LDNT, true>(B, s, o, g, ix, tr, st); \
          else launch_q4<IMGS, NW, NLOAD, FIRST, STORE, LDNT, false>(B, s, o, g, ix, tr, st); } }

#define MVAR(NW, NLOAD, FIRST, STORE, LDNT, DBG) \
    Variant{ "m nw=" #NW " nload=" #NLOAD " first=" #FIRST " store=" #STORE " ldnt=" #LDNT " dbg=" #DBG, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m<NW, NLOAD, FIRST, STORE, LDNT, DBG>(B, s, o, g, ix, tr, st); } }

#define M2VAR(NW, NLOAD, FIRST, STORE, LDNT, DBG) \
    Variant{ "m2 nw=" #NW " nload=" #NLOAD " first=" #FIRST " store=" #STORE " ldnt=" #LDNT " dbg=" #DBG, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m2<NW, NLOAD, FIRST, STORE, LDNT, DBG>(B, s, o, g, ix, tr, st); } }

#define M3VAR(NLOAD, FIRST, STORE, LDNT) \
    Variant{ "m3 nload=" #NLOAD " first=" #FIRST " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m3<NLOAD, FIRST, STORE, LDNT>(B, s, o, g, ix, tr, st); } }

#define M4VAR(NLOAD, STORE, LDNT) \
    Variant{ "m4 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m4<NLOAD, STORE, LDNT>(B, s, o, g, ix, tr, st); } }

#define M6VAR(NLOAD, STORE, LDNT) \
    Variant{ "m6 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m6<NLOAD, STORE, LDNT>(B, s, o, g, ix, tr, st); } }
#define M7VAR(NLOAD, STORE, LDNT) \
    Variant{ "m7 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m7<NLOAD, STORE, LDNT>(B, s, o, g, ix, tr, st); } }
#define M8VAR(NLOAD, STORE, LDNT) \
    Variant{ "m8 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m8<NLOAD, STORE, LDNT>(B, s, o, g, ix, tr, st); } }
#define M9VAR(NLOAD, STORE, LDNT, AWAIT) \
    Variant{ "m9 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m9<NLOAD, STORE, LDNT, AWAIT>(B, s, o, g, ix, tr, st); } }
#define M10VAR(NLOAD, STORE, LDNT, AWAIT) \
    Variant{ "m10 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m10<NLOAD, STORE, LDNT, AWAIT>(B, s, o, g, ix, tr, st); } }
#define M11VAR(NLOAD, STORE, LDNT, AWAIT) \
    Variant{ "m11 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m11<NLOAD, STORE, LDNT, AWAIT>(B, s, o, g, ix, tr, st); } }
#define M12VAR(NLOAD, STORE, LDNT, AWAIT) \
    Variant{ "m12 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m12<NLOAD, STORE, LDNT, AWAIT>(B, s, o, g, ix, tr, st); } }
#define M13VAR(NLOAD, STORE, LDNT, AWAIT, KB) \
    Variant{ "m13 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m13<NLOAD, STORE, LDNT, AWAIT, KB>(B, s, o, g, ix, tr, st); } }
#define M14VAR(NLOAD, STORE, LDNT, AWAIT, KB) \
    Variant{ "m14 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m14<NLOAD, STORE, LDNT, AWAIT, KB>(B, s, o, g, ix, tr, st); } }
#define M15VAR(NLOAD, STORE, LDNT, AWAIT, KB) \
    Variant{ "m15 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m15<NLOAD, STORE, LDNT, AWAIT, KB>(B, s, o, g, ix, tr, st); } }
#define M16VAR(NLOAD, STORE, LDNT, AWAIT, KB, BMODE) \
    Variant{ "m16 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB " bmode=" #BMODE, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m16<NLOAD, STORE, LDNT, AWAIT, KB, BMODE>(B, s, o, g, ix, tr, st); } }
#define M17VAR(NLOAD, STORE, LDNT, AWAIT, KB) \
    Variant{ "m17 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m17<NLOAD, STORE, LDNT, AWAIT, KB>(B, s, o, g, ix, tr, st); } }
#define M18VAR(NLOAD, STORE, LDNT, AWAIT, KB) \
    Variant{ "m18 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT " await=" #AWAIT " kb=" #KB, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m18<NLOAD, STORE, LDNT, AWAIT, KB>(B, s, o, g, ix, tr, st); } }
#define M5VAR(NLOAD, STORE, LDNT) \
    Variant{ "m5 nload=" #NLOAD " store=" #STORE " ldnt=" #LDNT, \
      [](const Bufs& B, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) { \
          launch_m5<NLOAD, STORE, LDNT>(B, s, o, g, ix, tr, st); } }

static void trace_report_m3(const Bufs& B, int nblocks, int launches)
{
    std::vector<long long> t((size_t)launches * nblocks * 16);
    CK(hipMemcpy(t.data(), B.trace, t.size() * 8, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> ph[11], lA, lB;
    std::vector<long long> s0(launches, 1LL << 62), s1(launches, 0), e0(launches, 1LL << 62), e1(launches, 0);
    for (int l = 0; l < launches; ++l)
        for (int b = 0; b < nblocks; ++b) {
            const long long* s = &t[((size_t)l * nblocks + b) * 16];
            if (l == launches - 1) { for (int i = 1; i <= 10; ++i) ph[i].push_back(s[i] / 2400.0); if (s[14]) { lA.push_back(s[14] / 2400.0); lB.push_back(s[15] / 2400.0); } }
            s0[l] = std::min(s0[l], s[12]); s1[l] = std::max(s1[l], s[12]);
            e0[l] = std::min(e0[l], s[13]); e1[l] = std::max(e1[l], s[13]);
        }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    {
        std::vector<long long> t2(nblocks * 2); CK(hipMemcpy(t2.data(), B.trace2, t2.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ia, ib, l0;
        for (int b = 0; b < nblocks; ++b) if (t2[b * 2]) { ia.push_back(t2[b * 2] / 2400.0); ib.push_back(t2[b * 2 + 1] / 2400.0); l0.push_back(t[((size_t)(launches - 1) * nblocks + b) * 16 + 11] / 2400.0); }
        if (!ia.empty()) printf("      loader 0 (since kernel entry): started %.2f | A issued %.2f | B issued %.2f\n", med(l0), med(ia), med(ib));
        CK(hipMemset(B.trace2, 0, 4096 * 16));
    }
    if (!lA.empty()) printf("      loader 0: its share of A landed %.2f (p90 %.2f), everything landed %.2f (p90 %.2f)\n", med(lA), pct(lA, .9), med(lB), pct(lB, .9));
    printf("      p10/p90: T ready %.2f/%.2f | A landed %.2f/%.2f | A staged %.2f/%.2f | B landed %.2f/%.2f | B staged %.2f/%.2f | retired %.2f/%.2f max %.2f\n",
           pct(ph[3], .1), pct(ph[3], .9), pct(ph[5], .1), pct(ph[5], .9), pct(ph[6], .1), pct(ph[6], .9), pct(ph[7], .1), pct(ph[7], .9),
           pct(ph[8], .1), pct(ph[8], .9), pct(ph[10], .1), pct(ph[10], .9), pct(ph[10], 1.0));
    printf("      stamps since start (us, medians): loads issued %.2f | ctrl+inv in %.2f | T ready %.2f | grid done %.2f | A landed %.2f | A staged %.2f | "
           "B landed %.2f | B staged %.2f | stores issued %.2f | retired %.2f\n",
           med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]), med(ph[6]), med(ph[7]), med(ph[8]), med(ph[9]), med(ph[10]));
    for (int l = launches - 2; l < launches; ++l) {
        printf("      launch %d: starts spread %.2f, first start -> first end %.2f, -> last end %.2f us", l,
               (s1[l] - s0[l]) / 100.0, (e0[l] - s0[l]) / 100.0, (e1[l] - s0[l]) / 100.0);
        if (l) printf("; gap after previous launch's last end %.2f us, start-to-start %.2f us", (s0[l] - e1[l - 1]) / 100.0, (s0[l] - s0[l - 1]) / 100.0);
        printf("\n");
    }
}

// 16-slot trace of the m kernel: phases per workgroup + wall-clock body and gaps over `launches` back-to-back launches
static void trace_report_m(const Bufs& B, int nblocks, int launches)
{
    std::vector<long long> t((size_t)launches * nblocks * 16);
    CK(hipMemcpy(t.data(), B.trace, t.size() * 8, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> ph[8], phA;
    std::vector<long long> s0(launches, 1LL << 62), s1(launches, 0), e0(launches, 1LL << 62), e1(launches, 0);
    for (int l = 0; l < launches; ++l)
        for (int b = 0; b < nblocks; ++b) {
            const long long* s = &t[((size_t)l * nblocks + b) * 16];
            if (l == launches - 1) {
                for (int i = 0; i < 7; ++i) ph[i].push_back((s[i + 1] - s[i]) / 2400.0);
                ph[7].push_back((s[11] - s[0]) / 2400.0);
                if (s[12]) phA.push_back((s[12] - s[0]) / 2400.0);
            }
            s0[l] = std::min(s0[l], s[8]); s1[l] = std::max(s1[l], s[8]);
            e0[l] = std::min(e0[l], s[9]); e1[l] = std::max(e1[l], s[9]);
        }
    printf("      phases (us, medians): issue %.2f | ctrl wait %.2f | T+barrier %.2f | grid %.2f | wait img %.2f | taps %.2f | drain %.2f ; DMA landed at %.2f\n",
           med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]), med(ph[6]), med(ph[7]));
    if (!phA.empty()) printf("      image A's share of loader 0 landed at %.2f (median)\n", med(phA));
    for (int l = 0; l < launches; ++l) {
        printf("      launch %d: starts spread %.2f, first start -> first end %.2f, -> last end %.2f us", l,
               (s1[l] - s0[l]) / 100.0, (e0[l] - s0[l]) / 100.0, (e1[l] - s0[l]) / 100.0);
        if (l) printf("; gap after previous launch's last end %.2f us, start-to-start %.2f us", (s0[l] - e1[l - 1]) / 100.0, (s0[l] - s0[l - 1]) / 100.0);
        printf("\n");
    }
}

static void trace_report(const Bufs& B, int nblocks)
{
    std::vector<long long> t(nblocks * 8);
    CK(hipMemcpy(t.data(), B.trace, t.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> d[5];
    long long w0 = 1LL << 62;
    for (int b = 0; b < nblocks; ++b) w0 = std::min(w0, t[b * 8 + 7]);
    double spread = 0, endmax = 0; std::vector<double> starts;
    for (int b = 0; b < nblocks; ++b) {
        const long long* s = &t[b * 8];
        for (int i = 0; i < 4; ++i) d[i].push_back((s[i + 1] - s[i]) / 2400.0);
        const double life = (s[4] - s[0]) / 2400.0;
        d[4].push_back(life);
        const double st = (s[7] - w0) / 100.0;
        starts.push_back(st);
        spread = std::max(spread, st); endmax = std::max(endmax, st + life);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("      trace: T %.2f | grid %.2f | wait img %.2f | taps+stores %.2f | life med %.2f max %.2f us; "
           "starts med %.2f max %.2f us; first start -> last end %.2f us\n",
           med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]), mx(d[4]), med(starts), spread, endmax);
}

int main(int argc, char** argv)
{
    const char* consts = argc > 1 ? argv[1] : "scripts/ubench/warp_lab_consts.bin";
    const char* libpath = argc > 2 ? argv[2] : "tps_pp_amd/libtpspp_hip.so";
    const int iters = argc > 3 ? atoi(argv[3]) : 1500;
    const char* only = argc > 4 ? argv[4] : "";
    // ---- constants ----
    std::vector<float> hinv(K * K), hphat((size_t)n * K), hident(F * 2);
    {
        FILE* f = fopen(consts, "rb");
        if (!f) { printf("cannot open %s\n", consts); return 1; }
        if (fread(hinv.data(), 4, hinv.size(), f) != hinv.size() || fread(hphat.data(), 4, hphat.size(), f) != hphat.size() ||
            fread(hident.data(), 4, hident.size(), f) != hident.size()) { printf("short consts file\n"); return 1; }
        fclose(f);
    }
    std::vector<float> hphat_t((size_t)K * n);
    for (int p = 0; p < n; ++p) for (int q = 0; q < K; ++q) hphat_t[(size_t)q * n + p] = hphat[(size_t)p * K + q];

    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    warp_fwd_t warp_fwd = (warp_fwd_t)dlsym(lib, "tpspp_warp_fwd");
    auto set_trace = (int (*)(long long*))dlsym(lib, "tpspp_warp_set_trace");
    if (!warp_fwd || !set_trace) { printf("symbols missing\n"); return 1; }

    Bufs B;
    const size_t img_bytes = (size_t)N * C * n * 4, ctrl_bytes = (size_t)N * F * 2 * 4;
    uint32_t seed = 12345;
    std::vector<float> himg((size_t)N * C * n), hctrl((size_t)N * F * 2);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&B.in[s], img_bytes)); CK(hipMalloc(&B.out[s], img_bytes)); CK(hipMalloc(&B.ctrl[s], ctrl_bytes));
        for (auto& x : himg) x = (float)(lcg(seed) >> 8) / 8388608.0f - 1.0f;
        for (size_t i = 0; i < hctrl.size(); ++i)
            hctrl[i] = hident[i % (F * 2)] + 0.05f * ((float)(lcg(seed) >> 8) / 8388608.0f - 1.0f);
        if (s == 1) {   // a nasty set: large perturbations (clamped / out-of-image taps), specials in the image
            for (size_t i = 0; i < hctrl.size(); ++i) hctrl[i] = hident[i % (F * 2)] + 0.8f * ((float)(lcg(seed) >> 8) / 8388608.0f - 1.0f);
            for (size_t i = 0; i < himg.size(); i += 997) himg[i] = -0.0f;
        }
        CK(hipMemcpy(B.in[s], himg.data(), img_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.ctrl[s], hctrl.data(), ctrl_bytes, hipMemcpyHostToDevice));
        CK(hipMemset(B.out[s], 0xff, img_bytes));
    }
    CK(hipMalloc(&B.inv, K * K * 4)); CK(hipMalloc(&B.p_hat, hphat.size() * 4)); CK(hipMalloc(&B.p_hat_t, hphat_t.size() * 4));
    CK(hipMemcpy(B.inv, hinv.data(), K * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.p_hat, hphat.data(), hphat.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.p_hat_t, hphat_t.data(), hphat_t.size() * 4, hipMemcpyHostToDevice));
    {   // packed table of the m7 kernel: thread t = (r, c) of the 52-wide padded half-row; [wave][6][lane][4]
        const int PW = 52, nthr = 16 * PW, KG = 6;
        std::vector<float> pk((size_t)13 * KG * 64 * 4, 0.0f);
        for (int t = 0; t < nthr; ++t) {
            const int r = t / PW, c = t % PW, pix = r * W + c, w = t / 64, l = t % 64;
            for (int q = 0; q < K; ++q) pk[(((size_t)w * KG + q / 4) * 64 + l) * 4 + q % 4] = hphat[(size_t)pix * K + q];
        }
        CK(hipMalloc(&B.p_hat_pk, pk.size() * 4));
        CK(hipMemcpy(B.p_hat_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        // block mapping of the m10 kernel: half-wavefront = 4 columns x 8 rows
        std::fill(pk.begin(), pk.end(), 0.0f);
        for (int t = 0; t < nthr; ++t) {
            const int hw = t / 32, l5 = t % 32, rg = hw / 13, cg = hw % 13;
            const int r = rg * 8 + l5 / 4, c = cg * 4 + l5 % 4, pix = r * W + c, w = t / 64, l = t % 64;
            for (int q = 0; q < K; ++q) pk[(((size_t)w * KG + q / 4) * 64 + l) * 4 + q % 4] = hphat[(size_t)pix * K + q];
        }
        CK(hipMalloc(&B.p_hat_pk2, pk.size() * 4));
        CK(hipMemcpy(B.p_hat_pk2, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&B.ref[i], img_bytes));
    CK(hipMalloc(&B.refgrid, (size_t)N * n * 8)); CK(hipMalloc(&B.refidx, (size_t)N * n * 8));
    CK(hipMalloc(&B.grid, (size_t)N * n * 8)); CK(hipMalloc(&B.idx, (size_t)N * n * 8));
    CK(hipMalloc(&B.trace, 4096 * 16 * 8)); CK(hipMalloc(&B.trace2, 4096 * 2 * 8)); CK(hipMemset(B.trace2, 0, 4096 * 16));

    auto prod = [&](int set, float* out, float* grid, int32_t* idx) {
        int rc = warp_fwd(B.in[set], C, H, W, nullptr, 0, 0, 0, B.ctrl[set], nullptr, B.inv, B.p_hat, K, nullptr,
                          B.p_hat_t, 1, N, F, H, W, out, nullptr, grid, idx, nullptr);
        if (rc) { printf("tpspp_warp_fwd rc=%d\n", rc); exit(1); }
    };
    // references: set 0 (bench-like) and set 1 (nasty), the latter with grid + idx
    prod(0, B.ref[0], nullptr, nullptr);
    prod(1, B.ref[1], B.refgrid, B.refidx);
    CK(hipDeviceSynchronize());
    std::vector<float> href[2] = {std::vector<float>((size_t)N * C * n), std::vector<float>((size_t)N * C * n)};
    std::vector<float> hrefgrid((size_t)N * n * 2); std::vector<int32_t> hrefidx((size_t)N * n * 2);
    for (int i = 0; i < 2; ++i) CK(hipMemcpy(href[i].data(), B.ref[i], img_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hrefgrid.data(), B.refgrid, hrefgrid.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hrefidx.data(), B.refidx, hrefidx.size() * 4, hipMemcpyDeviceToHost));

    const double bytes = (double)N * (2.0 * C * n * 4 + F * 2 * 4);
    auto report = [&](const char* name, float us) {
        printf("%-58s %7.2f us/launch  %6.3f TB/s  frac %.3f\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
        fflush(stdout);
    };

    if (!*only) {
        // ---- launch floor / dispatch spread ----
        long long* sp; CK(hipMalloc(&sp, 8192 * 16));
        struct Shape { int blocks, threads, lds; };
        for (Shape s : {Shape{256, 1024, 92000}, Shape{256, 512, 92000}, Shape{256, 448, 92000}, Shape{256, 256, 92000},
                        Shape{512, 256, 67000}, Shape{512, 512, 45000}, Shape{1024, 256, 0}, Shape{512, 128, 67000}}) {
            CK(hipFuncSetAttribute((const void*)spread_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            float us = period_us([&](int) { hipLaunchKernelGGL(spread_k, dim3(s.blocks), dim3(s.threads), s.lds, 0, sp); }, 500);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(s.blocks * 2);
            CK(hipMemcpy(h.data(), sp, h.size() * 8, hipMemcpyDeviceToHost));
            long long mn = 1LL << 62, mxv = 0;
            for (int b = 0; b < s.blocks; ++b) { mn = std::min(mn, h[b * 2]); mxv = std::max(mxv, h[b * 2]); }
            printf("empty %4d x %4d thr, %5d B LDS: period %.2f us, starts spread %.2f us\n", s.blocks, s.threads, s.lds, us, (mxv - mn) / 100.0);
        }
        // ---- copies of the same bytes, by store policy ----
        const int n4 = (int)(img_bytes / 16);
        for (int blocks : {1024, 2048}) {
            char nm[96];
            snprintf(nm, 96, "copy %d x 256, plain stores", blocks);
            report(nm, period_us([&](int i) { hipLaunchKernelGGL(copy_k<0>, dim3(blocks), dim3(256), 0, 0, (const v4f*)B.in[i % SETS], (char*)B.out[i % SETS], n4); }, 1000));
            snprintf(nm, 96, "copy %d x 256, nt stores", blocks);
            report(nm, period_us([&](int i) { hipLaunchKernelGGL(copy_k<1>, dim3(blocks), dim3(256), 0, 0, (const v4f*)B.in[i % SETS], (char*)B.out[i % SETS], n4); }, 1000));
            snprintf(nm, 96, "copy %d x 256, sc1 stores", blocks);
            report(nm, period_us([&](int i) { hipLaunchKernelGGL(copy_k<2>, dim3(blocks), dim3(256), 0, 0, (const v4f*)B.in[i % SETS], (char*)B.out[i % SETS], n4); }, 1000));
            snprintf(nm, 96, "copy %d x 256, sc0 sc1 stores", blocks);
            report(nm, period_us([&](int i) { hipLaunchKernelGGL(copy_k<3>, dim3(blocks), dim3(256), 0, 0, (const v4f*)B.in[i % SETS], (char*)B.out[i % SETS], n4); }, 1000));
        }
    }

    if (!*only || !strcmp(only, "copy")) {
        // block-contiguous copies with the launch-to-launch gap (wall clock): does a store policy avoid the end-of-kernel flush?
        long long* tr; CK(hipMalloc(&tr, 4 * 256 * 2 * 8));
        auto run = [&](const char* name, auto kern) {
            const int per_block4 = (int)(img_bytes / 16 / 256);
            const float us = period_us([&](int i) { hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, (const v4f*)B.in[i % SETS], (char*)B.out[i % SETS], per_block4, (long long*)nullptr); }, 1000);
            for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, (const v4f*)B.in[(l + 3) % SETS], (char*)B.out[(l + 3) % SETS], per_block4, tr + l * 512);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(4 * 512); CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
            long long s0[4], e1[4];
            for (int l = 0; l < 4; ++l) { s0[l] = 1LL << 62; e1[l] = 0; for (int b = 0; b < 256; ++b) { s0[l] = std::min(s0[l], h[l * 512 + b * 2]); e1[l] = std::max(e1[l], h[l * 512 + b * 2 + 1]); } }
            printf("%-44s period %.2f us; body %.2f us; gap %.2f us\n", name, us, (e1[2] - s0[2]) / 100.0, (s0[3] - e1[2]) / 100.0);
        };
        run("copy 256x1024 blk, plain ld, plain st", copy_blk_k<0, 0>);
        run("copy 256x1024 blk, plain ld, nt st", copy_blk_k<1, 0>);
        run("copy 256x1024 blk, plain ld, sc1 st", copy_blk_k<2, 0>);
        run("copy 256x1024 blk, plain ld, sc0sc1 st", copy_blk_k<3, 0>);
        run("copy 256x1024 blk, nt ld, plain st", copy_blk_k<0, 1>);
        run("copy 256x1024 blk, nt ld, nt st", copy_blk_k<1, 1>);
        run("copy 256x1024 blk, nt ld, sc1 st", copy_blk_k<2, 1>);
    }

    if (!strcmp(only, "dma")) {
        long long* tr; CK(hipMalloc(&tr, 256 * 16 * 2 * 8));
        CK(hipFuncSetAttribute((const void*)dma_probe_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)dma_probe_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        struct Cfg { int waves, per_wave, nt, regs; };
        for (Cfg c : {Cfg{1, 1, 1, 0}, Cfg{3, 1, 1, 0}, Cfg{3, 4, 1, 0}, Cfg{3, 13, 1, 0}, Cfg{3, 25, 1, 0}, Cfg{3, 13, 0, 0}, Cfg{3, 25, 0, 0},
                      Cfg{6, 13, 1, 0}, Cfg{12, 6, 1, 0}, Cfg{3, 13, 1, 1}, Cfg{3, 25, 1, 1}, Cfg{12, 6, 1, 1}, Cfg{1, 1, 1, 1}}) {
            std::vector<double> iss, land;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemset(tr, 0, 256 * 16 * 2 * 8));
                // touch something else first so that the clocks are up and the previous set is cold
                hipLaunchKernelGGL(copy_k<0>, dim3(1024), dim3(256), 0, 0, (const v4f*)B.in[(rep + 7) % SETS], (char*)B.out[(rep + 7) % SETS], (int)(img_bytes / 16));
                auto kern = c.nt ? dma_probe_k<1> : dma_probe_k<0>;
                hipLaunchKernelGGL(kern, dim3(256), dim3(c.waves * 64), 80 * 1024, 0, B.in[rep % SETS], c.per_wave, 76800, tr, c.regs);
                CK(hipDeviceSynchronize());
                if (rep < 2) continue;
                std::vector<long long> h(256 * 16 * 2); CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                for (int b = 0; b < 256; ++b) for (int w = 0; w < c.waves; ++w) { iss.push_back(h[(b * 16 + w) * 2] / 100.0); land.push_back(h[(b * 16 + w) * 2 + 1] / 100.0); }
            }
            std::sort(iss.begin(), iss.end()); std::sort(land.begin(), land.end());
            printf("probe waves=%2d per_wave=%2d (%5.1f KB/CU) nt=%d %s: issued med %.2f us; landed p10 %.2f med %.2f p90 %.2f max %.2f us (s_memtime at 100 MHz)\n", c.waves, c.per_wave,
                   c.waves * c.per_wave * 1.0, c.nt, c.regs ? "regs" : "lds ", iss[iss.size() / 2], land[land.size() / 10], land[land.size() / 2], land[land.size() * 9 / 10], land.back());
        }
        return 0;
    }

    // ---- production kernel ----
    report("production tpspp_warp_fwd (lds-mirror)", period_us([&](int i) { prod(i % SETS, B.out[i % SETS], nullptr, nullptr); }, iters));
    {
        CK(hipMemset(B.trace, 0, 4096 * 64));
        set_trace(B.trace); prod(3, B.out[3], nullptr, nullptr); CK(hipDeviceSynchronize()); set_trace(nullptr);
        trace_report(B, 256);
    }

    // the library's image-pair kernel through the C ABI (prepared table)
    float* prepared = nullptr;
    {
        auto prep_floats = (size_t (*)(int, int, int))dlsym(lib, "tpspp_prepared_table_floats");
        auto prep = (int (*)(const float*, int, int, int, int, float*, void*))dlsym(lib, "tpspp_prepare_mirror_table");
        if (prep_floats && prep) {
            CK(hipMalloc(&prepared, prep_floats(H, W, F) * 4));
            if (prep(B.p_hat, K, H, W, F, prepared, nullptr)) { printf("prepare failed\n"); return 1; }
            CK(hipDeviceSynchronize());
        }
    }
    Variant libpair{"lib pair kernel (tpspp_warp_fwd, prepared table)",
        [&, warp_fwd, prepared](const Bufs& Bf, int s, float* o, float* g, int32_t* ix, long long*, hipStream_t) {
            warp_fwd(Bf.in[s], C, H, W, nullptr, 0, 0, 0, Bf.ctrl[s], nullptr, Bf.inv, Bf.p_hat, K, nullptr, prepared, 1 | 8, N, F, H, W, o, nullptr, g, ix, nullptr); }};
    std::vector<Variant> vars = {
        M15VAR(3, 1, 1, 6, 4), M18VAR(3, 1, 1, 6, 4), M18VAR(3, 1, 1, 6, 2), M18VAR(3, 0, 1, 6, 4),
    };
    if (prepared) vars.push_back(libpair);
    std::vector<float> hout((size_t)N * C * n), hgrid((size_t)N * n * 2); std::vector<int32_t> hidx((size_t)N * n * 2);
    for (auto& v : vars) {
        if (*only && v.name.find(only) == std::string::npos) continue;
        printf("-- %s\n", v.name.c_str()); fflush(stdout);
        // correctness: set 0 plain, set 1 with grid + idx
        bool ok = true; size_t bad = 0;
        const bool is_m = (v.name[0] == 'm' && v.name[1] == ' ');
        const bool dbg = v.name[0] == 'm' && v.name[1] != '3' && v.name[1] != '4' && v.name[1] != '5' && v.name[1] != '6' && v.name[1] != '7' && v.name[1] != '8' && v.name[1] != '9' && v.name[1] != '1' && v.name.find("dbg=0") == std::string::npos && v.name.find("dbg=4") == std::string::npos && v.name.find("dbg=8") == std::string::npos && v.name.find("dbg=16") == std::string::npos && v.name.find("dbg=24") == std::string::npos;
        for (int s = 0; s < 2 && !dbg; ++s) {
            CK(hipMemset(B.out[s], 0xff, img_bytes));
            if (s == 1) { CK(hipMemset(B.grid, 0xff, (size_t)N * n * 8)); CK(hipMemset(B.idx, 0xff, (size_t)N * n * 8)); }
            v.run(B, s, B.out[s], (s && !is_m) ? B.grid : nullptr, (s && !is_m) ? B.idx : nullptr, nullptr, 0);
            CK(hipDeviceSynchronize());
            printf("   ran set %d\n", s); fflush(stdout);
            CK(hipMemcpy(hout.data(), B.out[s], img_bytes, hipMemcpyDeviceToHost));
            if (memcmp(hout.data(), href[s].data(), img_bytes) != 0) {
                ok = false;
                for (size_t i = 0; i < hout.size(); ++i) if (memcmp(&hout[i], &href[s][i], 4)) { if (!bad) printf("   first diff set %d at %zu: %g vs %g\n", s, i, hout[i], href[s][i]); ++bad; }
            }
            if (s == 1 && !is_m) {
                CK(hipMemcpy(hgrid.data(), B.grid, hgrid.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hidx.data(), B.idx, hidx.size() * 4, hipMemcpyDeviceToHost));
                if (memcmp(hgrid.data(), hrefgrid.data(), hgrid.size() * 4) || memcmp(hidx.data(), hrefidx.data(), hidx.size() * 4)) { ok = false; printf("   grid/idx differ\n"); }
            }
        }
        const float us = period_us([&](int i) { v.run(B, i % SETS, B.out[i % SETS], nullptr, nullptr, nullptr, 0); }, iters);
        char nm[160]; snprintf(nm, 160, "%s %s", v.name.c_str(), dbg ? "[dbg]" : ok ? "[bit-exact]" : "[MISMATCH]");
        report(nm, us);
        if (!ok) printf("   %zu differing output words\n", bad);
        CK(hipMemset(B.trace, 0, 4096 * 128));
        if (v.name.compare(0, 3, "lib") == 0) continue;
        if (v.name.compare(0, 2, "m3") == 0 || v.name.compare(0, 2, "m4") == 0 || v.name.compare(0, 2, "m5") == 0 || v.name.compare(0, 2, "m6") == 0 || v.name.compare(0, 2, "m7") == 0 || v.name.compare(0, 2, "m8") == 0 || v.name.compare(0, 2, "m9") == 0 || v.name.compare(0, 3, "m10") == 0 || v.name.compare(0, 3, "m11") == 0 || v.name.compare(0, 3, "m12") == 0 || v.name.compare(0, 3, "m13") == 0 || v.name.compare(0, 3, "m14") == 0 || v.name.compare(0, 3, "m15") == 0 || v.name.compare(0, 3, "m16") == 0 || v.name.compare(0, 3, "m17") == 0 || v.name.compare(0, 3, "m18") == 0) {
            const int L = 4;
            for (int l = 0; l < 20; ++l) v.run(B, l % SETS, B.out[l % SETS], nullptr, nullptr, nullptr, 0);
            for (int l = 0; l < L; ++l) v.run(B, (l + 3) % SETS, B.out[(l + 3) % SETS], nullptr, nullptr, B.trace + (size_t)l * 256 * 16, 0);
            CK(hipDeviceSynchronize());
            trace_report_m3(B, 256, L);
            continue;
        }
        if (v.name[0] == 'm') {
            const int L = 4;
            for (int l = 0; l < 20; ++l) v.run(B, l % SETS, B.out[l % SETS], nullptr, nullptr, nullptr, 0);
            for (int l = 0; l < L; ++l) v.run(B, (l + 3) % SETS, B.out[(l + 3) % SETS], nullptr, nullptr, B.trace + (size_t)l * 256 * 16, 0);
            CK(hipDeviceSynchronize());
            trace_report_m(B, 256, L);
            continue;
        }
        v.run(B, 3, B.out[3], nullptr, nullptr, B.trace, 0); CK(hipDeviceSynchronize());
        const int imgs = v.name.find("imgs=1") != std::string::npos ? 1 : 2;
        trace_report(B, (N + imgs - 1) / imgs);
    }
    // ---- interleaved re-timing: box / clock drift hits every variant of a round alike ----
    {
        const int rounds = 7, it = 700;
        std::vector<std::vector<float>> t(vars.size() + 1);
        for (int r = 0; r < rounds; ++r) {
            t[0].push_back(period_us([&](int i) { prod(i % SETS, B.out[i % SETS], nullptr, nullptr); }, it, 20));
            for (size_t k = 0; k < vars.size(); ++k) {
                if (*only && vars[k].name.find(only) == std::string::npos) continue;
                t[k + 1].push_back(period_us([&](int i) { vars[k].run(B, i % SETS, B.out[i % SETS], nullptr, nullptr, nullptr, 0); }, it, 20));
            }
        }
        printf("interleaved, %d rounds x %d launches: min / median us per launch\n", rounds, it);
        for (size_t k = 0; k <= vars.size(); ++k) {
            if (t[k].empty()) continue;
            std::sort(t[k].begin(), t[k].end());
            printf("  %-56s %6.2f / %6.2f   frac %.3f\n", k ? vars[k - 1].name.c_str() : "production", t[k][0], t[k][t[k].size() / 2], bytes / t[k][t[k].size() / 2] / 1e6 / 8.0);
        }
    }
    return 0;
}
