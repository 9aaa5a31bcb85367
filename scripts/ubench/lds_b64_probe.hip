// Does ds_read_b64 work on a 4-byte-aligned (not 8-byte-aligned) LDS address on gfx950, and at what rate?
// The bilinear taps (x0, x0 + 1) of a pixel are adjacent floats: one ds_read_b64 could replace two ds_read_b32.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/lds_b64_probe.hip -o scripts/ubench/lds_b64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: two ds_read_b32 (addr, addr + 4); 1: ds_read_b64 at addr; 2: ds_read2_b32 offset1:1
template <int MODE>
__global__ void __launch_bounds__(1024) k(const int* __restrict__ offs, float* out, int iters, int check)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 16000; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    unsigned base = (unsigned)(size_t)sm + 4u * (unsigned)offs[threadIdx.x];
    float acc0 = 0.f, acc1 = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            float a, b;
            if (MODE == 0) {
                asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(base), "n"(j * 12800 % 38400), "n"(j * 12800 % 38400 + 4));
            } else if (MODE == 1) {
                v2f x; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(base), "n"(j * 12800 % 38400)); a = x.x; b = x.y;
            } else {
                v2f x; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x) : "v"(base), "n"(j * 3200 % 200), "n"(j * 3200 % 200 + 1)); a = x.x; b = x.y;
            }
            asm volatile("s_waitcnt lgkmcnt(8)");
            acc0 += a; acc1 += b;
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (check) { out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = acc0; out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = acc1; }
    else if (acc0 + acc1 == 12345.f) out[0] = acc0;
}

int main()
{
    const int T = 832, blocks = 256;
    int* doffs; float* dout; hipMalloc(&doffs, T * 4); hipMalloc(&dout, blocks * T * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Pat { const char* name; std::vector<int> o; };
    std::vector<Pat> pats;
    auto mk = [&](const char* nm, auto f) { Pat p; p.name = nm; for (int t = 0; t < T; ++t) p.o.push_back(f(t)); pats.push_back(p); };
    // the pair kernel's thread -> pixel mapping: half-wavefront = 4 columns x 8 rows at a row pitch of 100 floats
    auto pix = [](int t, int dx) { int hw = t >> 5, l5 = t & 31, rg = hw / 13, cg = hw % 13; int r = rg * 8 + (l5 >> 2), c = cg * 4 + (l5 & 3); return r * 100 + (c + dx > 98 ? 98 : c + dx); };
    mk("pair mapping, even shift (8-byte aligned where c is even)", [&](int t) { return pix(t, 0); });
    mk("pair mapping, +1 shift", [&](int t) { return pix(t, 1); });
    mk("all lanes 8-byte aligned, consecutive pairs", [&](int t) { return 2 * (t % 64) + 200 * (t / 64); });
    mk("all lanes odd (4-byte aligned only), consecutive pairs", [&](int t) { return 2 * (t % 64) + 1 + 200 * (t / 64); });
    mk("lanes consecutive dwords (overlapping pairs)", [&](int t) { return (t % 64) + 200 * (t / 64); });
    for (auto& p : pats) {
        hipMemcpy(doffs, p.o.data(), T * 4, hipMemcpyHostToDevice);
        printf("%s\n", p.name);
        std::vector<float> ref;
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(dout, 0, blocks * T * 8);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(T), 64000, 0, doffs, dout, 1, 1);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(T), 64000, 0, doffs, dout, 1, 1);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(T), 64000, 0, doffs, dout, 1, 1);
            hipDeviceSynchronize();
            std::vector<float> h(T * 2); hipMemcpy(h.data(), dout, T * 8, hipMemcpyDeviceToHost);
            bool same = true;
            if (mode == 0) ref = h; else if (mode == 1) same = h == ref;
            float ms = 0;
            const int iters = 2000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(T), 64000, 0, doffs, dout, iters, 0);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(T), 64000, 0, doffs, dout, iters, 0);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(T), 64000, 0, doffs, dout, iters, 0);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipError_t e = hipGetLastError();
            const double pairs = (double)iters * 12 * 13;   // wavefront-level pair reads per CU
            printf("   %-14s %8.3f ms  %6.1f ns per wavefront pair-read per CU  %s %s\n", mode == 0 ? "2 x b32" : mode == 1 ? "b64" : "read2_b32 (x, x+1; other offsets)", ms,
                   ms * 1e6 / pairs, mode == 1 ? (same ? "[same values as 2 x b32]" : "[VALUES DIFFER]") : "", e == hipSuccess ? "" : hipGetErrorString(e));
        }
    }
    return 0;
}
