// How long does the dispatcher take to start every workgroup of a launch?  (wall_clock64 = 100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int VG>
__global__ void __launch_bounds__(1024) k(long long* out, float* sink)
{
    extern __shared__ float sm[];
    long long t = wall_clock64();
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) acc[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(acc[i]));
    float s = 0; 
#pragma unroll
    for (int i = 0; i < VG; ++i) s += acc[i];
    if (threadIdx.x == 0) { out[blockIdx.x] = t; }
    if (s == -1.f) { sm[threadIdx.x] = s; sink[0] = sm[0]; }
}
int main()
{
    long long* d; hipMalloc(&d, 8192 * 8); float* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)&k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&k<56>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto run = [&](const char* nm, auto kern, int blocks, int threads, int lds) {
        double worst = 0, med = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, sink);
            hipDeviceSynchronize();
            std::vector<long long> h(blocks); hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            if (rep) { worst = std::max(worst, (h.back() - h[0]) / 100.0); med = (h[blocks / 2] - h[0]) / 100.0; }
        }
        printf("%-28s blocks %5d x %4d thr, LDS %6d B: starts spread %.2f us (median %.2f)\n", nm, blocks, threads, lds, worst, med);
    };
    run("8 VGPR", k<8>, 256, 1024, 0);
    run("8 VGPR", k<8>, 256, 1024, 80000);
    run("56 VGPR", k<56>, 256, 1024, 0);
    run("56 VGPR", k<56>, 256, 1024, 80000);
    run("56 VGPR", k<56>, 512, 512, 40000);
    run("56 VGPR", k<56>, 1024, 256, 20000);
    run("56 VGPR", k<56>, 1024, 256, 0);
    run("8 VGPR", k<8>, 1024, 256, 0);
    run("8 VGPR", k<8>, 4096, 64, 0);
    return 0;
}
