// Micro-benchmark: what does a plain coalesced copy of the bench workload's bytes cost per launch?
// (19.66 MB in + 19.66 MB out, rotating over buffers > 256 MB so the Infinity Cache cannot hold them)
// Gives the practical ceiling the warp kernel is judged against at batch 512.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) copy_k(const float4* __restrict__ src, float4* __restrict__ dst, int n4)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void __launch_bounds__(1024) copy_lds_k(const float4* __restrict__ src, float4* __restrict__ dst, int per_block4)
{
    // each block copies per_block4 float4 through registers in one shot (all loads first)
    const float4* s = src + (size_t)blockIdx.x * per_block4;
    float4* d = dst + (size_t)blockIdx.x * per_block4;
    float4 r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { int q = threadIdx.x + i * 1024; if (q < per_block4) r[i] = s[q]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) { int q = threadIdx.x + i * 1024; if (q < per_block4) d[q] = r[i]; }
}

__global__ void empty_k() {}

int main()
{
    const size_t bytes = 512ull * 3 * 32 * 100 * 4;   // 19.66 MB
    const int n4 = bytes / 16;
    const int nbuf = 16;
    std::vector<float4*> in(nbuf), out(nbuf);
    for (int i = 0; i < nbuf; ++i) { hipMalloc(&in[i], bytes); hipMalloc(&out[i], bytes); hipMemset(in[i], 1, bytes); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 20; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch(i);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters;
        printf("%-44s %8.2f us/launch  %6.3f TB/s (in+out)\n", name, us, 2.0 * bytes / us / 1e6);
    };
    timeit("empty kernel", [&](int i) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(1024), 0, 0); });
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192})
    {
        char nm[64]; snprintf(nm, 64, "grid-stride float4 copy, %d x 256", blocks);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, in[i % nbuf], out[i % nbuf], n4); });
    }
    timeit("256 blocks x 1024 thr, 76.8 KB each, loads-first", [&](int i) {
        hipLaunchKernelGGL(copy_lds_k, dim3(256), dim3(1024), 0, 0, in[i % nbuf], out[i % nbuf], n4 / 256); });
    timeit("hipMemcpyAsync D2D", [&](int i) { hipMemcpyAsync(out[i % nbuf], in[i % nbuf], bytes, hipMemcpyDeviceToDevice, 0); });
    timeit("copy, same buffer (cache-resident)", [&](int i) { hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, 0, in[0], out[0], n4); });
    return 0;
}
