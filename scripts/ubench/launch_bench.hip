// Back-to-back launch period on one stream: the floor under any "one launch per step" kernel.
//   empty kernel, 256 x 1024 threads (the shape of the classic warp launch), with / without 80 KB LDS;
//   the same writing 19.7 MB (the warp's output bytes) so that the kernel boundary has dirty L2 lines.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) empty_k(float* sink)
{
    extern __shared__ float sm[];
    if (threadIdx.x == 2048) sink[0] = sm[0];
}
__global__ void __launch_bounds__(1024) write_k(float4* out, int per_block4)
{
    float4* o = out + (size_t)blockIdx.x * per_block4;
    for (int i = threadIdx.x; i < per_block4; i += blockDim.x) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void __launch_bounds__(1024) copy_k(const float4* in, float4* out, int per_block4)
{
    const float4* s = in + (size_t)blockIdx.x * per_block4;
    float4* o = out + (size_t)blockIdx.x * per_block4;
    for (int i = threadIdx.x; i < per_block4; i += blockDim.x) o[i] = s[i];
}
template <class F> float period_us(F launch, int n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}
int main()
{
    const int sets = 14;
    const size_t bytes = 19701760;                       // 512 x 3 x 32 x 100 x 4, padded to 256 blocks
    const int per_block4 = (int)(bytes / 16 / 256);
    float4 *in, *out; float* sink;
    hipMalloc(&in, bytes * sets); hipMalloc(&out, bytes * sets); hipMalloc(&sink, 4);
    hipMemset(in, 0, bytes * sets);
    hipFuncSetAttribute((const void*)&empty_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("empty 256x1024            : %.2f us/launch\n", period_us([&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(1024), 0, 0, sink); }, 2000));
    printf("empty 256x1024, 80 KB LDS : %.2f us/launch\n", period_us([&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(1024), 80000, 0, sink); }, 2000));
    printf("empty 256x256             : %.2f us/launch\n", period_us([&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, 0, sink); }, 2000));
    printf("write 19.7 MB 256x1024    : %.2f us/launch\n", period_us([&](int i) { hipLaunchKernelGGL(write_k, dim3(256), dim3(1024), 0, 0, out + (size_t)(i % sets) * (bytes / 16), per_block4); }, 2000));
    printf("copy  19.7+19.7 MB 256x1024: %.2f us/launch\n", period_us([&](int i) { hipLaunchKernelGGL(copy_k, dim3(256), dim3(1024), 0, 0, in + (size_t)(i % sets) * (bytes / 16), out + (size_t)(i % sets) * (bytes / 16), per_block4); }, 2000));
    printf("copy  same, 1024x256       : %.2f us/launch\n", period_us([&](int i) { hipLaunchKernelGGL(copy_k, dim3(1024), dim3(256), 0, 0, in + (size_t)(i % sets) * (bytes / 16), out + (size_t)(i % sets) * (bytes / 16), per_block4 / 4); }, 2000));
    return 0;
}
