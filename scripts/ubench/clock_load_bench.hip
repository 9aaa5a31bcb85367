// Which clock does the chip hold under which load?  s_memtime counts shader cycles, wall_clock64 a constant 100 MHz.
// One 768-thread workgroup per CU: wavefronts 0-7 run the convolution's inner loop (v_mfma_f32_32x32x16_bf16 fed by
// ds_read_b128), wavefronts 8-11 optionally stream HBM into LDS with LDS-DMA (1 KB per instruction, a fresh 64 MB region).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(768) k(float* out, long long* clk, const char* src, int chunks, int do_mfma, int do_lds, int do_dma)
{
    extern __shared__ u32x4 s[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) { u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; s[e] = v; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    if (wv >= 8) {
        if (do_dma) {
            const unsigned dst = (unsigned)(size_t)(s + 4096 + (wv - 8) * 256);
            const char* p = src + (((size_t)blockIdx.x * 4 + (wv - 8)) % 1024) * (size_t)(1 << 20) + lane * 16;   // 1 MB windows of a 1 GB buffer
            for (int c = 0; c < chunks; ++c) {
                for (int i = 0; i < 8; ++i)
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst + (i & 3) * 1024), "v"(p + (((size_t)c * 8 + i) & 1023) * 1024) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    } else if (do_mfma) {
        f32x16 acc[2][2];
        for (int f = 0; f < 2; ++f) for (int h = 0; h < 2; ++h) for (int i = 0; i < 16; ++i) acc[f][h][i] = 0.f;
        const u32x4* wb = s + half * 64 + l31;
        const u32x4* pb = s + 1152 + half * 660 + wv * 66 * 2 + l31;
        bf16x8 fa[2], fb[2];
        fa[0] = __builtin_bit_cast(bf16x8, wb[0]); fa[1] = __builtin_bit_cast(bf16x8, wb[32]);
        fb[0] = __builtin_bit_cast(bf16x8, pb[0]); fb[1] = __builtin_bit_cast(bf16x8, pb[32]);
        for (int c = 0; c < chunks; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (do_lds) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    fa[0] = __builtin_bit_cast(bf16x8, wb[tap * 128]); fa[1] = __builtin_bit_cast(bf16x8, wb[tap * 128 + 32]);
                    fb[0] = __builtin_bit_cast(bf16x8, pb[ky * 66 + kx]); fb[1] = __builtin_bit_cast(bf16x8, pb[ky * 66 + kx + 32]);
                }
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[f], acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[f], acc[f][1], 0, 0, 0);
                }
            }
        }
        float t = 0.f;
        for (int f = 0; f < 2; ++f) for (int h = 0; h < 2; ++h) for (int i = 0; i < 16; ++i) t += acc[f][h][i];
        if (t == 12345.f) out[threadIdx.x] = t;
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main()
{
    float* d; hipMalloc(&d, 4096 * 4);
    long long* c; hipMalloc(&c, 256 * 2 * 8);
    char* src; hipMalloc(&src, (size_t)1 << 30); hipMemset(src, 0, (size_t)1 << 30);
    const int chunks = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"MFMA only", "MFMA + LDS fragment reads", "LDS-DMA stream only", "MFMA + LDS reads + LDS-DMA stream"};
    const int cfg[4][3] = {{1, 0, 0}, {1, 1, 0}, {0, 0, 1}, {1, 1, 1}};
    for (int v = 0; v < 4; ++v) {
        float best = 1e9f; double mhz = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(768), (4096 + 1024) * 16, 0, d, c, src, chunks, cfg[v][0], cfg[v][1], cfg[v][2]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            if (ms < best) { best = ms; mhz = (double)h[0] / ((double)h[1] / 100.0); }
        }
        const double mf = 256.0 * 8 * chunks * 36.0 * 32768.0;
        const double gb = 256.0 * 4 * chunks * 8.0 * 1024.0;
        printf("%-36s %.3f ms  shader clock %.0f MHz  %s%.0f TFLOP/s  %s%.2f TB/s\n", names[v], best, mhz,
               cfg[v][0] ? "" : "(", cfg[v][0] ? mf / best / 1e9 : 0.0, cfg[v][2] ? "" : "(", cfg[v][2] ? gb / best / 1e9 : 0.0);
    }
    return 0;
}
