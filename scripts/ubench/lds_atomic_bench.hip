// LDS atomic / read-modify-write rates on gfx950: ds_add_f32 vs ds_add_u32 vs ds_read + add + ds_write, by address pattern.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/lds_atomic_bench.hip -o scripts/ubench/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: ds_add_f32, 1: ds_add_u32, 2: non-atomic read-modify-write (one wavefront per plane: no race), 3: ds_read only,
// 4: float add as a compare-and-swap loop (ds_cmpst_rtn_b32), 5: ds_add_u64 (fixed point)
// PAT 0: lane -> consecutive words, 1: stride 2, 2: pairs of lanes share a word, 3: 4 lanes share a word, 4: pseudo-random in 4096 words
template <int MODE, int PAT>
__global__ void __launch_bounds__(256) k(float* out, int iters, long long* cyc)
{
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* pl = sm + wv * 4096;                  // a 16-KB plane per wavefront
    for (int i = lane; i < 4096; i += 64) pl[i] = 0.0f;
    __syncthreads();
    int a;
    if (PAT == 0) a = lane; else if (PAT == 1) a = 2 * lane; else if (PAT == 2) a = lane >> 1; else if (PAT == 3) a = lane >> 2;
    else a = (lane * 2654435761u >> 7) & 4095;
    const long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ad = (a + 64 * u + 17 * it) & 4095;
            if (MODE == 0) atomicAdd(pl + ad, 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(pl) + ad, 1u);
            else if (MODE == 2) { const float v = pl[ad]; pl[ad] = v + 1.0f; }
            else if (MODE == 4) {
                unsigned* up = reinterpret_cast<unsigned*>(pl) + ad;
                unsigned old = *up, assumed;
                do { assumed = old; old = atomicCAS(up, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
            }
            else if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(pl) + (ad >> 1), 1ull);
            else if (MODE == 6) atomicAdd(reinterpret_cast<double*>(pl) + (ad >> 1), 1.0);
            else if (MODE == 7) unsafeAtomicAdd(reinterpret_cast<double*>(pl) + (ad >> 1), 1.0);
            else acc += pl[ad];
        }
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = acc;
    for (int i = lane; i < 4096; i += 64) s += pl[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int PAT> static int run(const char* name, float* out, long long* cyc)
{
    const int iters = 2000, blocks = 256;
    CK(hipFuncSetAttribute((const void*)k<MODE, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpc : {1, 2}) {
        hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks * wpc), dim3(256), 64 * 1024, 0, out, iters, cyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks * wpc), dim3(256), 64 * 1024, 0, out, iters, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)blocks * wpc * 256 * iters * 8;
        // per CU: wavefront instructions per us and lane-ops per clock at a nominal 2.4 GHz
        printf("%-44s %d wg/CU: %7.3f ms  %7.1f G lane-ops/s chip  = %5.2f lane-ops/clk/CU\n", name, wpc, ms, ops / ms / 1e6,
               ops / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}

int main()
{
    float* out; long long* cyc; CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    run<3, 0>("ds_read_b32, consecutive", out, cyc);
    run<0, 0>("ds_add_f32, consecutive", out, cyc);
    run<0, 1>("ds_add_f32, stride 2", out, cyc);
    run<0, 2>("ds_add_f32, pairs share a word", out, cyc);
    run<0, 3>("ds_add_f32, 4 lanes share a word", out, cyc);
    run<0, 4>("ds_add_f32, scattered", out, cyc);
    run<6, 0>("atomicAdd(double) in LDS, consecutive 8-byte words", out, cyc);
    run<6, 2>("atomicAdd(double) in LDS, pairs share a word", out, cyc);
    run<7, 0>("unsafeAtomicAdd(double) in LDS, consecutive", out, cyc);
    run<1, 0>("ds_add_u32, consecutive", out, cyc);
    run<1, 2>("ds_add_u32, pairs share a word", out, cyc);
    run<1, 4>("ds_add_u32, scattered", out, cyc);
    run<4, 0>("CAS-loop float add, consecutive", out, cyc);
    run<4, 1>("CAS-loop float add, stride 2", out, cyc);
    run<4, 2>("CAS-loop float add, pairs share a word", out, cyc);
    run<4, 3>("CAS-loop float add, 4 lanes share a word", out, cyc);
    run<4, 4>("CAS-loop float add, scattered", out, cyc);
    run<5, 1>("ds_add_u64, consecutive 8-byte words", out, cyc);
    run<5, 0>("ds_add_u64, pairs share a word", out, cyc);
    run<2, 0>("read + add + write, consecutive", out, cyc);
    run<2, 1>("read + add + write, stride 2", out, cyc);
    run<2, 4>("read + add + write, scattered", out, cyc);
    return 0;
}
