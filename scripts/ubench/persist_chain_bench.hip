// The decoder's chain of dependent step projections (gemm_chain_bench.hip: 6.35 us per launch) as ONE persistent launch:
// 16 token blocks x 16 workgroups; the 16 workgroups of a token block ("cluster") exchange their 4 KB output tiles every
// phase, clusters never talk to each other (images are independent).  Per phase and workgroup: the 64 KB weight slice is
// requested BEFORE the cluster barrier (it depends on nothing), then
//   barrier   every workgroup of the cluster has stored the previous phase's tile: stores are write-through (sc0 sc1),
//             drained (vmcnt(0)), one lane takes a ticket on the cluster's counter (agent-scope relaxed atomic) and polls it
//             with sc1 loads + s_sleep (MI355X_MICROARCH.md: no release / acquire fences -- those write back / invalidate
//             whole caches);
//   X         the cluster's 32 x 512 fp32 activations with sc0 sc1 loads (they bypass this CU's L1 and this XCD's L2, which
//             may hold the lines of two phases ago);
//   product   24 v_mfma_f32_32x32x16_bf16 per wavefront + the 4-way reduction through LDS, 4 KB store.
// Every tile carries its phase number; a consumer counts the X words that carry the wrong one (staleness / race detector).
// Spins are bounded: a timeout raises a flag and the workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 load16_sys(const float* p)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store16_sys(float* p, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// MODE 0: barrier only; 1: + X loads (sc0 sc1) and the 4 KB store; 2: + weight loads and the product
// PRE: weights requested before the barrier (1) or behind it (0)
// STAGE: 1 = X arrives as whole rows (1 KB per wavefront instruction, sc0 sc1) and is re-read from LDS in fragment order
template <int MODE, int PRE, int STAGE = 0, int MAP = 0, int ST = 0, int AT = 0>
__global__ void __launch_bounds__(256) pk(float* bufA, float* bufB, const u32x4* __restrict__ W, int* counters, int phases,
                                          int* errors)
{
    __shared__ float sRed[4][16][64];
    __shared__ __attribute__((aligned(16))) float sX[STAGE ? 32 * 516 : 4];
    __shared__ int sOk;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // cluster = token block; the 16 workgroups of a cluster sit 16 blocks apart in launch order -> on 2 XCDs (block % 8)
    // MAP 0: the 16 workgroups of a cluster 16 blocks apart -> all on XCD tb % 8; MAP 1: 16 consecutive blocks -> two per XCD
    const int tb = MAP ? blockIdx.x >> 4 : blockIdx.x & 15, ct = MAP ? blockIdx.x & 15 : blockIdx.x >> 4;
    int* cnt = counters + tb * 32;                          // one counter per cluster, 128 bytes apart
    int bad = 0;
    for (int p = 0; p < phases; ++p) {
        const float* X = (p & 1) ? bufB : bufA;
        float* out = (p & 1) ? bufA : bufB;
        const u32x4* wp = W + (size_t)(p % 36) * 65536 + ((size_t)(ct * 32 + wv * 8) * 4 + half) * 32 + l31;
        u32x4 ah[8], al[8];
        if (MODE >= 2 && PRE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ah[j] = wp[j * 128]; al[j] = wp[j * 128 + 64]; }
        }
        // ---- cluster barrier: everybody has stored phase p - 1 ----
        if (p > 0) {
            __syncthreads();                                // (this workgroup's stores were drained below)
            if (tid == 0) {
                if (AT) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (executes in this XCD's L2)
                else __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int ok = 0;
                for (int spin = 0; spin < (1 << 22); ++spin) {
                    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 16 * p) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                sOk = ok;
            }
            __syncthreads();
            if (!sOk) { if (tid == 0) atomicAdd(errors + 1, 1); return; }
        }
        f32x4 v = {0.f, 2.f, 3.f, 4.f};
        if (MODE >= 1) {
            f32x4 xa[8][2];
            if (STAGE) {
                // wavefront w: rows 8 w .. 8 w + 7 of the cluster's 32 x 512 block, two 1-KB instructions per row
                f32x4 t[16];
                const float* xr = X + (size_t)(tb * 32 + 8 * wv) * 512 + 4 * lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = load16_sys(xr + (size_t)(i >> 1) * 512 + 256 * (i & 1));
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                               "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15])
                             :: "memory");
                __syncthreads();                            // (the previous phase's fragment reads are done)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<f32x4*>(&sX[(8 * wv + (i >> 1)) * 516 + 256 * (i & 1) + 4 * lane]) = t[i];
                __syncthreads();
                const float* xs = &sX[l31 * 516 + 16 * (wv * 8) + 8 * half];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xa[j][0] = *reinterpret_cast<const f32x4*>(xs + 16 * j);
                    xa[j][1] = *reinterpret_cast<const f32x4*>(xs + 16 * j + 4);
                }
            } else {
            const float* xp = X + (size_t)(tb * 32 + l31) * 512 + 16 * (wv * 8) + 8 * half;
#pragma unroll
            for (int j = 0; j < 8; ++j) { xa[j][0] = load16_sys(xp + 16 * j); xa[j][1] = load16_sys(xp + 16 * j + 4); }
            }
            if (MODE >= 2 && !PRE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ah[j] = wp[j * 128]; al[j] = wp[j * 128 + 64]; }
            }
            // (the asm loads are invisible to the compiler's wait-count tracking: the wait is tied to every loaded register so
            // that no use -- not even a register copy -- can be scheduled in front of it)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]), "+v"(xa[2][0]), "+v"(xa[2][1]),
                           "+v"(xa[3][0]), "+v"(xa[3][1]), "+v"(xa[4][0]), "+v"(xa[4][1]), "+v"(xa[5][0]), "+v"(xa[5][1]),
                           "+v"(xa[6][0]), "+v"(xa[6][1]), "+v"(xa[7][0]), "+v"(xa[7][1])
                         :: "memory");
            // every X word was written in phase p - 1 as the value (p - 1) (phase 0 reads the initial zeros)
            const float want = (float)(p > 0 ? p - 1 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad += (xa[j][0][e] != want) + (xa[j][1][e] != want);
            if (MODE >= 2) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u32x4 b;
                    b[0] = __builtin_bit_cast(unsigned, xa[j][0][0]); b[1] = __builtin_bit_cast(unsigned, xa[j][0][2]);
                    b[2] = __builtin_bit_cast(unsigned, xa[j][1][0]); b[3] = __builtin_bit_cast(unsigned, xa[j][1][2]);
                    const bf16x8 A = __builtin_bit_cast(bf16x8, ah[j]), A2 = __builtin_bit_cast(bf16x8, al[j]), B = __builtin_bit_cast(bf16x8, b);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
                }
                __syncthreads();                            // (the previous phase's partial sums have been read)
#pragma unroll
                for (int r = 0; r < 16; ++r) sRed[wv][r][lane] = acc[r];
                __syncthreads();
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s += (sRed[0][4 * wv + e][lane] + sRed[1][4 * wv + e][lane]) + (sRed[2][4 * wv + e][lane] + sRed[3][4 * wv + e][lane]);
                v[0] = s * 0.0f;                            // (weights are zero: keeps the dependence, not the value)
            }
            const float ph = (float)p + v[0];
            if (ST) *reinterpret_cast<f32x4*>(out + (size_t)(tb * 32 + l31) * 512 + ct * 32 + 8 * wv + 4 * half) = f32x4{ph, ph, ph, ph};   // plain: the line stays in this XCD's L2
            else store16_sys(out + (size_t)(tb * 32 + l31) * 512 + ct * 32 + 8 * wv + 4 * half, f32x4{ph, ph, ph, ph});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE, int PRE, int STAGE = 0, int MAP = 0, int ST = 0, int AT = 0> void run(const char* name, float* a, float* b, u32x4* w, int* counters, int* errors)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(a, 0, 512 * 512 * 4); hipMemset(b, 0, 512 * 512 * 4);
        hipMemset(counters, 0, 16 * 32 * 4); hipMemset(errors, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((pk<MODE, PRE, STAGE, MAP, ST, AT>), dim3(256), dim3(256), 0, 0, a, b, w, counters, n, errors);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int err[2]; hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost);
        printf("%-34s %6.2f us per phase   stale words %d, timeouts %d\n", name, ms * 1e3f / n, err[0], err[1]);
    }
}

int main()
{
    float *a, *b; u32x4* w; int *counters, *errors;
    hipMalloc(&a, 512 * 512 * 4); hipMalloc(&b, 512 * 512 * 4);
    hipMalloc(&w, (size_t)36 * 65536 * 16);
    hipMalloc(&counters, 16 * 32 * 4); hipMalloc(&errors, 8);
    hipMemset(w, 0, (size_t)36 * 65536 * 16);
    run<0, 0>("cluster barrier only", a, b, w, counters, errors);
    run<1, 0>("+ X (sc0 sc1) + 4 KB store", a, b, w, counters, errors);
    run<2, 0>("+ W behind the barrier + product", a, b, w, counters, errors);
    run<2, 1>("+ W before the barrier + product", a, b, w, counters, errors);
    run<1, 0, 1>("X as whole rows via LDS + store", a, b, w, counters, errors);
    run<2, 1, 1>("W before barrier, X via LDS, product", a, b, w, counters, errors);
    // round 5: placement and store flavour (the rows above: cluster on ONE XCD, sc0 sc1 stores, agent atomics)
    run<2, 1, 1, 1, 0, 0>("  cluster spread over 8 XCDs", a, b, w, counters, errors);
    run<2, 1, 1, 0, 1, 0>("  one XCD, PLAIN stores", a, b, w, counters, errors);
    run<2, 1, 1, 0, 1, 1>("  one XCD, plain stores, wg atomics", a, b, w, counters, errors);
    run<0, 0, 0, 1, 0, 0>("barrier only, spread", a, b, w, counters, errors);
    run<0, 0, 0, 0, 0, 1>("barrier only, one XCD, wg atomics", a, b, w, counters, errors);
    return 0;
}
