// Is v_mfma_f32_16x16x4_f32 (and 32x32x2) bit-identical to the k-ascending fmaf chain from zero that the reference's
// torch.bmm performs (oracle/tps_oracle.c)?  D[16x16] = sum_k A[i][k] B[k][j], K = 24 as 6 instructions of 4 k.
// Operand layout (cdna_hip_programming.md): A: lane l -> row l % 16, k = l / 16; B: lane l -> column l % 16, k = l / 16;
// D: lane l -> column l % 16, rows 4 (l / 16) + {0..3}.
// Data with heavy cancellation (terms of magnitude 1e4 whose sum is ~1) so that any other summation order shows.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D, int K)
{
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[i * K + k0 + kk], b = B[(k0 + kk) * 16 + i];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * kk + r) * 16 + i] = acc[r];
}

int main()
{
    const int K = 24;
    float hA[16 * K], hB[K * 16], hD[256], ref[256];
    int bad_total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        srand(trial + 1);
        for (int i = 0; i < 16 * K; ++i) hA[i] = (float)((rand() % 20001) - 10000) * (1.0f / 7.0f) * ((trial & 1) ? 1.0f : 1e-3f);
        for (int i = 0; i < K * 16; ++i) hB[i] = (float)((rand() % 20001) - 10000) * (1.0f / 3.0f);
        if (trial % 3 == 0) for (int i = 0; i < 16; ++i) hA[i * K + K - 1] = 0.0f, hA[i * K] = 0.0f;     // zero padding at both ends
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = fmaf(hA[i * K + k], hB[k * 16 + j], acc);
                ref[i * 16 + j] = acc;
            }
        float *dA, *dB, *dD;
        hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += memcmp(&hD[i], &ref[i], 4) != 0;
        bad_total += bad;
        if (bad && trial < 4) printf("trial %d: %d of 256 differ, e.g. %.9g vs %.9g\n", trial, bad, hD[0], ref[0]);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    printf("v_mfma_f32_16x16x4_f32 vs k-ascending fmaf chain: %d mismatching elements in 200 trials of 256\n", bad_total);
    return 0;
}
