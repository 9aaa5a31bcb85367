// Probe of ds_read_b64_tr_b16 on gfx950: what does lane l receive when the 16 lanes of a group point at a
// [4 rows][16 columns] block of 16-bit elements (row pitch free)?   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int pitch)
{
    __shared__ __attribute__((aligned(16))) unsigned short t[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) t[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, a = l & 15, g = l >> 4;
    const unsigned short* p = t + (a >> 2) * pitch + 16 * (g & 1) + 4 * (a & 3) + (g >> 1) * 8 * pitch;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[4 * l + j] = (unsigned short)v[j];
}
int main()
{
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int pitch : {32, 76}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int a = l & 15, g = l >> 4;
                const int want = (j + (g >> 1) * 8) * pitch + 16 * (g & 1) + a;      // row j, column a of the group's block
                if (h[4 * l + j] != want) ok = 0;
            }
        printf("pitch %d: lane l gets [row j][column l&15] of its group's block: %s\n", pitch, ok ? "yes" : "NO");
        if (!ok) for (int l = 0; l < 20; ++l) printf("  lane %d: %d %d %d %d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
