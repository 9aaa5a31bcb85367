// down0 + down0_1 (and down1 + down1_1) of the TPS++ regressor, ResNet45v2 wiring, as ONE kernel on the bf16 matrix cores:
//
//     feat = bf16(relu(W0 in[p] + b0))                        1x1, 32 -> 64 at full resolution (32 x 128)
//     out  = bf16(relu(conv3x3, stride 2, pad 1 (feat) + bd)) 64 -> 64 at half resolution (16 x 64), blocked layout
//
// Reference: TPS_PP.forward, mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:560-563 (`self.down0_1(self.down0(outs[0]))`,
// `self.down1_1(self.down1(outs[1]))`).
//
// Why: as separate launches feat0 / feat1 (0.54 GB per 512 images) are written by the fused front (tpspp_front_bf16.hip,
// a third of its time) only to be read once, by the two stride-2 convolutions (which are bound by those reads).  The front
// still needs them as operands of its 192-deep product and keeps computing them in registers; here they are computed a
// second time (2 of 38 matrix instructions per output fragment), straight into the LDS patch the 3x3 product reads, and
// never exist in HBM: this kernel reads 0.13 GB instead of 0.27 GB and the front writes 0.33 GB instead of 0.87 GB.
//
// Same arithmetic as the unfused composition, bit for bit: the 1x1 product is the front's (two k-steps of
// v_mfma_f32_32x32x16_bf16 in channel order, + bias, ReLU, one rounding to bf16), the 3x3 product is
// conv_tiled_bf16_kernel's / conv3_blk_persist_kernel's (chunks of 16 channels ascending, taps row-major inside a chunk,
// one accumulator), zero padding applied to feat (not to `in`).
//
// Organisation (a workgroup = 4 wavefronts walks down an image, one output row of 64 pixels per step):
//   * the 3x3 weight never touches LDS: wavefront (f, h2) owns output pixels 32 f .. 32 f + 31 and output channels
//     32 h2 .. 32 h2 + 31 for the life of the workgroup, i.e. 36 A fragments = 144 registers loaded once;
//   * feat rows live in a ring of 3 LDS rows in the patch layout of the stride-2 kernels: [channel group][parity][pad |
//     64 columns] 16-byte units, so the 32 pixels of a fragment read 32 consecutive units for every tap and the left
//     padding column is a zero unit; a step adds input rows 2 oy and 2 oy + 1 (row 2 oy - 1 is the previous step's);
//   * producer: every wavefront takes one 32-pixel segment of each new row exactly as the front does -- 16-byte pieces of
//     the NCHW rows into a private [channel][32 pixels] tile, ds_read_b64_tr_b16 for the B operand, 4 matrix
//     instructions, the C/D registers paired into 16-byte units with v_permlane32_swap -- and writes the units into the
//     ring; the next step's pieces are already in flight (registers);
//   * two barriers per step (ring written / ring read); two workgroups per CU cover each other's;
//   * a step's two result units are stored one step late, in front of the next fetch (see the loop).
// Bound: HBM (0.13 GB in, 0.07 GB out per 512 images = 33 us at 6 TB/s); measured 62-68 us = 3.0-3.2 TB/s: the sum of a
// step's dependent phases (scripts/debug/trace_down_fused.py; per step and wavefront ~128 results x (add, max, half a
// conversion) + pairing in the producer = ~400 vector instructions, then the 36-deep chain of dependent products).
// Tried and dropped (round 4): the two phases on different wavefronts -- 4 consumers (the weight registers) + 4 producers
// (the 1x1 bias in registers) per workgroup, ring of 5 rows, LDS counters instead of barriers, pieces two steps ahead --
// bit-identical and no faster (64-72 us): the producers' vector instructions, not the phases' order, set the step time.
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct DownFParams {
    const unsigned short* in;       // (N, 32, H, W) bf16
    const u32x4* w0;                // [2 k-steps][2][64][8]
    const float* b0;
    const u32x4* wd;                // [4 chunks][9 taps][2][64][8]
    const float* bd;
    unsigned short* out;            // (N, 8, H / 2, 64, 8) bf16
    int N, H, W;                    // W == 128
    int rows_per_unit;              // output rows per work unit (divides H / 2)
    int relu;
};

// -DTPSPP_DOWNF_TRACE: workgroup 0 records s_memtime at its phase boundaries (wavefront w: stamps 64 w ..), read back with
// tpspp_debug_downf_trace (scripts/debug/trace_down_fused.py); compiled out of the product.
#ifdef TPSPP_DOWNF_TRACE
__device__ long long g_trace[4 * 64];
#define TRACE_INIT() long long* trp_ = g_trace + wv * 64; int tri_ = 0; const bool tr_ = blockIdx.x == 0 && lane == 0
#define STAMP() do { if (tr_ && tri_ < 64) trp_[tri_++] = __builtin_amdgcn_s_memtime(); } while (0)
// a stamp taken once `v` is available
#define STAMP2(v) do { asm volatile("s_nop 0" :: "v"(v)); STAMP(); } while (0)
#else
#define STAMP2(v) do {} while (0)
#define TRACE_INIT() do {} while (0)
#define STAMP() do {} while (0)
#endif

constexpr int kW = 128, kWo = 64;
constexpr int RP = 72;                           // units per (channel group, parity) run: pad | 64 columns | 7 unused
                                                 // (72: the even and the odd run of a producer's 16-lane write group are
                                                 // 32 banks apart)
constexpr int kRowUnits = 8 * 2 * RP;            // a feature row: 1152 units = 18 KB
constexpr int kRing = 3;
constexpr int kW0Units = 256;
constexpr int kTileBytes = 2048;                 // a wavefront's input tile [32 channels][32 pixels]
constexpr int kSmemBytes = kW0Units * 16 + 128 * 4 + 4 * kTileBytes + kRing * kRowUnits * 16;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u32x2 read_tr(const unsigned short* p)
{
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

__global__ void __launch_bounds__(256, 2)
down_fused_kernel(const DownFParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* const sW0 = reinterpret_cast<u32x4*>(smem);
    float* const sBias = reinterpret_cast<float*>(smem + kW0Units * 16);               // b0 | bd
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned short* const tile = reinterpret_cast<unsigned short*>(smem + kW0Units * 16 + 512 + wv * kTileBytes);
    u32x4* const ring = reinterpret_cast<u32x4*>(smem + kW0Units * 16 + 512 + 4 * kTileBytes);
    const int half = lane >> 5, l31 = lane & 31;

    TRACE_INIT();
    STAMP();
    sW0[tid] = P.w0[tid];
    if (tid < 128) sBias[tid] = tid < 64 ? P.b0[tid] : P.bd[tid - 64];
    if (tid < kRing * 16) {                                   // the padding column of every run, zero for good
        const u32x4 z = {0u, 0u, 0u, 0u};
        ring[(tid >> 4) * kRowUnits + (tid & 15) * RP] = z;
    }
    // this wavefront's 3x3 weight: output channels 32 h2 + l31, k group `half` of every (chunk, tap)
    const int f = wv >> 1, h2 = wv & 1;
    bf16x8 wa[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wa[c][t] = __builtin_bit_cast(bf16x8, P.wd[((c * 9 + t) * 2 + half) * 64 + 32 * h2 + l31]);
    __syncthreads();
    STAMP();

    const int H = P.H, Ho = H >> 1, plane = H * kW;
    const int RS = P.rows_per_unit, upi = Ho / RS, nunits = P.N * upi;
    const int lb = (((lane & 15) >> 2) + 8 * half) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int px = 32 * wv + l31;                             // the producer's pixel: segment = wavefront
    const int punit = (px & 1) * RP + 1 + (px >> 1);          // its unit inside a (channel group) pair of runs

    // 16-byte pieces of the wavefront's segment of input row iy: piece p = lane + 64 i -> channel p >> 2, pixels 8 (p & 3) ..
    auto fetch = [&](int n, int iy, u32x4 (&r)[2]) {
        const unsigned short* src = P.in + (size_t)n * 32 * plane + (size_t)iy * kW + 32 * wv;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = lane + 64 * i;
            r[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(p >> 2) * plane + (p & 3) * 8);
        }
    };
    // feat of the wavefront's segment of input row iy -> ring
    auto produce = [&](int iy, const u32x4 (&r)[2]) {
        reinterpret_cast<u32x4*>(tile)[lane] = r[0];
        reinterpret_cast<u32x4*>(tile)[lane + 64] = r[1];
        asm volatile("" ::: "memory");
        u32x4 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x2 k0 = read_tr(tile + lb + (16 * j) * 32), k1 = read_tr(tile + lb + (16 * j + 4) * 32);
            b[j][0] = k0[0]; b[j][1] = k0[1]; b[j][2] = k1[0]; b[j][3] = k1[1];
        }
        asm volatile("" ::: "memory");                        // ... before the next segment overwrites the tile
        STAMP2(b[1][3]);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, sW0[(2 * j + half) * 64 + l31]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, sW0[(2 * j + half) * 64 + 32 + l31]);
            const bf16x8 bb = __builtin_bit_cast(bf16x8, b[j]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1], 0, 0, 0);
        }
        STAMP2(acc[1][15]);
        u32x4* const row = ring + ((iy + 1) % kRing) * kRowUnits + punit;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // the bias goes on with packed adds and the ReLU is taken AFTER the rounding, on the packed pair (a negative
                // bf16 is a negative int16, so max(., 0) is the same ReLU; -0 and NaN -> +0 / NaN as before is not needed:
                // s > 0 ? s : 0 maps NaN to 0 and so does the signed max of a NaN pattern with the sign bit set only --
                // positive-NaN patterns cannot come out of finite operands): 6 instructions per 4 results instead of 10
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + 32 * t + 8 * g + 4 * half);
                f32x2 lo, hi, blo, bhi;
                lo[0] = acc[t][4 * g]; lo[1] = acc[t][4 * g + 1]; hi[0] = acc[t][4 * g + 2]; hi[1] = acc[t][4 * g + 3];
                blo[0] = b4.x; blo[1] = b4.y; bhi[0] = b4.z; bhi[1] = b4.w;
                lo = lo + blo; hi = hi + bhi;
                const s16x2 z = {0, 0};
                pk[g][0] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(lo[0], lo[1])), z));
                pk[g][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(hi[0], hi[1])), z));
            }
            // the two half-wavefronts hold the two 8-byte halves of every unit: after the swap the lower one owns the
            // unit of channel group 4 t + g, the upper one that of 4 t + g + 1
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const u32x2 d0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const u32x2 d1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                row[(4 * t + g + half) * (2 * RP)] = unit;
            }
        }
    };

    u32x4 pa[2], pb[2];                                       // the pieces of the next step's two rows
    u32x4 hold[2];                                            // the last step's two result units and where they go
    unsigned short* hold_p = nullptr;
    bool held = false;                                        // (uniform)
    int u = blockIdx.x;
    if (u < nunits) {
        const int n = u / upi, oy = (u - n * upi) * RS;
        fetch(n, 2 * oy, pa);
        fetch(n, 2 * oy + 1, pb);
    }
    for (; u < nunits; u += gridDim.x) {
        const int n = u / upi, oy_s = (u - n * upi) * RS;
        __syncthreads();                                      // the previous unit's last row has been read
        if (oy_s == 0) {
            // input row -1: zero padding (slot 0)
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = tid; i < kRowUnits; i += 256) ring[i] = z;
        } else {
            u32x4 pc[2];
            fetch(n, 2 * oy_s - 1, pc);
            produce(2 * oy_s - 1, pc);
        }
        for (int oy = oy_s; oy < oy_s + RS; ++oy) {
            if (oy > oy_s) __syncthreads();                   // the rows this step overwrites have been read
            STAMP();
            produce(2 * oy, pa);
            produce(2 * oy + 1, pb);
            STAMP();
            // The previous step's results leave HERE, in front of the next fetch: a wavefront's vector-memory operations
            // retire in issue order and the compiler waits for the fetched pieces with vmcnt(0) at the top of a step, so
            // stores issued BEHIND the fetch (at the end of the step that computed them) are drained there as well --
            // their write acknowledgements, 1100-2200 of a step's 6700 cycles (scripts/debug/trace_down_fused.py).
            // Waiting by count instead (s_waitcnt vmcnt(2): "all but the two stores") is not safe: with the machine full
            // (530 images) stores were seen to retire before older loads.
            if (held) {
                *reinterpret_cast<u32x4*>(hold_p) = hold[0];
                *reinterpret_cast<u32x4*>(hold_p + (size_t)2 * Ho * kWo * 8) = hold[1];
            }
            {
                // the next step's pieces: in flight under this step's products
                int nn = n, noy = oy + 1;
                bool more = true;
                if (noy == oy_s + RS) {
                    const int nu = u + (int)gridDim.x;
                    more = nu < nunits;
                    nn = nu / upi;
                    noy = (nu - nn * upi) * RS;
                }
                if (more) { fetch(nn, 2 * noy, pa); fetch(nn, 2 * noy + 1, pb); }
            }
            __syncthreads();                                  // the three rows of this step are in the ring
            STAMP();
            // ---- 3x3, stride 2: input rows 2 oy - 1 + ky in ring slots (2 oy + ky) % 3 ----
            const u32x4* rows[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rows[ky] = ring + ((2 * oy + ky) % kRing) * kRowUnits + half * (2 * RP) + 32 * f + l31;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            // B fragments are requested kPD products ahead (a register ring; the scheduling barriers keep the order: left
            // to itself the compiler issues every read right in front of its product -- 36 LDS latencies per step)
            constexpr int kPD = 3;
            bf16x8 fb[kPD + 1];
            auto fetch_b = [&](int i, int slot) {
                const int c = i / 9, t = i - 9 * c, ky = t / 3, kx = t - 3 * ky;
                // kx = 0: odd column 2 ox - 1 (run 1, slot ox); kx = 1: even column 2 ox (run 0, slot ox + 1);
                // kx = 2: odd column 2 ox + 1 (run 1, slot ox + 1)
                const int off = (2 * c) * (2 * RP) + (kx == 1 ? 0 : RP) + (kx == 0 ? 0 : 1);
                fb[slot] = __builtin_bit_cast(bf16x8, rows[ky][off]);
            };
#pragma unroll
            for (int i = 0; i < kPD; ++i) fetch_b(i, i);
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                if (i + kPD < 36) fetch_b(i + kPD, (i + kPD) % (kPD + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i / 9][i % 9], fb[i % (kPD + 1)], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP();
            // ---- bias, ReLU, rounding; 16-byte units of the blocked output ----
            u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = acc[4 * g + e] + sBias[64 + 32 * h2 + 8 * g + 4 * half + e];
                    v[e] = (P.relu && !(s > 0.0f)) ? 0.0f : s;
                }
                pk[g][0] = pack_bf16(v[0], v[1]); pk[g][1] = pack_bf16(v[2], v[3]);
            }
            hold_p = P.out + ((((size_t)n * 8 + 4 * h2 + half) * Ho + oy) * kWo + 32 * f + l31) * 8;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const u32x2 d0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const u32x2 d1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                hold[g >> 1] = unit;                          // channel group 4 h2 + g + half
            }
            held = true;
            STAMP();
        }
    }
    if (held) {
        *reinterpret_cast<u32x4*>(hold_p) = hold[0];
        *reinterpret_cast<u32x4*>(hold_p + (size_t)2 * Ho * kWo * 8) = hold[1];
    }
}



// ---- the same kernel for the three-term split ("bf16x3": fp32 tensors, products hi*hi + hi*lo + lo*hi) ----------------------
// in (N, 32, H, 128) fp32, out (N, 8, H/2, 64, 8) fp32 (the blocked fp32 layout of tpspp_conv2d_bf16_fwd, code 3).  Bit for bit
// front_x3_kernel's feat0 / feat1 followed by the three-term 3x3 stride-2 convolution: the producer is front_x3's arithmetic
// (fp32 inputs split in registers, six matrix instructions per k-step in its order, bias + ReLU in fp32, the fp32 result
// split into hi and lo -- never rounded as a whole), the ring holds a hi and a lo row image, the consumer adds
// Ah Bh, Ah Bl, Al Bh per (chunk, tap) in the convolution kernel's order.  The 36 hi and 36 lo weight fragments of a
// wavefront are 288 registers: one workgroup of four wavefronts per CU (512 registers each).
constexpr int kSmemX3 = 2 * kW0Units * 16 + 128 * 4 + 2 * kRing * kRowUnits * 16;

__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(v0, v1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(v0 - h0, v1 - h1);
}

struct DownXParams {
    const float* in;                // (N, 32, H, W) fp32
    const u32x4* w0;                // [hi|lo][2 k-steps][2][64][8]
    const float* b0;
    const u32x4* wd;                // [4 chunks][hi|lo][9 taps][2][64][8]
    const float* bd;
    float* out;                     // (N, 8, H / 2, 64, 8) fp32
    int N, H, W;
    int rows_per_unit;
    int relu;
};

__global__ void __launch_bounds__(256, 1)
down_fused_x3_kernel(const DownXParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* const sW0 = reinterpret_cast<u32x4*>(smem);                                   // hi slab | lo slab
    float* const sBias = reinterpret_cast<float*>(smem + 2 * kW0Units * 16);           // b0 | bd
    u32x4* const ringH = reinterpret_cast<u32x4*>(smem + 2 * kW0Units * 16 + 512);
    u32x4* const ringL = ringH + kRing * kRowUnits;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    for (int i = tid; i < 2 * kW0Units; i += 256) sW0[i] = P.w0[i];
    if (tid < 128) sBias[tid] = tid < 64 ? P.b0[tid] : P.bd[tid - 64];
    if (tid < kRing * 16) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        ringH[(tid >> 4) * kRowUnits + (tid & 15) * RP] = z;
        ringL[(tid >> 4) * kRowUnits + (tid & 15) * RP] = z;
    }
    const int f = wv >> 1, h2 = wv & 1;
    bf16x8 wah[4][9], wal[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wah[c][t] = __builtin_bit_cast(bf16x8, P.wd[(((c * 2 + 0) * 9 + t) * 2 + half) * 64 + 32 * h2 + l31]);
            wal[c][t] = __builtin_bit_cast(bf16x8, P.wd[(((c * 2 + 1) * 9 + t) * 2 + half) * 64 + 32 * h2 + l31]);
        }
    __syncthreads();

    const int H = P.H, Ho = H >> 1, plane = H * kW;
    const int RS = P.rows_per_unit, upi = Ho / RS, nunits = P.N * upi;
    const int px = 32 * wv + l31;
    const int punit = (px & 1) * RP + 1 + (px >> 1);
    const unsigned lo = (unsigned)(l31 + 8 * half * plane);

    // the 16 fp32 inputs of this lane for a segment: channels 16 j + 8 half + e of pixel 32 wv + l31 of row iy
    auto fetch = [&](int n, int iy, float (&v)[2][8]) {
        const float* base = P.in + (size_t)n * 32 * plane + (size_t)iy * kW + 32 * wv;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = (base + (size_t)(16 * j + e) * plane)[lo];
    };
    auto produce = [&](int iy, const float (&v)[2][8]) {
        u32x4 ih[2], il[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, l;
                split2(v[j][2 * q], v[j][2 * q + 1], h, l);
                ih[j][q] = h; il[j][q] = l;
            }
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        constexpr int LO = kW0Units;                          // 16-byte units between the hi and the lo slab
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, sW0[(2 * j + half) * 64 + l31]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, sW0[(2 * j + half) * 64 + 32 + l31]);
            const bf16x8 a0l = __builtin_bit_cast(bf16x8, sW0[LO + (2 * j + half) * 64 + l31]);
            const bf16x8 a1l = __builtin_bit_cast(bf16x8, sW0[LO + (2 * j + half) * 64 + 32 + l31]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, ih[j]), bl = __builtin_bit_cast(bf16x8, il[j]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, acc[1], 0, 0, 0);
        }
        const int ro = ((iy + 1) % kRing) * kRowUnits + punit;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x2 ph[4], pl[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sum = acc[t][4 * g + e] + sBias[32 * t + 8 * g + 4 * half + e];
                    r[e] = sum > 0.0f ? sum : 0.0f;
                }
                unsigned h01, l01, h23, l23;
                split2(r[0], r[1], h01, l01);
                split2(r[2], r[3], h23, l23);
                ph[g][0] = h01; ph[g][1] = h23; pl[g][0] = l01; pl[g][1] = l23;
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const u32x2 d0 = __builtin_amdgcn_permlane32_swap(ph[g][0], ph[g + 1][0], false, false);
                const u32x2 d1 = __builtin_amdgcn_permlane32_swap(ph[g][1], ph[g + 1][1], false, false);
                u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                ringH[ro + (4 * t + g + half) * (2 * RP)] = unit;
                const u32x2 e0 = __builtin_amdgcn_permlane32_swap(pl[g][0], pl[g + 1][0], false, false);
                const u32x2 e1 = __builtin_amdgcn_permlane32_swap(pl[g][1], pl[g + 1][1], false, false);
                u32x4 ul; ul[0] = e0[0]; ul[1] = e1[0]; ul[2] = e0[1]; ul[3] = e1[1];
                ringL[ro + (4 * t + g + half) * (2 * RP)] = ul;
            }
        }
    };

    float pa[2][8], pb[2][8];
    float4 hold[4];
    float* hold_p = nullptr;
    bool held = false;
    int u = blockIdx.x;
    if (u < nunits) {
        const int n = u / upi, oy = (u - n * upi) * RS;
        fetch(n, 2 * oy, pa);
        fetch(n, 2 * oy + 1, pb);
    }
    for (; u < nunits; u += gridDim.x) {
        const int n = u / upi, oy_s = (u - n * upi) * RS;
        __syncthreads();
        if (oy_s == 0) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = tid; i < kRowUnits; i += 256) { ringH[i] = z; ringL[i] = z; }
        } else {
            float pc[2][8];
            fetch(n, 2 * oy_s - 1, pc);
            produce(2 * oy_s - 1, pc);
        }
        for (int oy = oy_s; oy < oy_s + RS; ++oy) {
            if (oy > oy_s) __syncthreads();
            produce(2 * oy, pa);
            produce(2 * oy + 1, pb);
            if (held) {                                       // the previous step's results leave in front of the next fetch
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(hold_p + (size_t)g * Ho * kWo * 8) = hold[g];
            }
            {
                int nn = n, noy = oy + 1;
                bool more = true;
                if (noy == oy_s + RS) {
                    const int nu = u + (int)gridDim.x;
                    more = nu < nunits;
                    nn = nu / upi;
                    noy = (nu - nn * upi) * RS;
                }
                if (more) { fetch(nn, 2 * noy, pa); fetch(nn, 2 * noy + 1, pb); }
            }
            __syncthreads();
            const int rbase = half * (2 * RP) + 32 * f + l31;
            int rows[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rows[ky] = ((2 * oy + ky) % kRing) * kRowUnits + rbase;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            constexpr int kPD = 2;
            bf16x8 fbh[kPD + 1], fbl[kPD + 1];
            auto fetch_b = [&](int i, int slot) {
                const int c = i / 9, t = i - 9 * c, ky = t / 3, kx = t - 3 * ky;
                const int off = (2 * c) * (2 * RP) + (kx == 1 ? 0 : RP) + (kx == 0 ? 0 : 1);
                fbh[slot] = __builtin_bit_cast(bf16x8, ringH[rows[ky] + off]);
                fbl[slot] = __builtin_bit_cast(bf16x8, ringL[rows[ky] + off]);
            };
#pragma unroll
            for (int i = 0; i < kPD; ++i) fetch_b(i, i);
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                if (i + kPD < 36) fetch_b(i + kPD, (i + kPD) % (kPD + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wah[i / 9][i % 9], fbh[i % (kPD + 1)], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wah[i / 9][i % 9], fbl[i % (kPD + 1)], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wal[i / 9][i % 9], fbh[i % (kPD + 1)], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bias, ReLU; this lane's four channels of channel group 4 h2 + g are 16 bytes of that group's 32-byte unit
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sum = acc[4 * g + e] + sBias[64 + 32 * h2 + 8 * g + 4 * half + e];
                    r[e] = (P.relu && !(sum > 0.0f)) ? 0.0f : sum;
                }
                hold[g] = make_float4(r[0], r[1], r[2], r[3]);
            }
            hold_p = P.out + ((((size_t)n * 8 + 4 * h2) * Ho + oy) * kWo + 32 * f + l31) * 8 + 4 * half;
            held = true;
        }
    }
    if (held) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(hold_p + (size_t)g * Ho * kWo * 8) = hold[g];
    }
}


// ---- the same kernel for the exact-fp32 configuration (v_mfma_f32_32x32x2_f32) ------------------------------------------------
// in (N, 32, H, 128) fp32, out (N, 64, H/2, 64) fp32 NCHW.  Bit for bit front_kernel's feat0 / feat1 (tpspp_front.hip: channel
// pairs ascending, two accumulators, bias + ReLU) followed by conv_tiled_f32_kernel<3, 2, 2, ...> (tpspp_conv.hip: chunks of
// 4 channels ascending, taps row-major inside a chunk, the chunk's two channel pairs inside a tap, one accumulator).  A
// wavefront's weight is 16 chunks x 9 taps x 2 pairs = 288 one-register A fragments; the ring holds fp32 rows
// [channel][parity][pad | 64 columns] (a fragment's 32 pixels read 32 consecutive floats); one workgroup per CU.
constexpr int RPF = 80;                          // floats per (channel, parity) run (80: the even and the odd run of a
                                                 // producer's write are 16 banks apart)
constexpr int kRowF = 64 * 2 * RPF;              // floats per feature row
constexpr int kSmemF32 = (32 * 64 + 128 + kRing * kRowF) * 4;

__device__ __forceinline__ constexpr int feat_of(int ks, int half)
{
    return 32 * (ks >> 4) + (ks & 3) + 8 * ((ks & 15) >> 2) + 4 * half;      // the channel of result register ks (C/D layout)
}

struct DownFP32Params {
    const float* in;                // (N, 32, H, W)
    const float* w0;                // [32 k][64]
    const float* b0;
    const float* wd;                // [16 chunks][9 taps][4][64]
    const float* bd;
    float* out;                     // (N, 64, H / 2, 64)
    int N, H, W;
    int rows_per_unit;
    int relu;
};

__global__ void __launch_bounds__(256, 1)
down_fused_f32_kernel(const DownFP32Params P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const sW0 = reinterpret_cast<float*>(smem);
    float* const sBias = sW0 + 32 * 64;                                                 // b0 | bd
    float* const ring = sBias + 128;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    for (int i = tid; i < 32 * 64; i += 256) sW0[i] = P.w0[i];
    if (tid < 128) sBias[tid] = tid < 64 ? P.b0[tid] : P.bd[tid - 64];
    for (int i = tid; i < kRing * 128; i += 256) ring[(i >> 7) * kRowF + (i & 127) * RPF] = 0.0f;     // the padding columns
    const int f = wv >> 1, h2 = wv & 1;
    float wr[16][9][2];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) wr[c][t][c2] = P.wd[((c * 9 + t) * 4 + 2 * c2 + half) * 64 + 32 * h2 + l31];
    __syncthreads();

    const int H = P.H, Ho = H >> 1, plane = H * kW;
    const int RS = P.rows_per_unit, upi = Ho / RS, nunits = P.N * upi;
    const int px = 32 * wv + l31;
    const int pidx = (px & 1) * RPF + 1 + (px >> 1);

    // the lane's 16 inputs of a segment: channels 2 ks + half of pixel 32 wv + l31 of row iy
    auto fetch = [&](int n, int iy, float (&v)[16]) {
        const float* base = P.in + ((size_t)n * 32 + half) * plane + (size_t)iy * kW + px;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) v[ks] = base[(size_t)(2 * ks) * plane];
    };
    auto produce = [&](int iy, const float (&v)[16]) {
        f32x16 a0, a1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = 0.0f; a1[i] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW0[(2 * ks + half) * 64 + l31], v[ks], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW0[(2 * ks + half) * 64 + 32 + l31], v[ks], a1, 0, 0, 0);
        }
        float* const row = ring + ((iy + 1) % kRing) * kRowF + pidx;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const float r = (ks < 16 ? a0[ks & 15] : a1[ks & 15]) + sBias[feat_of(ks, half)];
            row[feat_of(ks, half) * (2 * RPF)] = r > 0.0f ? r : 0.0f;
        }
    };

    float pa[16], pb[16];
    int u = blockIdx.x;
    if (u < nunits) {
        const int n = u / upi, oy = (u - n * upi) * RS;
        fetch(n, 2 * oy, pa);
        fetch(n, 2 * oy + 1, pb);
    }
    for (; u < nunits; u += gridDim.x) {
        const int n = u / upi, oy_s = (u - n * upi) * RS;
        __syncthreads();
        if (oy_s == 0) {
            for (int i = tid; i < kRowF; i += 256) ring[i] = 0.0f;
        } else {
            float pc[16];
            fetch(n, 2 * oy_s - 1, pc);
            produce(2 * oy_s - 1, pc);
        }
        for (int oy = oy_s; oy < oy_s + RS; ++oy) {
            if (oy > oy_s) __syncthreads();
            produce(2 * oy, pa);
            produce(2 * oy + 1, pb);
            {
                int nn = n, noy = oy + 1;
                bool more = true;
                if (noy == oy_s + RS) {
                    const int nu = u + (int)gridDim.x;
                    more = nu < nunits;
                    nn = nu / upi;
                    noy = (nu - nn * upi) * RS;
                }
                if (more) { fetch(nn, 2 * noy, pa); fetch(nn, 2 * noy + 1, pb); }
            }
            __syncthreads();
            const int rbase = half * (2 * RPF) + 32 * f + l31;
            int rows[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rows[ky] = ((2 * oy + ky) % kRing) * kRowF + rbase;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            // (chunk c of 4 channels, tap t, pair c2): channel 4 c + 2 c2 + half at the tap's column
            constexpr int kPD = 6;
            float fb[kPD + 1];
            auto fetch_b = [&](int i, int slot) {
                const int c = i / 18, r = i - 18 * c, t = r >> 1, c2 = r & 1, ky = t / 3, kx = t - 3 * ky;
                const int off = (4 * c + 2 * c2) * (2 * RPF) + (kx == 1 ? 0 : RPF) + (kx == 0 ? 0 : 1);
                fb[slot] = ring[rows[ky] + off];
            };
#pragma unroll
            for (int i = 0; i < kPD; ++i) fetch_b(i, i);
#pragma clang loop unroll(full)
            for (int i = 0; i < 288; ++i) {
                if (i + kPD < 288) fetch_b(i + kPD, (i + kPD) % (kPD + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[i / 18][(i % 18) >> 1][i & 1], fb[i % (kPD + 1)], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            float* const ob = P.out + (((size_t)n * 64 + 32 * h2 + 4 * half) * Ho + oy) * kWo + 32 * f + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 8 * (r >> 2) + (r & 3);          // + 32 h2 + 4 half
                const float sum = acc[r] + sBias[64 + 32 * h2 + 4 * half + co];
                ob[(size_t)co * Ho * kWo] = (P.relu && !(sum > 0.0f)) ? 0.0f : sum;
            }
        }
    }
}

}  // namespace

#ifdef TPSPP_DOWNF_TRACE
extern "C" __attribute__((visibility("default"))) int tpspp_debug_downf_trace(long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(long long) * n);
}
#endif

TPSPP_EXPORT int tpspp_down_fused_bf16_fwd(const void* in, const void* w0, const float* b0, const void* wd, const float* bd,
                                           void* out, int N, int H, int W, int relu, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && w0 && b0 && wd && bd && out, "tpspp_down_fused_bf16_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && H > 0 && (H % 2) == 0 && W == kW,
                  "tpspp_down_fused_bf16_fwd: needs an even height and a width of 128 (got %d x %d)", H, W);
    TPSPP_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(w0) |
                    reinterpret_cast<size_t>(wd)) & 15) == 0, "tpspp_down_fused_bf16_fwd: tensors must be 16-byte aligned");
    if (N == 0) return TPSPP_OK;
    TPSPP_REQUIRE((long)N * (H / 2) < 0x7fffffffL, "tpspp_down_fused_bf16_fwd: batch too large");
    DownFParams P;
    P.in = static_cast<const unsigned short*>(in);
    P.w0 = static_cast<const u32x4*>(w0); P.b0 = b0;
    P.wd = static_cast<const u32x4*>(wd); P.bd = bd;
    P.out = static_cast<unsigned short*>(out);
    P.N = N; P.H = H; P.W = W; P.relu = relu ? 1 : 0;
    // work units: whole images when there are enough of them for two workgroups per CU, else strips of output rows (a
    // strip below the top of an image computes the feature row above it once more)
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        ncu = 256;
    }
    const int Ho = H / 2, slots = 2 * ncu;
    int rs = Ho;
    while (rs > 1 && (rs % 2) == 0 && (long)N * (Ho / rs) < slots) rs /= 2;
    P.rows_per_unit = rs;
    const long nunits = (long)N * (Ho / rs);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&down_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kSmemBytes);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(down_fused_kernel, dim3((unsigned)(nunits < slots ? nunits : slots)), dim3(256), kSmemBytes,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_down_fused_bf16_fwd");
}

TPSPP_EXPORT int tpspp_down_fused_x3_fwd(const float* in, const void* w0, const float* b0, const void* wd, const float* bd,
                                         float* out, int N, int H, int W, int relu, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && w0 && b0 && wd && bd && out, "tpspp_down_fused_x3_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && H > 0 && (H % 2) == 0 && W == kW,
                  "tpspp_down_fused_x3_fwd: needs an even height and a width of 128 (got %d x %d)", H, W);
    TPSPP_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(w0) |
                    reinterpret_cast<size_t>(wd)) & 15) == 0, "tpspp_down_fused_x3_fwd: tensors must be 16-byte aligned");
    if (N == 0) return TPSPP_OK;
    TPSPP_REQUIRE((long)N * (H / 2) < 0x7fffffffL, "tpspp_down_fused_x3_fwd: batch too large");
    DownXParams P;
    P.in = in; P.w0 = static_cast<const u32x4*>(w0); P.b0 = b0; P.wd = static_cast<const u32x4*>(wd); P.bd = bd; P.out = out;
    P.N = N; P.H = H; P.W = W; P.relu = relu ? 1 : 0;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        ncu = 256;
    }
    const int Ho = H / 2, slots = ncu;                      // one workgroup per CU
    int rs = Ho;
    while (rs > 1 && (rs % 2) == 0 && (long)N * (Ho / rs) < 2 * slots) rs /= 2;
    P.rows_per_unit = rs;
    const long nunits = (long)N * (Ho / rs);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&down_fused_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kSmemX3);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(down_fused_x3_kernel, dim3((unsigned)(nunits < slots ? nunits : slots)), dim3(256), kSmemX3,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_down_fused_x3_fwd");
}

TPSPP_EXPORT int tpspp_down_fused_f32_fwd(const float* in, const float* w0_slab, const float* b0, const float* wd_tiled,
                                          const float* bd, float* out, int N, int H, int W, int relu, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(in && w0_slab && b0 && wd_tiled && bd && out, "tpspp_down_fused_f32_fwd: null pointer");
    TPSPP_REQUIRE(N >= 0 && H > 0 && (H % 2) == 0 && W == kW,
                  "tpspp_down_fused_f32_fwd: needs an even height and a width of 128 (got %d x %d)", H, W);
    if (N == 0) return TPSPP_OK;
    TPSPP_REQUIRE((long)N * (H / 2) < 0x7fffffffL, "tpspp_down_fused_f32_fwd: batch too large");
    DownFP32Params P;
    P.in = in; P.w0 = w0_slab; P.b0 = b0; P.wd = wd_tiled; P.bd = bd; P.out = out;
    P.N = N; P.H = H; P.W = W; P.relu = relu ? 1 : 0;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        ncu = 256;
    }
    const int Ho = H / 2, slots = ncu;                      // one workgroup per CU
    int rs = Ho;
    while (rs > 1 && (rs % 2) == 0 && (long)N * (Ho / rs) < 2 * slots) rs /= 2;
    P.rows_per_unit = rs;
    const long nunits = (long)N * (Ho / rs);
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&down_fused_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kSmemF32);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(down_fused_f32_kernel, dim3((unsigned)(nunits < slots ? nunits : slots)), dim3(256), kSmemF32,
                       tpspp::as_stream(stream), P);
    return tpspp::check_launch("tpspp_down_fused_f32_fwd");
}
