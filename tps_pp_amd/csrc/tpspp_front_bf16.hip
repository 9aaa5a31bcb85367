// The pointwise front of the TPS++ regressor (ResNet45v2 wiring) in ONE kernel on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation), register-chained: the bf16 twin of tpspp_front.hip.
//
//     feat0 = relu(W0 outs0[p] + b0)                       1x1, 32 -> 64, full resolution (32x128)
//     feat1 = relu(W1 outs1[p] + b1)                       1x1, 32 -> 64
//     feat2 = relu(W2 x[p/2] + b2)                         1x1, 64 -> 64, half resolution (16x64)
//     feat_grid = relu(Wg [feat0; feat1; feat2] + bg)      1x1, 192 -> 64  (cat + nearest Upsample)
//
// As four separate launches these layers are HBM-bound at 2-3 TB/s (a single K-chunk each, feat0 / feat1 /
// feat2 written and read back: 4.1 MB per image); fused, an image moves 0.66 MB in and 1.7 MB out.
//
// A wavefront owns 32 consecutive full-resolution pixels of a row; a lane is (pixel l31, k-half h).  The B
// operand of a 32x32x16 step is 8 consecutive k per lane: for the three small GEMMs those are 8 input channels
// of the lane's pixel (read out of the wavefront's LDS tile, see "data movement").  Their results come back in the C/D layout --
// lane (pixel, h) holds output channels 8g + 4h + {0..3} -- and two consecutive groups g are exactly the 8 values
// the lane must supply as B operand of one k-step of the 192-deep GEMM, provided its weight slab is permuted on
// the host to that k-slot order (slot 8h+e of a 16-channel group = channel [0,1,2,3,8,9,10,11,4,5,6,7,12,..,15][8h+e]).
// So feat0 / feat1 / feat2 are rounded to bf16 once (the same rounding their stores to HBM use: the unfused
// composition computes feat_grid from exactly these values), packed with v_cvt_pk_bf16_f32 and fed straight
// back; activations cross no barrier.  Weight fragments are 16-B LDS reads of host-arranged
// [k-step][h][64 cout][8 k] slabs (40 KB in all, loaded once per persistent workgroup).
//
// Reference: TPS_PP.forward / TPS_PP.grid, mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:560-562,580-585.
// Bound: HBM (2.4 MB per image at 17 kMAC per pixel).
#include "tpspp_common.h"

namespace {

constexpr int kWave = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FrontBParams {
    const unsigned short* o0; const unsigned short* o1;   // (N, 32, H, W) bf16
    const unsigned short* x;                              // (N, 64, H/2, W/2) bf16
    const u32x4* w0; const u32x4* w1;                     // [2 k-steps][2][64][8] bf16
    const u32x4* w2;                                      // [4][2][64][8]
    const u32x4* wg;                                      // [12][2][64][8], k-slots in chain order
    const float* b0; const float* b1; const float* b2; const float* bg;
    unsigned short* feat0; unsigned short* feat1;         // (N, 64, H, W) bf16
    unsigned short* feat2;                                // (N, 64, H/2, W/2) bf16
    void* feat_grid;                                      // (N, 64, H, W) bf16 or fp32
    int fg_f32;
    int blk;                                              // feat0 / feat1 / feat2 in the blocked layout (N, 8, H, W, 8)
    int N, H, W;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));

// ---- data movement -------------------------------------------------------------------------------------------------
// A lane's MFMA operands are 2-byte elements of 16 (32) different channel planes, and its results are 2-byte elements
// of 64 planes.  Moved as such -- 64 loads and 128 stores of 64 B per half-wavefront and segment, or the same number
// of 2-byte LDS accesses behind wide loads and stores -- the kernel sits at 3.2 TB/s whatever the memory does (it
// takes 353 us with every load and store removed).  gfx950 has the instruction for this: ds_read_b64_tr_b16 hands
// lane (l & 15) of a 16-lane group column l & 15 of a [4 rows][16 columns] block of 16-bit elements whose rows the
// lanes point at (scripts/ubench/tr_probe.hip).  So
//   * inputs arrive in 16-byte pieces (8 pixels of a channel) and are laid down as they come, [channel][32 pixels];
//     a transposing read of 4 channel rows gives a lane 4 consecutive k of ITS pixel: 2 reads per k-step instead of 8
//     (the half-resolution input is doubled along x on its way into the tile, so it reads the same way);
//   * results are written [pixel][64 channels] (8 bytes per lane and group: 4 consecutive channels), and the
//     transposing read of 4 pixel rows gives a lane 4 consecutive pixels of one channel: 2 reads per 16-byte piece
//     of the NCHW output rows.
// 84 LDS and 20 global instructions per lane and segment instead of 312 + 20 (or 192 global ones).  Every wavefront
// owns its 8 KB tile: no barrier, a wavefront's LDS operations execute in order.  Pitches: 64 B for the input rows
// (4 rows -> 4 disjoint 8-bank ranges of the 64 banks), 152 B for the output rows (writes: 16 lanes x 8 B on 32
// distinct banks; reads: rows at 0 / 38 / 12 / 50 dwords mod 64).
constexpr int kTileBytes = 8192;            // inputs: outs0 [32][32] | outs1 [32][32] | x doubled [64][32]; then the outputs
constexpr int kOutPitch = 76;               // 16-bit elements per pixel row of the output tile

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2 read_tr(const unsigned short* p)
{
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// one input tile of 2 KB in global memory: 128 pieces of 16 bytes, two per lane; a channel's row segment holds
// 1 << QB pieces.  The pieces of the NEXT segment are fetched into registers while this one is computed.
// The loads are inline asm so that the wait for them stands where the kernel wants it (wait_fetched, at the END of the
// segment that was computed under them) instead of where the compiler would put it.
// Rounds 2-3 waited by COUNT there -- s_waitcnt vmcnt(n) with n = the stores issued since the fetch, "all but the
// segment's stores have completed" -- on the premise that a wavefront's loads and stores retire in issue order.  They do
// not: round 4's variant without the feat0 / feat1 stores (n = 4-8 instead of 12-20) read stale pieces with every CU busy
// (test_front_bf16_full_machine_matches_chunked_runs: 530 images against chunked runs) -- stores were acknowledged
// before older loads had returned, which lets the count drop below n early.  The wait is vmcnt(0) in every variant now;
// same box, same run: 281-285 us against 308-330 us counted (blocked bf16, 512 images), i.e. nothing lost.
template <int QB>
__device__ __forceinline__ void fetch_tile(const unsigned short* __restrict__ src, int plane, int lane, u32x4 (&r)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = lane + 64 * i;
        const unsigned short* g = src + (size_t)(p >> QB) * plane + (p & ((1 << QB) - 1)) * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(g) : "memory");
    }
}

// every vector-memory operation of this wavefront has completed (NSTORES = 0; a count > 0 is not a safe way to skip
// younger stores, see above).  The registers are operands so that their uses stay behind the wait.
template <int NSTORES>
__device__ __forceinline__ void wait_fetched(u32x4 (&a)[2], u32x4 (&b)[2], u32x4 (&c)[2])
{
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]) : "n"(NSTORES) : "memory");
}

__device__ __forceinline__ void put_tile(unsigned short* lds, const u32x4 (&r)[2], int lane)
{
    reinterpret_cast<u32x4*>(lds)[lane] = r[0];
    reinterpret_cast<u32x4*>(lds)[lane + 64] = r[1];
}

// the half-resolution tile, every pixel twice: piece p (channel p >> 1, pixels 8 (p & 1) ..) -> 32 bytes of row p >> 1
__device__ __forceinline__ void put_tile_doubled(unsigned short* lds, const u32x4 (&r)[2], int lane)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        u32x4 lo, hi;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            lo[2 * d] = __builtin_amdgcn_perm(r[i][d], r[i][d], 0x01000100u);
            lo[2 * d + 1] = __builtin_amdgcn_perm(r[i][d], r[i][d], 0x03020302u);
            hi[2 * d] = __builtin_amdgcn_perm(r[i][2 + d], r[i][2 + d], 0x01000100u);
            hi[2 * d + 1] = __builtin_amdgcn_perm(r[i][2 + d], r[i][2 + d], 0x03020302u);
        }
        const int p = lane + 64 * i;
        reinterpret_cast<u32x4*>(lds)[2 * p] = lo;
        reinterpret_cast<u32x4*>(lds)[2 * p + 1] = hi;
    }
}

// B operand of NK k-steps out of an input tile [channel][32 pixels]: channels 16j + 8h + e of the lane's pixel.
// `lb`: the lane's element offset ((a >> 2) + 8 half) * 32 + 16 * pixel-block + 4 (a & 3), a = lane & 15.
template <int NK>
__device__ __forceinline__ void load_b(const unsigned short* tile, int lb, u32x4 (&b)[NK])
{
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const u32x2 k0 = read_tr(tile + lb + (16 * j) * 32), k1 = read_tr(tile + lb + (16 * j + 4) * 32);
        b[j][0] = k0[0]; b[j][1] = k0[1]; b[j][2] = k1[0]; b[j][3] = k1[1];
    }
}

// out[t] = slab^T in  over NK k-steps, both 32-channel tiles
template <int NK>
__device__ __forceinline__ void gemm(const u32x4* __restrict__ slab, const u32x4 (&in)[NK], int half, int l31,
                                     f32x16 (&acc)[2])
{
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bb = __builtin_bit_cast(bf16x8, in[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1], 0, 0, 0);
    }
}

// bias + ReLU + one rounding to bf16; returns the 64 features as chain-ordered B fragments (4 k-steps) and, when
// `st`, writes them into row `px` of the wavefront's output tile [pixel][64 channels]
// `gdst` (blocked output, or null): channel group 0's 16-byte unit of this lane's pixel; the groups are `gstride`
// elements apart.  The two half-wavefronts hold the two 8-byte halves of every unit: v_permlane32_swap pairs them up so
// that the lower half-wavefront owns the whole unit of group g and the upper one that of group g + 1 -- four 16-byte
// stores per 64 channels, each two 512-byte runs, and no output tile / transposition
__device__ __forceinline__ void finish(const f32x16 (&acc)[2], const float* __restrict__ bias, int half,
                                       unsigned short* tile, int px, bool st, u32x4* __restrict__ out,
                                       unsigned short* gdst = nullptr, size_t gstride = 0)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (round 4) the bias goes on with packed adds and the ReLU is taken AFTER the rounding, on the packed pair: a
            // negative bf16 is a negative int16, so max(., 0) is the same ReLU (-0 -> +0 either way); 6 vector
            // instructions per 4 results instead of 10, identical bits -- this kernel's vector ALU is its busiest unit
            // (4 x 64 results per lane and segment against 40 matrix instructions)
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const float4 b4 = *reinterpret_cast<const float4*>(bias + 32 * t + 8 * g + 4 * half);
            f32x2 lo, hi, blo, bhi;
            lo[0] = acc[t][4 * g]; lo[1] = acc[t][4 * g + 1]; hi[0] = acc[t][4 * g + 2]; hi[1] = acc[t][4 * g + 3];
            blo[0] = b4.x; blo[1] = b4.y; bhi[0] = b4.z; bhi[1] = b4.w;
            lo = lo + blo; hi = hi + bhi;
            const s16x2 z = {0, 0};
            pk[g][0] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(lo[0], lo[1])), z));
            pk[g][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(hi[0], hi[1])), z));
            // group g of tile t is k-step 2t + g/2, slots 4(g&1) .. 4(g&1)+3
            out[2 * t + (g >> 1)][2 * (g & 1)] = pk[g][0];
            out[2 * t + (g >> 1)][2 * (g & 1) + 1] = pk[g][1];
            if (!gdst && st) *reinterpret_cast<u32x2*>(tile + px * kOutPitch + 32 * t + 8 * g + 4 * half) = pk[g];
        }
        if (gdst) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const u32x2 d0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const u32x2 d1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                u32x4 unit; unit[0] = d0[0]; unit[1] = d1[0]; unit[2] = d0[1]; unit[3] = d1[1];
                if (st) *reinterpret_cast<u32x4*>(gdst + (size_t)(4 * t + g + half) * gstride) = unit;
            }
        }
    }
}

// the output tile [NPX pixels][64 channels] -> 16-byte pieces (8 pixels) of the 64 channel planes
template <int NPX>
__device__ __forceinline__ void flush_tile(const unsigned short* tile, char* __restrict__ dst, size_t plane_bytes, int lane)
{
    asm volatile("" ::: "memory");                     // the writes above are not reordered behind these reads
    constexpr int NI = NPX / 8;                        // pieces per lane: 64 channels x NPX/8 pieces / 64 lanes
    const int a = lane & 15, g4 = lane >> 4;
    u32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        // NPX = 32: the four groups take the four pixel blocks of channels 16 i ..; NPX = 16: two blocks x two channel sets
        const int q = NPX == 32 ? g4 : (g4 & 1);
        const int ch0 = NPX == 32 ? 16 * i : 32 * i + 16 * (g4 >> 1);
        const unsigned short* p = tile + (8 * q + (a >> 2)) * kOutPitch + ch0 + 4 * (a & 3);
        const u32x2 lo = read_tr(p), hi = read_tr(p + 4 * kOutPitch);
        v[i][0] = lo[0]; v[i][1] = lo[1]; v[i][2] = hi[0]; v[i][3] = hi[1];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = NPX == 32 ? g4 : (g4 & 1);
        const int ch0 = NPX == 32 ? 16 * i : 32 * i + 16 * (g4 >> 1);
        *reinterpret_cast<u32x4*>(dst + (size_t)(ch0 + a) * plane_bytes + q * 16) = v[i];
    }
    asm volatile("" ::: "memory");
}

// Persistent workgroups of NW wavefronts, one workgroup per CU: the 40 KB of weight slabs are read into LDS once --
// from global memory they cycle through a 32 KB L1 that keeps none of them, 2.7 GB of L2 traffic per 512 images,
// more than the tensors themselves -- and every wavefront walks its own row segments.
constexpr int kSlabUnits = (2 + 2 + 4 + 12) * 128;      // 16-byte units: w0 | w1 | w2 | wg
constexpr int kSlabBytes = kSlabUnits * 16 + 4 * 64 * 4;  // + the four bias vectors

// ST01: feat0 / feat1 are stored (false: operands of feat_grid only -- tpspp_down_fused_bf16_fwd recomputes them where
// they are consumed; blocked form only)
template <bool FG32, int NW, bool BLK, bool ST01 = true>
__global__ void __launch_bounds__(NW * 64)
front_bf16_kernel(const FrontBParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* sW = reinterpret_cast<u32x4*>(smem);
    float* sBias = reinterpret_cast<float*>(smem + kSlabUnits * 16);
    for (int i = threadIdx.x; i < kSlabUnits; i += NW * 64) {
        const u32x4* src = i < 256 ? P.w0 + i : i < 512 ? P.w1 + (i - 256) : i < 1024 ? P.w2 + (i - 512) : P.wg + (i - 1024);
        sW[i] = *src;
    }
    for (int i = threadIdx.x; i < 256; i += NW * 64)
        sBias[i] = (i < 64 ? P.b0 : i < 128 ? P.b1 : i < 192 ? P.b2 : P.bg)[i & 63];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int segs_per_row = P.W >> 5;
    const long nseg = (long)P.N * P.H * segs_per_row;
    const int plane = P.H * P.W, W2 = P.W >> 1, plane2 = (P.H >> 1) * W2;
    const size_t pb = (size_t)plane * 2;
    unsigned short* t0 = reinterpret_cast<unsigned short*>(smem + kSlabBytes + (size_t)wv * kTileBytes);
    unsigned short* t1 = t0 + 1024;
    unsigned short* t2 = t1 + 1024;
    unsigned short* to = t0;                                          // the output tile takes the inputs' place
    const int lb = (((lane & 15) >> 2) + 8 * half) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    u32x4 pf0[2], pf1[2], pf2[2];
    auto fetch = [&](long sg) {
        const int fx = (int)(sg % segs_per_row);
        const long frow = sg / segs_per_row;
        const int fy = (int)(frow % P.H);
        const size_t fn = (size_t)(frow / P.H);
        fetch_tile<2>(P.o0 + fn * 32 * plane + (size_t)fy * P.W + fx * 32, plane, lane, pf0);
        fetch_tile<2>(P.o1 + fn * 32 * plane + (size_t)fy * P.W + fx * 32, plane, lane, pf1);
        fetch_tile<1>(P.x + fn * 64 * plane2 + (size_t)(fy >> 1) * W2 + fx * 16, plane2, lane, pf2);
    };
    const long step = (long)gridDim.x * NW;
    const long first = (long)blockIdx.x * NW + wv;
    if (first < nseg) { fetch(first); wait_fetched<0>(pf0, pf1, pf2); }
    for (long seg = first; seg < nseg; seg += step) {
        const int sx = (int)(seg % segs_per_row);
        const long row = seg / segs_per_row;
        const int y = (int)(row % P.H);
        const int n = (int)(row / P.H);
        const size_t seg0 = (size_t)y * P.W + sx * 32;                // first pixel of the segment (uniform)
        const size_t seg2 = (size_t)(y >> 1) * W2 + sx * 16;
        // the slabs do not change from segment to segment; the optimiser must not see that (it would hoist 160
        // registers' worth of LDS reads out of this loop and spill them)
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        const u32x4* w0 = sW + opaque;
        const u32x4* w1 = w0 + 256;
        const u32x4* w2 = w0 + 512;
        const u32x4* wg = w0 + 1024;
        const float* bias = sBias + opaque;

        put_tile(t0, pf0, lane);
        put_tile(t1, pf1, lane);
        put_tile_doubled(t2, pf2, lane);
        asm volatile("" ::: "memory");
        const bool more = seg + step < nseg;                          // uniform
        if (more) fetch(seg + step);                                  // in flight under this segment's arithmetic
        u32x4 in0[2], in1[2], in2[4];
        load_b<2>(t0, lb, in0);
        load_b<2>(t1, lb, in1);
        load_b<4>(t2, lb, in2);
        asm volatile("" ::: "memory");                                // ... before the first output overwrites them

        u32x4 f[12];                               // [feat0 | feat1 | feat2] as B fragments of the 192-deep GEMM
        f32x16 acc[2];
        auto zero = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        };
        const size_t obase = (size_t)n * 64 * plane + seg0;           // uniform
        zero();
        gemm<2>(w0, in0, half, l31, acc);
        if constexpr (BLK) {
            // blocked feat0 / feat1 / feat2: 8-byte pieces straight from the result registers
            const size_t bbase = ((size_t)n * 8 * plane + seg0 + l31) * 8;
            // ST01 = false: the stores are skipped by a run-time (uniform) test of the pointer, not compiled out -- as a
            // compile-time variant the kernel spills (260 bytes of scratch per lane at 168 registers, 455 us instead of 250)
            const bool st01 = ST01 || P.feat0 != nullptr;
            finish(acc, bias, half, to, l31, st01, f, P.feat0 + bbase, (size_t)plane * 8);
            zero();
            gemm<2>(w1, in1, half, l31, acc);
            finish(acc, bias + 64, half, to, l31, st01, f + 4, P.feat1 + bbase, (size_t)plane * 8);
        } else {
            finish(acc, bias, half, to, l31, true, f);
            flush_tile<32>(to, reinterpret_cast<char*>(P.feat0 + obase), pb, lane);
            zero();
            gemm<2>(w1, in1, half, l31, acc);
            finish(acc, bias + 64, half, to, l31, true, f + 4);
            flush_tile<32>(to, reinterpret_cast<char*>(P.feat1 + obase), pb, lane);
        }
        zero();
        gemm<4>(w2, in2, half, l31, acc);
        // one lane of every 2x2 block holds the half-resolution pixel: even rows write it, from the even-pixel lanes
        const bool row2 = (y & 1) == 0;                               // uniform
        if constexpr (BLK) {
            finish(acc, bias + 128, half, to, l31 >> 1, row2 && (l31 & 1) == 0, f + 8,
                   P.feat2 + ((size_t)n * 8 * plane2 + seg2 + (l31 >> 1)) * 8, (size_t)plane2 * 8);
        } else {
            finish(acc, bias + 128, half, to, l31 >> 1, row2 && (l31 & 1) == 0, f + 8);
            if (row2) flush_tile<16>(to, reinterpret_cast<char*>(P.feat2 + (size_t)n * 64 * plane2 + seg2), (size_t)plane2 * 2, lane);
        }
        zero();
        gemm<12>(wg, f, half, l31, acc);
        if constexpr (FG32) {
            // fp32 feat_grid: [channel][32 pixels] fp32, 4-byte writes (consecutive lanes, consecutive banks), 16-byte reads
            float* tf = reinterpret_cast<float*>(to);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = acc[t][r] + bias[192 + c];
                    tf[c * 32 + l31] = v > 0.0f ? v : 0.0f;
                }
            asm volatile("" ::: "memory");
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const u32x4*>(tf)[lane + 64 * i];
            float* gdst = reinterpret_cast<float*>(P.feat_grid) + obase;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = lane + 64 * i;
                *reinterpret_cast<u32x4*>(gdst + (size_t)(p >> 3) * plane + (p & 7) * 4) = v[i];
            }
            asm volatile("" ::: "memory");
        } else {
            u32x4 unused[4];
            finish(acc, bias + 192, half, to, l31, true, unused);
            flush_tile<32>(to, reinterpret_cast<char*>(reinterpret_cast<unsigned short*>(P.feat_grid) + obase), pb, lane);
        }
        if (more) wait_fetched<0>(pf0, pf1, pf2);
    }
}

// ---- the same kernel for fp32 tensors with the three-term bf16 split ("bf16x3", tpspp_conv_bf16.hip) ---------------
// Inputs, feat0 / feat1 / feat2 and feat_grid are fp32 in memory; every operand is split hi = bf16(v), lo = bf16(v - hi)
// in registers and a product is hi*hi + hi*lo + lo*hi.  feat0 / feat1 / feat2 are chained WITHOUT an intermediate
// rounding (their fp32 values are split, exactly what the separate convolutions would do with the stored tensors).
// Slabs: [hi|lo][k-steps][2][64][8].
struct FrontXParams {
    const float* o0; const float* o1; const float* x;
    const u32x4* w0; const u32x4* w1; const u32x4* w2; const u32x4* wg;
    const float* b0; const float* b1; const float* b2; const float* bg;
    float* feat0; float* feat1; float* feat2; float* feat_grid;
    int N, H, W;
    int blk;                     // feat0 / feat1 / feat2 in the fp32 blocked layout (N, 8, H, W, 8) (tpspp_conv2d_bf16_fwd code 3)
};

__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(v0, v1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(v0 - h0, v1 - h1);
}

template <int NK>
__device__ __forceinline__ void load_b3(const float* __restrict__ base, unsigned lo, int plane, u32x4 (&bh)[NK], u32x4 (&bl)[NK])
{
    float v[NK][8];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = (base + (size_t)(16 * j + e) * plane)[lo];      // uniform base + lane offset
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2(v[j][2 * q], v[j][2 * q + 1], h, l);
            bh[j][q] = h; bl[j][q] = l;
        }
}

template <int NK>
__device__ __forceinline__ void gemm3(const u32x4* __restrict__ slab, const u32x4* __restrict__ inh, const u32x4* __restrict__ inl,
                                      int half, int l31, f32x16 (&acc)[2])
{
    constexpr int LO = NK * 128;                       // 16-B units between the hi and the lo slab
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + l31]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, slab[(2 * j + half) * 64 + 32 + l31]);
        const bf16x8 a0l = __builtin_bit_cast(bf16x8, slab[LO + (2 * j + half) * 64 + l31]);
        const bf16x8 a1l = __builtin_bit_cast(bf16x8, slab[LO + (2 * j + half) * 64 + 32 + l31]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, inh[j]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, inl[j]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, acc[1], 0, 0, 0);
    }
}

// bias + ReLU in fp32, fp32 store, and the result as chain-ordered hi / lo operands (4 k-steps)
// `bdst` (fp32 blocked output, or null): this lane's 16 bytes of channel group 0's unit of its pixel (+ 4 half floats already
// applied by the caller); the groups are `bstride` floats apart -- one 16-byte store per 4 results instead of four 4-byte ones
__device__ __forceinline__ void finish3(const f32x16 (&acc)[2], const float* __restrict__ bias, int half,
                                        float* __restrict__ dst, unsigned so, int plane, bool st,
                                        u32x4* __restrict__ outh, u32x4* __restrict__ outl,
                                        float* __restrict__ bdst = nullptr, size_t bstride = 0)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = acc[t][4 * g + e] + bias[32 * t + 8 * g + 4 * half + e];
                v[e] = s > 0.0f ? s : 0.0f;
            }
            unsigned h01, l01, h23, l23;
            split2(v[0], v[1], h01, l01);
            split2(v[2], v[3], h23, l23);
            outh[2 * t + (g >> 1)][2 * (g & 1)] = h01; outl[2 * t + (g >> 1)][2 * (g & 1)] = l01;
            outh[2 * t + (g >> 1)][2 * (g & 1) + 1] = h23; outl[2 * t + (g >> 1)][2 * (g & 1) + 1] = l23;
            if (st && bdst) {
                *reinterpret_cast<float4*>(bdst + (size_t)(4 * t + g) * bstride) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (st) {
                float* d = dst + (size_t)(32 * t + 8 * g) * plane;                   // uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) (d + (size_t)e * plane)[so] = v[e];
            }
        }
    }
}

// Persistent workgroups of 8 wavefronts, one per CU, with the 80 KB of hi and lo weight slabs in LDS: read from global
// memory they made 5.2 GB of L2 traffic per 512 images (80 KB per 32-pixel segment through a 32 KB L1), twice the tensors.
constexpr int kX3Waves = 8;
constexpr int kX3Units = 2 * (2 + 2 + 4 + 12) * 128;       // 16-byte units: w0 | w1 | w2 | wg, each [hi|lo]
constexpr int kX3Bytes = kX3Units * 16 + 4 * 64 * 4;

__global__ void __launch_bounds__(kX3Waves * 64)
front_x3_kernel(const FrontXParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* sW = reinterpret_cast<u32x4*>(smem);
    float* sBias = reinterpret_cast<float*>(smem + kX3Units * 16);
    for (int i = threadIdx.x; i < kX3Units; i += kX3Waves * 64) {
        const u32x4* src = i < 512 ? P.w0 + i : i < 1024 ? P.w1 + (i - 512) : i < 2048 ? P.w2 + (i - 1024) : P.wg + (i - 2048);
        sW[i] = *src;
    }
    for (int i = threadIdx.x; i < 256; i += kX3Waves * 64)
        sBias[i] = (i < 64 ? P.b0 : i < 128 ? P.b1 : i < 192 ? P.b2 : P.bg)[i & 63];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int segs_per_row = P.W >> 5;
    const long nseg = (long)P.N * P.H * segs_per_row;
    const int plane = P.H * P.W, W2 = P.W >> 1, plane2 = (P.H >> 1) * W2;
    for (long seg = (long)blockIdx.x * kX3Waves + wv; seg < nseg; seg += (long)gridDim.x * kX3Waves) {
        const int sx = (int)(seg % segs_per_row);
        const long row = seg / segs_per_row;
        const int y = (int)(row % P.H);
        const int n = (int)(row / P.H);
        const size_t seg0 = (size_t)y * P.W + sx * 32;
        const size_t seg2 = (size_t)(y >> 1) * W2 + sx * 16;
        const int xx = sx * 32 + l31;
        const unsigned lo = (unsigned)(l31 + 8 * half * plane);
        const size_t obase = (size_t)n * 64 * plane + seg0;
        const unsigned so = (unsigned)(l31 + 4 * half * plane);
        int opaque = 0;                                  // keep the slab reads inside the loop (tpspp_dgab.hip)
        asm volatile("" : "+s"(opaque));
        const u32x4* w0 = sW + opaque;
        const u32x4* w1 = w0 + 512;
        const u32x4* w2 = w0 + 1024;
        const u32x4* wg = w0 + 2048;
        const float* bias = sBias + opaque;

        const bool st01 = P.feat0 != nullptr;               // (uniform) feat0 / feat1 wanted in HBM at all: tpspp_down_fused_x3_fwd
        u32x4 fh[12], fl[12];
        f32x16 acc[2];
        auto zero = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        };
        {
            u32x4 ih[2], il[2];
            load_b3<2>(P.o0 + (size_t)n * 32 * plane + seg0, lo, plane, ih, il);
            zero();
            gemm3<2>(w0, ih, il, half, l31, acc);
            finish3(acc, bias, half, P.feat0 + obase, so, plane, st01, fh, fl,
                    P.blk ? P.feat0 + ((size_t)n * 8 * plane + seg0 + l31) * 8 + 4 * half : nullptr, (size_t)plane * 8);
        }
        {
            u32x4 ih[2], il[2];
            load_b3<2>(P.o1 + (size_t)n * 32 * plane + seg0, lo, plane, ih, il);
            zero();
            gemm3<2>(w1, ih, il, half, l31, acc);
            finish3(acc, bias + 64, half, P.feat1 + obase, so, plane, st01, fh + 4, fl + 4,
                    P.blk ? P.feat1 + ((size_t)n * 8 * plane + seg0 + l31) * 8 + 4 * half : nullptr, (size_t)plane * 8);
        }
        {
            u32x4 ih[4], il[4];
            load_b3<4>(P.x + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 8 * half * plane2), plane2, ih, il);
            zero();
            gemm3<4>(w2, ih, il, half, l31, acc);
            finish3(acc, bias + 128, half, P.feat2 + (size_t)n * 64 * plane2 + seg2, (unsigned)((l31 >> 1) + 4 * half * plane2),
                    plane2, ((y | xx) & 1) == 0, fh + 8, fl + 8,
                    P.blk ? P.feat2 + ((size_t)n * 8 * plane2 + seg2 + (l31 >> 1)) * 8 + 4 * half : nullptr, (size_t)plane2 * 8);
        }
        zero();
        gemm3<12>(wg, fh, fl, half, l31, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cu = 32 * t + (r & 3) + 8 * (r >> 2);
                float v = acc[t][r] + bias[192 + cu + 4 * half];
                v = v > 0.0f ? v : 0.0f;
                (P.feat_grid + obase + (size_t)cu * plane)[so] = v;
            }
        }
    }
}

}  // namespace

TPSPP_EXPORT int tpspp_front_bf16_fwd(const void* outs0, const void* outs1, const void* x,
                                      const void* w0, const float* b0, const void* w1, const float* b1,
                                      const void* w2, const float* b2, const void* wg, const float* bg,
                                      void* feat0, void* feat1, void* feat2, void* feat_grid, int feat_grid_f32,
                                      int N, int H, int W, int split3, tpspp_stream_t stream)
{
    TPSPP_REQUIRE(outs0 && outs1 && x && w0 && w1 && w2 && wg && b0 && b1 && b2 && bg && feat2 && feat_grid,
                  "tpspp_front_bf16_fwd: null pointer");
    // feat0 = feat1 = NULL: not stored (their consumers recompute them: tpspp_down_fused_bf16_fwd / _x3_fwd); blocked form only
    TPSPP_REQUIRE((feat0 && feat1) || (!feat0 && !feat1 && (feat_grid_f32 & 2)),
                  "tpspp_front_bf16_fwd: feat0 / feat1 may only be omitted together, in the blocked form");
    TPSPP_REQUIRE(N >= 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 32) == 0,
                  "tpspp_front_bf16_fwd: needs an even height and a width that is a multiple of 32");
    if (N == 0) return TPSPP_OK;
    const long nseg = (long)N * H * (W / 32);
    const long blocks = (nseg + 3) / 4;
    TPSPP_REQUIRE(blocks <= 0x7fffffffL, "tpspp_front_bf16_fwd: grid too large");
    if (split3) {
        FrontXParams X;
        X.o0 = static_cast<const float*>(outs0); X.o1 = static_cast<const float*>(outs1); X.x = static_cast<const float*>(x);
        X.w0 = static_cast<const u32x4*>(w0); X.w1 = static_cast<const u32x4*>(w1);
        X.w2 = static_cast<const u32x4*>(w2); X.wg = static_cast<const u32x4*>(wg);
        X.b0 = b0; X.b1 = b1; X.b2 = b2; X.bg = bg;
        X.feat0 = static_cast<float*>(feat0); X.feat1 = static_cast<float*>(feat1); X.feat2 = static_cast<float*>(feat2);
        X.feat_grid = static_cast<float*>(feat_grid);
        X.N = N; X.H = H; X.W = W;
        X.blk = (feat_grid_f32 & 2) ? 1 : 0;
        static bool x3_attr_done[tpspp::kMaxDevices] = {};
        if (tpspp::first_use_on_device(x3_attr_done)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_x3_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        const long wgs3 = (nseg + kX3Waves - 1) / kX3Waves;
        hipLaunchKernelGGL(front_x3_kernel, dim3((unsigned)(wgs3 < 256 ? wgs3 : 256)), dim3(kX3Waves * 64), kX3Bytes,
                           tpspp::as_stream(stream), X);
        return tpspp::check_launch("tpspp_front_bf16_fwd(x3)");
    }
    FrontBParams P;
    P.o0 = static_cast<const unsigned short*>(outs0); P.o1 = static_cast<const unsigned short*>(outs1);
    P.x = static_cast<const unsigned short*>(x);
    P.w0 = static_cast<const u32x4*>(w0); P.w1 = static_cast<const u32x4*>(w1);
    P.w2 = static_cast<const u32x4*>(w2); P.wg = static_cast<const u32x4*>(wg);
    P.b0 = b0; P.b1 = b1; P.b2 = b2; P.bg = bg;
    P.feat0 = static_cast<unsigned short*>(feat0); P.feat1 = static_cast<unsigned short*>(feat1);
    P.feat2 = static_cast<unsigned short*>(feat2); P.feat_grid = feat_grid; P.fg_f32 = (feat_grid_f32 & 1) ? 1 : 0;
    P.blk = (feat_grid_f32 & 2) ? 1 : 0;
    P.N = N; P.H = H; P.W = W;
    TPSPP_REQUIRE(((reinterpret_cast<size_t>(outs0) | reinterpret_cast<size_t>(outs1) | reinterpret_cast<size_t>(x) |
                    reinterpret_cast<size_t>(feat0) | reinterpret_cast<size_t>(feat1) | reinterpret_cast<size_t>(feat2) |
                    reinterpret_cast<size_t>(feat_grid)) & 15) == 0, "tpspp_front_bf16_fwd: tensors must be 16-byte aligned");
    // One workgroup per CU.  bf16 feat_grid: 4 wavefronts = one 128-pixel row per workgroup at a time; measured (same box,
    // interleaved with the previous kernel at 380 us): 3 / 4 / 5 / 6 / 8 wavefronts 373 / 308 / 420 / 383 / 375 us, two
    // workgroups of 4 per CU 371 us, non-temporal stores 369 us, consecutive rows per workgroup 351 us.
    static bool attr_done[tpspp::kMaxDevices] = {};
    if (tpspp::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<true, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<false, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<false, 12, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    // Blocked feat0 / feat1 / feat2 (512-byte store runs instead of 64-byte pieces of 256 planes): more wavefronts per
    // workgroup now help -- 4 / 6 / 8 / 10 / 12 wavefronts measured 357 / 333 / 307 / 311 / 299 us on one box (round 3).
    const int nw = P.fg_f32 ? 8 : (P.blk ? 12 : 4);
    const long wgs = (nseg + nw - 1) / nw;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    auto go = [&](auto kern, int nwv) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nwv * 64), kSlabBytes + nwv * kTileBytes, tpspp::as_stream(stream), P);
    };
    if (!feat0) {
        static bool attr2_done[tpspp::kMaxDevices] = {};
        if (tpspp::first_use_on_device(attr2_done)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<true, 8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&front_bf16_kernel<false, 12, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        if (P.fg_f32) go(front_bf16_kernel<true, 8, true, false>, 8); else go(front_bf16_kernel<false, 12, true, false>, 12);
    } else
    if (P.fg_f32) { if (P.blk) go(front_bf16_kernel<true, 8, true>, 8); else go(front_bf16_kernel<true, 8, false>, 8); }
    else          { if (P.blk) go(front_bf16_kernel<false, 12, true>, 12); else go(front_bf16_kernel<false, 4, false>, 4); }
    return tpspp::check_launch("tpspp_front_bf16_fwd");
}
