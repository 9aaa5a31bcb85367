// Token GEMM of the recogniser head on the bf16 matrix cores (round 4): out (Co, M) = act(W^T X + bias) [+ res] with
// channel-major fp32 activations X (K, M), M = images x tokens (32,768 at batch 512): the encoder's q|k|v / fc / w1 / w2
// projections and the decoder's one-off key / value projections of the encoder output in the bf16 and "bf16x3"
// configurations.
// Replaces (in those configurations): nn.Linear inside MultiHeadAttention / PositionwiseFeedForward,
// common/layers/transformer_layers.py:36-75, common/modules/transformer_module.py; encoders/nrtr_encoder.py:67-87.
//
// These products went through the convolution kernel as 1x1 convolutions before: 64 output channels per workgroup, so a
// 1536-wide projection staged the same activations 24 times, element by element (tpspp_conv_bf16_impl.h: "KC 2-byte loads
// and KC/2 packing instructions per position -- the instruction stream is what that staging is bound by"), and wrote its
// results 4 bytes per lane: 139 us for q|k|v, 123 / 80 us for the two projections with a residual.  Here:
//   * a workgroup owns 256 (128 where that leaves fewer than two workgroups per CU) output channels x 128 tokens: four
//     wavefronts of 64 (32) outputs x 128 tokens = 2 (1) x 4 v_mfma_f32_32x32x16_bf16 tiles, K in stages of 32; every
//     B fragment read from LDS feeds two matrix instructions, every weight fragment four;
//   * X arrives as 16-byte pieces of its rows (512 contiguous bytes per k), is rounded to bf16 (x3: split into hi and
//     lo) in registers and laid down [k][token] in LDS at a pitch of 80 words: ds_read_b64_tr_b16 hands a lane 4
//     consecutive k of ITS token, two reads per k-step (the four rows and two column blocks of a 32-lane access fall on 64
//     different banks: SQ_LDS_BANK_CONFLICT = 10 % of the LDS cycles, the LDS busy 14 % of the time); the next stage's
//     loads are in flight under this stage's matrix instructions, one barrier per stage;
//   * the weight is read straight from the convolution's arranged copy ([64 outputs][8 k] units = the A operand's register
//     image: one 16-byte load per lane and 32 x 16 block), one stage ahead;
//   * the accumulators (lane = token) leave through a [32 outputs][64 tokens] tile of LDS per wavefront, as 16-byte pieces
//     of the output rows -- 256 contiguous bytes per row and instruction; the residual is read the same way, all of a
//     pass's pieces before its first store (read next to the stores it cost 75 us: `out` may alias `res` for all the
//     compiler knows, so every 4-byte load waited for the store before it);
//   * block -> (token tile, output tile) keeps the workgroups that share a token tile on one XCD (blocks go to XCDs
//     round-robin), so X is re-used out of that XCD's L2 (measured: within 3 % of the plain order).
// fp32 accumulation over k ascending; bias, GELU (erf) and the fp32 residual in the epilogue; fp32 or bf16 output.
// MI355X, 32,768 tokens (kernel trace, scripts/debug/bench_encoder.py): q|k|v 512 -> 1536 in 92 us (0.56 PFLOP/s; its
// 268 MB of fp32 activations and results take 45 us at 6 TB/s), fc / w2 (512 outputs, residual) 43 us, w1 33 us; the
// three-term split 169 / 66 / 44 us.  What is left: matrix pipe busy 22 % (the loop alone, without global loads, runs at
// 47 %: one wavefront's LDS reads are not yet overlapped with its own matrix instructions), stores at HBM speed but not
// under another workgroup's loop.
#include "tpspp_common.h"
#include <cstdlib>
#include "tpspp_tokgemm.h"

namespace {

constexpr int kWave = 64;
constexpr int BM = 128, KST = 32;                // tokens per workgroup, k per stage
constexpr int PITCH = 160;                       // 16-bit elements per k row of the LDS tile (80 words)
constexpr int EPITCH = 68;                       // words per output row of the epilogue's tile

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    f32x2 v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u32x2 read_tr(const unsigned short* p)
{
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

template <bool X3, int NCI>
__global__ void __launch_bounds__(256, 2)
tok_gemm_kernel(const tpspp::TokGemmArgs P)
{
    constexpr int HL = X3 ? 2 : 1;
    constexpr int BN = 128 * NCI;                           // outputs per workgroup: four wavefronts of 32 NCI x 128 tokens
    constexpr int STAGE_BYTES = 2 * HL * KST * PITCH * 2, EPI_BYTES = 4 * 32 * EPITCH * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    auto sX = reinterpret_cast<unsigned short(*)[HL][KST][PITCH]>(smem);
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5, a = lane & 15, g = lane >> 4;
    const int K = P.K, M = P.M;
    const int nct = P.Co / BN, ntt = (M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tt = (slot / nct) * 8 + xcd, ctile = slot % nct;
    if (tt >= ntt) return;
    const int m0 = tt * BM, co0 = ctile * BN;

    // ---- staging: a stage is 32 k x 128 tokens = 1024 pieces of 16 bytes, four per thread ----
    float4 xr[4];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 5, c4 = idx & 31;
            int m = m0 + 4 * c4;
            if (m + 3 >= M) m = M - 4;                       // the tail tile re-reads valid tokens (its stores are masked)
            xr[i] = *reinterpret_cast<const float4*>(P.X + (size_t)(s * KST + row) * M + m);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 5, c4 = idx & 31;
            const unsigned h0 = pack2(xr[i].x, xr[i].y), h1 = pack2(xr[i].z, xr[i].w);
            u32x2 hv; hv[0] = h0; hv[1] = h1;
            *reinterpret_cast<u32x2*>(&sX[buf][0][row][4 * c4]) = hv;
            if (X3) {
                const float f0 = __builtin_bit_cast(float, h0 << 16), f1 = __builtin_bit_cast(float, h0 & 0xffff0000u);
                const float f2 = __builtin_bit_cast(float, h1 << 16), f3 = __builtin_bit_cast(float, h1 & 0xffff0000u);
                u32x2 lv; lv[0] = pack2(xr[i].x - f0, xr[i].y - f1); lv[1] = pack2(xr[i].z - f2, xr[i].w - f3);
                *reinterpret_cast<u32x2*>(&sX[buf][HL - 1][row][4 * c4]) = lv;
            }
        }
    };
    // ---- the weight: 16-byte unit of (64-output tile, k group gk, output co, hi / lo); the wavefront's 32 NCI outputs ----
    const u32x4* W = reinterpret_cast<const u32x4*>(P.W);
    const int cow = co0 + wv * (32 * NCI);                  // first output of this wavefront
    const int ct64 = cow >> 6, cin = cow & 63, nch = (K / 8) / P.kgc;
    auto a_unit = [&](int gk, int co, int hl) -> const u32x4* {
        const int chunk = gk / P.kgc, kg = gk - chunk * P.kgc;
        return W + ((size_t)((ct64 * nch + chunk) * HL + hl) * P.kgc + kg) * 64 + cin + co;
    };
    u32x4 ah[2][NCI], al[2][NCI];                           // [k-step of the stage][32-output tile]
    auto load_a = [&](int s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) {
                const int gk = (s * KST) / 8 + 2 * ks + half;
                ah[ks][ci] = *a_unit(gk, ci * 32 + l31, 0);
                if (X3) al[ks][ci] = *a_unit(gk, ci * 32 + l31, 1);
            }
    };

    f32x16 acc[NCI][4];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ci][ti][r] = 0.0f;

    const int nst = K / KST;
    load_stage(0);
    load_a(0);
    store_stage(0);
    __syncthreads();
    // the lane's element offset inside a stage's tile for the transposing reads: row 8 half + (a >> 2) (+ 4 for the second
    // read), column 16 (g & 1) + 4 (a & 3) (+ 32 ti)
    const int lb = (8 * half + (a >> 2)) * PITCH + 16 * (g & 1) + 4 * (a & 3);
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        u32x4 ch[2][NCI], cl[2][NCI];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) { ch[ks][ci] = ah[ks][ci]; if (X3) cl[ks][ci] = al[ks][ci]; }
        if (s + 1 < nst) { load_stage(s + 1); load_a(s + 1); }     // in flight under this stage's matrix instructions
        // the stage's eight B fragments (k-step ks, 32-token tile ti) are requested kPD fragments ahead of their products
        // through a register ring; the scheduling barriers keep that order (left alone the compiler issues every
        // fragment's reads right in front of its products: one LDS latency per fragment in every wavefront)
        constexpr int kPD = 2;
        u32x4 bh[kPD + 1], bl[kPD + 1];
        auto fetch_b = [&](int i, int slot) {
            const int ks = i >> 2, ti = i & 3;
            const unsigned short* p = &sX[buf][0][0][0] + lb + (16 * ks) * PITCH + 32 * ti;
            const u32x2 k0 = read_tr(p), k1 = read_tr(p + 4 * PITCH);
            bh[slot][0] = k0[0]; bh[slot][1] = k0[1]; bh[slot][2] = k1[0]; bh[slot][3] = k1[1];
            if (X3) {
                const unsigned short* q = &sX[buf][HL - 1][0][0] + lb + (16 * ks) * PITCH + 32 * ti;
                const u32x2 j0 = read_tr(q), j1 = read_tr(q + 4 * PITCH);
                bl[slot][0] = j0[0]; bl[slot][1] = j0[1]; bl[slot][2] = j1[0]; bl[slot][3] = j1[1];
            }
        };
#pragma unroll
        for (int i = 0; i < kPD; ++i) fetch_b(i, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = i >> 2, ti = i & 3, slot = i % (kPD + 1);
            if (i + kPD < 8) fetch_b(i + kPD, (i + kPD) % (kPD + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) {
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, ch[ks][ci]), Bh = __builtin_bit_cast(bf16x8, bh[slot]);
                acc[ci][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[ci][ti], 0, 0, 0);
                if (X3) {
                    const bf16x8 Al = __builtin_bit_cast(bf16x8, cl[ks][ci]), Bl = __builtin_bit_cast(bf16x8, bl[slot]);
                    acc[ci][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[ci][ti], 0, 0, 0);
                    acc[ci][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[ci][ti], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < nst) store_stage(buf ^ 1);               // (that buffer was last read before the previous barrier)
        __syncthreads();
    }

    // ---- epilogue: the accumulators (lane = token, register r = output 8 (r / 4) + 4 half + r % 4 of a 32-output tile)
    // go through a [32 outputs][64 tokens] fp32 tile of LDS per wavefront and leave as 16-byte pieces of the output rows
    // (256 contiguous bytes per row and store instruction); bias, GELU and the residual (read the same way) on the way out
    float* ep = reinterpret_cast<float*>(smem) + wv * (32 * EPITCH);
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[(8 * (r >> 2) + 4 * half + (r & 3)) * EPITCH + 32 * t2 + l31] = acc[ci][2 * tp + t2][r];
            float4 v[8], rs[8];
            const int m = m0 + 64 * tp + 4 * a;
            const bool ok = m < M;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = g + 4 * j;
                v[j] = *reinterpret_cast<const float4*>(ep + row * EPITCH + 4 * a);
                if (P.res && ok) rs[j] = *reinterpret_cast<const float4*>(P.res + (size_t)(cow + ci * 32 + row) * M + m);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = cow + ci * 32 + g + 4 * j;
                const float bv = P.bias ? P.bias[co] : 0.0f;
                float o[4] = {v[j].x + bv, v[j].y + bv, v[j].z + bv, v[j].w + bv};
                if (P.act == 2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = 0.5f * o[e] * (1.0f + erff(o[e] * 0.70710678118654752440f));
                if (P.res && ok) { o[0] += rs[j].x; o[1] += rs[j].y; o[2] += rs[j].z; o[3] += rs[j].w; }
                if (ok) {
                    const size_t off = (size_t)co * M + m;
                    if (P.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.out) + off) = make_float4(o[0], o[1], o[2], o[3]);
                    else {
                        unsigned h[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned u = __builtin_bit_cast(unsigned, o[e]);
                            u += 0x7fffu + ((u >> 16) & 1u);
                            h[e] = u >> 16;
                        }
                        u32x2 pk; pk[0] = h[0] | (h[1] << 16); pk[1] = h[2] | (h[3] << 16);
                        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(P.out) + off) = pk;
                    }
                }
            }
        }
}

}  // namespace

namespace tpspp {

bool tok_gemm_applicable(const TokGemmArgs& a)
{
    return a.K > 0 && a.K % KST == 0 && a.kgc > 0 && (a.K / 8) % a.kgc == 0 && a.Co > 0 && a.Co % 128 == 0 && a.M >= 4 &&
           a.M % 4 == 0 && (reinterpret_cast<uintptr_t>(a.X) % 16) == 0 && (a.act == 0 || a.act == 2) &&
           (!a.res || reinterpret_cast<uintptr_t>(a.res) % 16 == 0) && reinterpret_cast<uintptr_t>(a.out) % 16 == 0;
}

void launch_tok_gemm(const TokGemmArgs& a, hipStream_t st)
{
    const int ntt = (a.M + BM - 1) / BM;
    // 256 outputs per workgroup where that still leaves two workgroups per CU, else 128
    // (round 6: only the three-term split keeps the 256-output tile -- in plain bf16 the 128-output tile's 128 registers
    // (four workgroups per CU against two at 208) beat the halved re-staging of X: encoder 2.42 -> 2.33 ms at batch 512;
    // with the split it is the other way round, 3.15 against 3.21)
    const bool wide = a.x3 && a.Co % 256 == 0 && (a.Co / 256) * ntt >= 512;
    const int nct = a.Co / (wide ? 256 : 128);
    const unsigned blocks = (unsigned)(((ntt + 7) / 8) * 8 * nct);
    if (a.x3) {
        if (wide) hipLaunchKernelGGL((tok_gemm_kernel<true, 2>), dim3(blocks), dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((tok_gemm_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, a);
    } else {
        if (wide) hipLaunchKernelGGL((tok_gemm_kernel<false, 2>), dim3(blocks), dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((tok_gemm_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, a);
    }
}

}  // namespace tpspp

// out (Co, M) = act(W^T X + bias) [+ res] on channel-major tokens: the C-ABI face of the kernel above (include/tpspp.h)
TPSPP_EXPORT int tpspp_token_gemm_bf16_fwd(const float* X, const void* w_arranged, const float* bias, const float* res,
                                           void* out, int out_f32, int K, int Co, int M, int act, int split3,
                                           tpspp_stream_t stream)
{
    TPSPP_REQUIRE(X && w_arranged && out, "tpspp_token_gemm_bf16_fwd: null pointer");
    tpspp::TokGemmArgs a;
    a.X = X; a.W = w_arranged; a.bias = bias; a.res = res; a.out = out; a.out_f32 = out_f32 ? 1 : 0;
    a.K = K; a.Co = Co; a.M = M; a.act = act; a.x3 = split3 ? 1 : 0; a.kgc = tpspp_conv_bf16_chunk_channels(1) / 8;
    TPSPP_REQUIRE(tpspp::tok_gemm_applicable(a),
                  "tpspp_token_gemm_bf16_fwd: needs K a multiple of 32, Co a multiple of 128, M a multiple of 4 (>= 4), act 0 or 2 "
                  "and 16-byte aligned tensors (got K = %d, Co = %d, M = %d, act = %d)", K, Co, M, act);
    tpspp::launch_tok_gemm(a, tpspp::as_stream(stream));
    return tpspp::check_launch("tpspp_token_gemm_bf16_fwd");
}
