// The greedy decoder's STEP as one persistent launch (round 5).  Included by tpspp_head.hip
// inside its anonymous namespace (it reuses the step kernels' device code: Wide<>, self_attend, cross_attend, DGemm).
//
// Replaces, per decoding step: mmocr/models/textrecog/decoders/nrtr_decoder.py:153-177 (one position through
// `_attention` + classifier + soft-max / arg-max) = 6 x common/layers/transformer_layers.py:133-163 -- the same
// arithmetic, in the same order, as the launch-per-phase pipeline of tpspp_nrtr_decoder_fwd (its 50 launches per step): the
// exact-fp32 head's scores are BIT-IDENTICAL to that pipeline; the reduced-precision heads (bf16 / bf16x3) split a projection's
// K over eight wavefronts instead of four, so their scores agree within 2e-5 with identical decided tokens
// (tests/test_gpu_head.py::test_decoder_persistent_step_matches_the_launch_pipeline).
//
// Why one launch: a step is a chain of 50 dependent phases of 4-9 us each, and every launch pays 2.8 us before its first
// instruction plus 1.35 us of gap (scripts/ubench/gemm_chain_bench.hip: 6.35 us per dependent projection launch).  The
// images of a batch are independent, so a phase only ever needs data of its own 32 images: the 16 workgroups that own a
// 32-image token block ("cluster") synchronise among themselves, clusters never talk to each other, and nothing needs a
// grid-wide barrier.  scripts/ubench/persist_chain_bench.hip (the projection chain in this form): cluster barrier 0.7 us,
// whole phase 4.2 us against 6.35.
//
// Mechanics (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * everything a phase hands to the next one (x, y, a, q|k|v, hidden, logits) is stored, drained (vmcnt(0)) before the
//     workgroup arrives at the barrier, and read with system-scope loads (sc0 sc1: they bypass the reader's L1 and are served
//     by its XCD's L2) -- no release / acquire fences (a fence writes back / invalidates whole caches: 1.7-6.5 us each).
//     The stores: a cluster's 16 workgroups are the blocks 8 (16 j + ct) + x -- blocks go to the XCDs round-robin, so the
//     cluster sits on XCD x and shares ONE L2.  Each launch verifies that (every workgroup ORs its HW_REG_XCC_ID into the
//     cluster's mask in front of an extra barrier) and, if it holds, stores with ORDINARY stores: the line stays in that L2
//     and the cluster's reads hit it; if it does not hold (another partition mode, a different dispatcher), or with
//     TPSPP_HEAD_WRITE_THROUGH=1, with write-through stores (sc0 sc1) as the first version did -- correct under any
//     placement, the reader fetches over the fabric.  Same bits either way;
//   * barrier = one relaxed agent-scope atomic on the cluster's counter (monotonic over the whole batch: target = 16 x
//     barriers so far) + a bounded poll with s_sleep by one lane; a timeout raises *err, turns the step's scores of the
//     workgroup's images into NaN and the workgroup leaves (no hang: the GPU box survives a protocol bug);
//   * a projection's weights (64-192 KB per workgroup, read-only), column sums, bias and residual are requested BEFORE the
//     barrier, its activations after it: the 32 x K block of the cluster arrives as whole rows (1 KB per wavefront
//     instruction) and is re-read from LDS in the matrix core's fragment order (partial-line system-scope loads cost 0.8 us
//     more per phase);
//   * buffers are indexed so that an image's data is only ever touched by its own cluster (the launch pipeline packs the
//     cross-attention's q rows at pitch C into the q|k|v buffer -- harmless when phases are launches, a cross-cluster
//     overwrite here: q keeps the q|k|v row pitch);
//   * read-only operands (weights, encoder keys / values) and the self-attention caches (written by this launch, read by
//     LATER launches only) use plain accesses.
// One launch per DECODE (the kernel loops over the steps; ~2000 launches per batch before), 512 threads per workgroup.  Projections: all 8 wavefronts split K
// (dec_gemm_x3_kernel: 4 -- a 512-thread workgroup has 256 registers per lane; with an eighth of K per wavefront the
// weights of all three q|k|v tiles can be requested before the barrier), so a result's last bits may differ from the
// launch pipeline's: soft-max scores within 2e-5, decided tokens identical (tests/test_gpu_head.py).  Attentions: a
// wavefront owns both heads of a head pair of one image and runs every stage for both before the next stage starts (8
// wavefronts per CU must keep as many requests in flight as the launch form's 16).  Workgroup w of a cluster owns output
// tile w of a projection (tiles 3 w .. 3 w + 2 of q|k|v) and images 2 w, 2 w + 1 in the attentions.  At most 512 images
// per launch (16 clusters = 256 workgroups, one per CU, two clusters per XCD): larger batches run as several launches.
// Odd clusters start `stagger` late: with every cluster in the same phase the memory system is saturated during the
// attentions and idle during the latency-bound projections; half a phase apart the two halves of the chip alternate.
// (With the step loop inside the launch the clusters drift apart on their own: the stagger no longer measures.)
// Measured (batch 512, 40 steps, MI355X; launch pipeline -> clusters spread over the XCDs, write-through stores, one launch
// per step -> cluster per XCD, ordinary stores, one launch per decode): fp32 head 23.9 -> 22.5 -> 21.5 ms, bf16x3 head
// 21.0 -> 19.7 -> 18.9 ms, bf16 head 15.5 -> 15.0 -> 13.5 ms; per layer-step of the first persistent version
// (scripts/debug/trace_decoder_step.py): q|k|v 9.1, self-attention 14.0, x+fc 6.6, q 5.5, cross-attention 22.4 (launch
// form: 28.6), x+fc 6.8, w1 4.5, w2 6.5 us -- a projection phase is barrier 1.5 (incl. the cluster's skew) + rows 1.5 +
// products / reduction / epilogue 1.2-1.6 + store drain 0.5-0.8 us.
#pragma once

struct PLayer {
    const void *qkv_x, *wfc_x, *q_x, *wfc2_x, *w1_x, *w2_x;     // arranged weights: hi / lo bf16 (ops.arrange_x3) or fp32 (ops.arrange_f32)
    const float *qkv_cs, *qkv_b, *bfc, *q_cs, *q_b, *bfc2, *w1_cs, *w1_b, *b2;
    const void *Kx, *Vx;        // encoder keys / values of the layer, token-major (fp32 or bf16)
    void *Kc, *Vc;              // self-attention caches
};
constexpr int kPMaxLayers = 8;

struct PStep {
    PLayer L[kPMaxLayers];
    int n_layers;
    float *x, *y, *a, *qkv, *hid, *logits;    // step buffers (token-major), exchanged inside the launch
    const void* cls_x; const float* cls_cs; const float* cls_b; int num_out;
    const float *emb, *pos;
    int* tokens; int Lt; float* out; int greedy; int pad_idx;
    const int* valid_len;
    int N, n0, C, T, H, d_inner, step, nsteps, Lsteps, Lmax;   // images [n0, n0 + up to 512) of N; steps [step, step + nsteps) in this launch
    int* counters; int bar_base;              // cluster counters (128 B apart; word 1: the cluster's XCD mask), barriers passed before this launch
    int nclusters;                            // clusters of this launch (grid = ceil(nclusters / 8) x 128 workgroups: see the kernel)
    int pairs;                                // (image, head) pairs per wavefront in the attention phases (2; a run-time value: the
                                              // pair loop must stay a loop -- unrolled, the compiler interleaves two attentions' registers)
    int* err;
    int stagger;                              // odd clusters start this many 10-ns ticks late (see the kernel); 0 = together
    long long* trace;                         // optional (tpspp_head_set_trace): 64 wall-clock stamps (100 MHz) per workgroup and step
    int no_plain;                             // lab / test switch (TPSPP_HEAD_WRITE_THROUGH=1): write-through stores whatever the placement
    int timeout_k;                            // cluster-barrier timeout, units of 1024 ticks of the 100 MHz wall clock (10.24 us)
    int test_stall_step;                      // test hook (TPSPP_HEAD_TEST_STALL=step): workgroup 3 of the launch's first cluster sits out
                                              // two timeouts at the start of that step -- its partners time out: the failure path, on demand; -1 off
};

constexpr int kPXPitch = 516;                 // floats per staged X row (512 + 4: fragment reads hit all banks)

// ---- cluster barrier ----------------------------------------------------------------------------------------------------
// (workgroup barriers as raw s_barrier + lgkmcnt only: a __syncthreads() also waits for every outstanding VMEM request, i.e.
// for exactly the prefetches -- weights, residual, cached keys -- that are meant to fly across this barrier)
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sFlag[0]: the barrier's verdict for the workgroup; sFlag[1]: the timeout in units of 1024 ticks of the 100 MHz wall clock
// (PShared::flag / ::timeout_k, set once per launch from PStep::timeout_k).  The timeout is WALL-CLOCK time, not a poll count:
// a cluster whose partners are not resident yet (the device is partly occupied by another stream's kernel) waits for them as
// long as the host allows (tpspp.h: TPSPP_HEAD_TIMEOUT_MS, default 4 s), whatever a poll costs.
__device__ __forceinline__ bool cluster_barrier(int* cnt, int target, int* sFlag, int* err)
{
    wg_barrier_lds();                                        // every wavefront has drained its stores (drain_stores())
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        const long long t0 = (long long)wall_clock64();
        const long long lim = (long long)sFlag[1] << 10;
        for (unsigned spin = 1;; ++spin) {
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            if ((spin & 63u) == 0 && (long long)wall_clock64() - t0 > lim) break;
        }
        if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sFlag[0] = ok;
    }
    wg_barrier_lds();
    return sFlag[0] != 0;
}

// ---- a projection phase: out (M, Co) = act(LN?(X) W + bias) [+ res] for the cluster's 32 tokens, tiles ct0 .. ct0 + NT - 1.
// The arithmetic is dec_gemm_x3_kernel's, instruction for instruction (same fragments, k order, three-term products,
// reduction order, LayerNorm statistics, epilogue); what differs: X arrives as rows -> LDS -> fragments, a workgroup's
// tiles are worked off one after the other (the next tile's weights are requested under the current tile's products: two
// tiles of weights in registers instead of three -- a 512-thread workgroup has 256 registers per lane), stores are
// system-scope.  Called by ALL 8 wavefronts (the barriers count every wavefront); wavefronts 0 - 3 do the products.
struct PGemm {
    const float* X; const void* Wp; const float* bias; const float* colsum; const float* res; float* out;
    int M, Co; float eps; int act;
    int ldo;                    // row pitch of `out` and `res` in floats (Co, except the q projection: see the step kernel)
    bool plain;                 // the cluster sits on one XCD: ordinary stores (st16_x)
};

// KSW = 16-wide k-steps per wavefront = K / 128: the EIGHT wavefronts of the workgroup split K (the launch-per-phase kernel
// splits it over four: a 512-thread workgroup has 256 registers per lane, and with an eighth of K per wavefront the weights
// of all three q|k|v tiles -- 96 registers -- can be requested before the barrier)
template <int KSW>
__device__ __forceinline__ void pgemm_load_w(du32x4 (&ah)[KSW], du32x4 (&al)[KSW], const PGemm& G, int ct, int wv, int half, int l31)
{
    constexpr int KS = 8 * KSW;                             // 16-wide k-steps of the whole K
    const du32x4* wp = reinterpret_cast<const du32x4*>(G.Wp) + ((size_t)(ct * KS + wv * KSW) * 4 + half) * 32 + l31;
#pragma unroll
    for (int j = 0; j < KSW; ++j) { ah[j] = wp[(size_t)j * 128]; al[j] = wp[(size_t)j * 128 + 64]; }
}

// all 8 wavefronts: the cluster's 32 x K block of X (rows m0 .. m0 + 31, clamped to M - 1) -> sX
template <int K>
__device__ __forceinline__ void stage_rows(const float* X, int m0, int M, float* sX, int wv8, int lane)
{
    // row = 4 K bytes = K / 256 wavefront instructions of 1 KB; wavefront w: rows 4 w .. 4 w + 3
    constexpr int PPR = K / 256;                             // pieces per row
    hf32x4 t[4 * PPR];
#pragma unroll
    for (int i = 0; i < 4 * PPR; ++i) {
        const int r = 4 * wv8 + i / PPR;
        const int m = m0 + r < M ? m0 + r : M - 1;
        t[i] = ld16_sys(X + (size_t)m * K + 256 * (i % PPR) + 4 * lane);
    }
    if constexpr (PPR == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
#pragma unroll
    for (int i = 0; i < 4 * PPR; ++i)
        *reinterpret_cast<hf32x4*>(&sX[(4 * wv8 + i / PPR) * kPXPitch + 256 * (i % PPR) + 4 * lane]) = t[i];
}

struct PShared {
    float sX[32 * kPXPitch];
    float sRed[8][16][kWave];
    float sS1[16][32], sS2[16][32];
    int flag, timeout_k;                      // cluster_barrier's sFlag[0], sFlag[1] (adjacent, in this order)
};

// One projection phase, called by all 8 wavefronts.  Returns false after a barrier timeout.
// Differences from dec_gemm_x3_kernel in the arithmetic: K is split over 8 wavefronts instead of 4, so the 8 partial sums
// of an output meet as ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) and the LayerNorm sums as 16 partials in ascending
// order -- the last bits of a result may differ from the launch-per-phase pipeline's (tests: <= 2e-6 of the scores).
template <int KSW, bool LN, int NT>
__device__ __forceinline__ bool pgemm_phase(const PGemm& G, int tb, int ct0, PShared& S, int* cnt, int target, int* err,
                                          long long* sub = nullptr)
{
    constexpr int K = 128 * KSW;
    // (opaque per phase: otherwise the compiler hoists every phase's lane-dependent addresses out of the layer loop and
    // spills them -- 78 scratch stores in the prologue)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ntiles = (G.Co + 31) >> 5;
    const bool wg_active = ct0 < ntiles;                   // (w1: 8 tiles, classifier: 3 -- the other workgroups only pass the barrier)
    const int m0 = tb * 32;                                 // (tb: the cluster's token block in the whole batch)
    du32x4 ah[NT][KSW], al[NT][KSW];
    // in front of the barrier: everything that does not depend on the previous phase -- the weights, and the epilogue's
    // operands (column sums, bias, and the residual: it was produced at least two phases ago, i.e. behind barriers this
    // workgroup has already passed)
    const int m = m0 + l31;
    const int mc = m < G.M ? m : G.M - 1;
    float4 c4[NT], bb[NT];
    hf32x4 r4[NT];
    if (wg_active) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const int ct = ct0 + tl;
            pgemm_load_w<KSW>(ah[tl], al[tl], G, ct < ntiles ? ct : ntiles - 1, wv, half, l31);
            const int c = ct * 32 + 8 * (wv & 3) + 4 * half;
            c4[tl] = make_float4(0.f, 0.f, 0.f, 0.f); bb[tl] = c4[tl]; r4[tl] = hf32x4{0.f, 0.f, 0.f, 0.f};
            if (wv < 4 && ct < ntiles && m < G.M && c < G.Co) {
                if (LN) c4[tl] = *reinterpret_cast<const float4*>(G.colsum + c);
                if (G.bias) bb[tl] = *reinterpret_cast<const float4*>(G.bias + c);
                if (G.res) {
                    // (compiler-visible system-scope loads: an inline-asm load whose wait sits far away may have its
                    // destination register copied by the compiler before the data has arrived)
                    const unsigned long long* rp = reinterpret_cast<const unsigned long long*>(G.res + (size_t)mc * G.ldo + c);
                    const unsigned long long lo = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long hi = __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    r4[tl][0] = __builtin_bit_cast(float, (unsigned)lo); r4[tl][1] = __builtin_bit_cast(float, (unsigned)(lo >> 32));
                    r4[tl][2] = __builtin_bit_cast(float, (unsigned)hi); r4[tl][3] = __builtin_bit_cast(float, (unsigned)(hi >> 32));
                }
            }
        }
    }
    if (sub && tid == 0) sub[0] = (long long)__builtin_amdgcn_s_memtime();
    if (!cluster_barrier(cnt, target, &S.flag, err)) return false;
    if (sub && tid == 0) sub[1] = (long long)__builtin_amdgcn_s_memtime();
    if (!wg_active) return true;
    stage_rows<K>(G.X, m0, G.M, S.sX, wv, lane);
    __syncthreads();
    if (sub && tid == 0) sub[2] = (long long)__builtin_amdgcn_s_memtime();
    float s1 = 0.0f, s2 = 0.0f, mean = 0.0f, rstd = 1.0f;
    const float* xs = S.sX + l31 * kPXPitch + 16 * (wv * KSW) + 8 * half;
    // the lane's B fragments (hi / lo halves of its 8 k per k-step) once, for every tile
    du32x4 bh[KSW], bl[KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + 16 * j), x1 = *reinterpret_cast<const float4*>(xs + 16 * j + 4);
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned pk = dpack2(x[2 * q], x[2 * q + 1]);
            const float h0 = __builtin_bit_cast(float, pk << 16), h1 = __builtin_bit_cast(float, pk & 0xffff0000u);
            bh[j][q] = pk;
            bl[j][q] = dpack2(x[2 * q] - h0, x[2 * q + 1] - h1);
            if (LN) {
                s1 += x[2 * q] + x[2 * q + 1];
                s2 = fmaf(x[2 * q], x[2 * q], s2);
                s2 = fmaf(x[2 * q + 1], x[2 * q + 1], s2);
            }
        }
    }
    if (LN) { S.sS1[wv * 2 + half][l31] = s1; S.sS2[wv * 2 + half][l31] = s2; }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int ct = ct0 + tl;
        const int c = ct * 32 + 8 * (wv & 3) + 4 * half;
        // wavefronts 0 - 3 finish the tile: accumulator registers 4 w .. 4 w + 3 = outputs 32 ct + 8 w + 4 half + (0 .. 3) of token l31
        const bool mine = wv < 4 && ct < ntiles && m < G.M && c < G.Co;
        const size_t o = (size_t)mc * G.ldo + (c < G.Co ? c : 0);
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const dbf16x8 Bh = __builtin_bit_cast(dbf16x8, bh[j]), Bl = __builtin_bit_cast(dbf16x8, bl[j]);
            const dbf16x8 Ah = __builtin_bit_cast(dbf16x8, ah[tl][j]), Al = __builtin_bit_cast(dbf16x8, al[tl][j]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) S.sRed[wv][r][lane] = acc[r];
        __syncthreads();
        if (LN && tl == 0) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { t1 += S.sS1[j][l31]; t2 += S.sS2[j][l31]; }
            mean = t1 / (float)K;
            rstd = 1.0f / sqrtf(fmaxf(t2 / (float)K - mean * mean, 0.0f) + G.eps);
        }
        if (mine) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * wv + e;
                v[e] = ((S.sRed[0][r][lane] + S.sRed[1][r][lane]) + (S.sRed[2][r][lane] + S.sRed[3][r][lane])) +
                       ((S.sRed[4][r][lane] + S.sRed[5][r][lane]) + (S.sRed[6][r][lane] + S.sRed[7][r][lane]));
            }
            if (LN) {
                v[0] = rstd * (v[0] - mean * c4[tl].x); v[1] = rstd * (v[1] - mean * c4[tl].y);
                v[2] = rstd * (v[2] - mean * c4[tl].z); v[3] = rstd * (v[3] - mean * c4[tl].w);
            }
            if (G.bias) { v[0] += bb[tl].x; v[1] += bb[tl].y; v[2] += bb[tl].z; v[3] += bb[tl].w; }
            if (G.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
            }
            if (G.res) { v[0] += r4[tl][0]; v[1] += r4[tl][1]; v[2] += r4[tl][2]; v[3] += r4[tl][3]; }
            st16_x(G.out + o, hf32x4{v[0], v[1], v[2], v[3]}, G.plain);
        }
        if (tl + 1 < NT) __syncthreads();                  // the partial sums have been read: sRed is free for the next tile
    }
    if (sub && tid == 0) sub[3] = (long long)__builtin_amdgcn_s_memtime();
    drain_stores();
    if (sub && tid == 0) sub[4] = (long long)__builtin_amdgcn_s_memtime();
    return true;
}

// ---- the same phase with EXACT fp32 products (the exact-fp32 head): dec_gemm_f32_kernel's arithmetic, instruction for
// instruction -- v_mfma_f32_32x32x2_f32 on the same fragments, K split over the same 8 wavefronts, the same reduction order --
// so the scores are bit-identical to the launch pipeline's.  KUW = 8-wide k-units per wavefront = K / 64.
template <int KUW, bool LN, int NT>
__device__ __forceinline__ bool pgemm_phase_f32(const PGemm& G, int tb, int ct0, PShared& S, int* cnt, int target, int* err)
{
    constexpr int K = 64 * KUW, KU = 8 * KUW;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // (opaque per phase: see pgemm_phase)
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ntiles = (G.Co + 31) >> 5;
    const bool wg_active = ct0 < ntiles;
    const int m0 = tb * 32;
    const int m = m0 + l31;
    const int mc = m < G.M ? m : G.M - 1;
    float4 wa[NT][KUW], c4[NT], bb[NT];
    hf32x4 r4[NT];
    if (wg_active) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const int ct = ct0 + tl;
            const float4* wp = reinterpret_cast<const float4*>(G.Wp) + ((size_t)((ct < ntiles ? ct : ntiles - 1) * KU + wv * KUW) * 2 + half) * 32 + l31;
#pragma unroll
            for (int j = 0; j < KUW; ++j) wa[tl][j] = wp[(size_t)j * 64];
            const int c = ct * 32 + 8 * (wv & 3) + 4 * half;
            c4[tl] = make_float4(0.f, 0.f, 0.f, 0.f); bb[tl] = c4[tl]; r4[tl] = hf32x4{0.f, 0.f, 0.f, 0.f};
            if (wv < 4 && ct < ntiles && m < G.M && c < G.Co) {
                if (LN) c4[tl] = *reinterpret_cast<const float4*>(G.colsum + c);
                if (G.bias) bb[tl] = *reinterpret_cast<const float4*>(G.bias + c);
                if (G.res) r4[tl] = ld16_sys_v(G.res + (size_t)mc * G.ldo + c);
            }
        }
    }
    if (!cluster_barrier(cnt, target, &S.flag, err)) return false;
    if (!wg_active) return true;
    stage_rows<K>(G.X, m0, G.M, S.sX, wv, lane);
    __syncthreads();
    float s1 = 0.0f, s2 = 0.0f, mean = 0.0f, rstd = 1.0f;
    const float* xs = S.sX + l31 * kPXPitch + 8 * (wv * KUW) + 4 * half;
    float4 xa[KUW];
#pragma unroll
    for (int j = 0; j < KUW; ++j) {
        xa[j] = *reinterpret_cast<const float4*>(xs + 8 * j);
        if (LN) {
            const float x[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1 += x[e]; s2 = fmaf(x[e], x[e], s2); }
        }
    }
    if (LN) { S.sS1[wv * 2 + half][l31] = s1; S.sS2[wv * 2 + half][l31] = s2; }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int ct = ct0 + tl;
        const int c = ct * 32 + 8 * (wv & 3) + 4 * half;
        const bool mine = wv < 4 && ct < ntiles && m < G.M && c < G.Co;
        const size_t o = (size_t)mc * G.ldo + (c < G.Co ? c : 0);
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < KUW; ++j) {
            const float x[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
            const float w[4] = {wa[tl][j].x, wa[tl][j].y, wa[tl][j].z, wa[tl][j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], x[e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) S.sRed[wv][r][lane] = acc[r];
        __syncthreads();
        if (LN && tl == 0) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { t1 += S.sS1[j][l31]; t2 += S.sS2[j][l31]; }
            mean = t1 / (float)K;
            rstd = 1.0f / sqrtf(fmaxf(t2 / (float)K - mean * mean, 0.0f) + G.eps);
        }
        if (mine) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = 0.0f;
#pragma unroll
                for (int pz = 0; pz < 8; pz += 2) t += S.sRed[pz][4 * wv + e][lane] + S.sRed[pz + 1][4 * wv + e][lane];
                v[e] = t;
            }
            if (LN) {
                v[0] = rstd * (v[0] - mean * c4[tl].x); v[1] = rstd * (v[1] - mean * c4[tl].y);
                v[2] = rstd * (v[2] - mean * c4[tl].z); v[3] = rstd * (v[3] - mean * c4[tl].w);
            }
            if (G.bias) { v[0] += bb[tl].x; v[1] += bb[tl].y; v[2] += bb[tl].z; v[3] += bb[tl].w; }
            if (G.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
            }
            if (G.res) { v[0] += r4[tl][0]; v[1] += r4[tl][1]; v[2] += r4[tl][2]; v[3] += r4[tl][3]; }
            st16_x(G.out + o, hf32x4{v[0], v[1], v[2], v[3]}, G.plain);
        }
        if (tl + 1 < NT) __syncthreads();
    }
    drain_stores();
    return true;
}

// one entry point for both arithmetic forms (K = 128 KSW)
template <bool F32, int KSW, bool LN, int NT>
__device__ __forceinline__ bool pgemm(const PGemm& G, int tb, int ct0, PShared& S, int* cnt, int target, int* err, long long* sub = nullptr)
{
    if constexpr (F32) return pgemm_phase_f32<2 * KSW, LN, NT>(G, tb, ct0, S, cnt, target, err);
    else return pgemm_phase<KSW, LN, NT>(G, tb, ct0, S, cnt, target, err, sub);
}

// ---- the attention phases (their own functions: each gets its own register allocation) -------------------------------------
template <typename KV>
__device__ __forceinline__ bool pself_phase(const PStep& P, const PLayer& W, int step, int ab, int ah0, int lane, PShared& S, int* cnt, int target, bool plain)
{
    asm volatile("" : "+v"(lane));                         // (opaque per phase: see pgemm_phase)
    auto bar = [&]() { return cluster_barrier(cnt, target, &S.flag, P.err); };
    bool ok;
    if (ab < P.N)
        // both heads of the wavefront at once; the cluster barrier sits behind the cache requests (self_attend's `pre`)
        ok = self_attend<KV, true, 2>(P.qkv, P.C, P.N, P.H, step, P.Lmax, reinterpret_cast<KV*>(W.Kc), reinterpret_cast<KV*>(W.Vc),
                                      P.tokens, P.Lt, P.pad_idx, P.a, 0, ab, ah0, lane, bar, plain);
    else
        ok = bar();
    drain_stores();
    return ok;
}

// TBIG (round 6): the instantiation for more than 64 encoder tokens (one wavefront per head, four 64-token groups).  Compiled into
// the same kernel as the 64-token path it cost that path's kernel 60 spilled registers and ~1 % of a decode: two instantiations.
template <typename KV, bool TBIG>
__device__ __forceinline__ bool pcross_phase(const PStep& P, const PLayer& W, int ab, int ah0, int lane, PShared& S, int* cnt, int target, bool plain)
{
    asm volatile("" : "+v"(lane));                         // (opaque per phase: see pgemm_phase)
    typedef Wide<KV> Wd;
    constexpr int EPL = Wd::EPL, GS = kDK / EPL;
    auto bar = [&]() { return cluster_barrier(cnt, target, &S.flag, P.err); };
    if (ab >= P.N) return bar();
    int nvalid = P.valid_len ? P.valid_len[ab] : P.T;
    nvalid = nvalid < P.T ? nvalid : P.T;
    const float* qrow = P.qkv + (size_t)ab * 3 * P.C;     // (q: pitch 3 C, see the step kernel)
    bool ok = true;
    if constexpr (!TBIG) {
        ok = cross_attend2<KV, true>(qrow, reinterpret_cast<const KV*>(W.Kx), reinterpret_cast<const KV*>(W.Vx), P.C, P.T, nvalid, ab, ah0,
                                     lane, P.a, bar, plain);
    } else {
        ok = bar();
        const int dl = lane % GS;
#pragma unroll 1
        for (int i = 0; ok && i < P.pairs; ++i) {
            const int h = ah0 + i;
            float q[EPL];
#pragma unroll
            for (int e = 0; e < EPL / 4; ++e) {
                const hf32x4 rq = ld16_sys_v(qrow + kDK * h + EPL * dl + 4 * e);
#pragma unroll
                for (int j = 0; j < 4; ++j) q[4 * e + j] = rq[j] * 0.125f;
            }
            cross_attend<KV, true, 4>(q, reinterpret_cast<const KV*>(W.Kx), reinterpret_cast<const KV*>(W.Vx), P.C, P.N, P.T, nvalid,
                                      ab, h, lane, P.a, 0, plain);
        }
    }
    drain_stores();
    return ok;
}

// ---- the step ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld4_sys(const float* p)
{
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// KSW2 = d_inner / 128 (2 or 4).  grid = clusters x 16 workgroups of 512 threads; dynamic LDS = sizeof(PShared).
template <typename KV, int KSW2, bool F32, bool TBIG = false>
__global__ void __launch_bounds__(512)
dec_step_persist_kernel(const PStep P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
    PShared& S = *reinterpret_cast<PShared*>(p_smem);
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (cluster, workgroup in it): the 16 workgroups of a cluster are 16 blocks with the SAME blockIdx.x % 8, i.e.
    // -- as blocks are observed to be dealt to the XCDs round-robin -- on one XCD, whose L2 then carries the cluster's
    // exchanges (verified below, per launch: nothing depends on the placement for correctness).  Cluster c = 8 j + x is
    // blocks 8 (16 j + ct) + x; blocks of clusters >= nclusters leave.
    const int tb = ((blockIdx.x >> 7) << 3) + (blockIdx.x & 7), ct = (blockIdx.x >> 3) & 15;   // cluster (32 images), workgroup within it
    if (tb >= P.nclusters) return;
    int* cnt = P.counters + tb * 32;
    int bar = P.bar_base;
    const int N = P.N, C = P.C, H = P.H;
    const int tbg = (P.n0 >> 5) + tb;                         // the cluster's token block in the whole batch
    float *x = P.x, *y = P.y;
    // this wavefront's two (image, head) pairs in the attention phases: images 2 ct + (wv >> 2), heads 2 (wv & 3) + {0, 1}
    const int ab = tbg * 32 + 2 * ct + (wv >> 2);
    const int ah0 = 2 * (wv & 3);

    // Clusters are independent, so they need not run the same phase at the same time: with every cluster in its
    // cross-attention at once the memory system is saturated for 28 us (134 MB of fp32 keys / values per layer-step at
    // 4.8 TB/s) and idle during the latency-bound projections.  Odd clusters start half a layer-step late: one half of the
    // chip streams keys / values while the other half runs its projections.
    if (P.stagger > 0 && (tb & 1)) {
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < P.stagger) __builtin_amdgcn_s_sleep(8);
    }
    // A barrier timeout is made LOUD (include/tpspp.h, tpspp_nrtr_decoder_fwd): *err = 1 (the host hands it to the caller as
    // *status_out and the Python wrapper raises), and the scores of the workgroup's two images become NaN for EVERY step from the
    // failing one on -- no later step of `out` is left uninitialised -- before the workgroup leaves.
    auto fail_from = [&](int step0) {
        if (wv < 2) {
            const int b = tbg * 32 + 2 * ct + wv;
            if (b < N)
                for (int s = step0; s < P.Lsteps; ++s)
                    for (int c = lane; c < P.num_out; c += kWave) P.out[((size_t)b * P.Lsteps + s) * P.num_out + c] = __builtin_nanf("");
        }
    };
    if (tid == 0) S.timeout_k = P.timeout_k;
    // (one launch per step, TPSPP_HEAD_STEP_LAUNCHES: a decode that has failed in an earlier launch does not wait out a timeout
    // in each of the remaining ones)
    if (P.step > 0 && __hip_atomic_load(P.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { fail_from(P.step); return; }
    // ---- placement check (one extra cluster barrier per launch): every workgroup ORs the id of the XCD it runs on into the
    // cluster's mask; one bit set = the whole cluster shares an L2 and its exchanged data goes out as ORDINARY stores (the
    // line stays in that L2: st16_x), else as write-through stores as in the first version of this kernel.  Measured at
    // batch 512 (scripts/debug/bench_decoder_modes.py): fp32 -8 %, bf16x3 -8 %, bf16 -10 % per decode.
    bool plain;
    {
        if (tid == 0) {
            const unsigned id = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
            // (word 1 of the cluster's counters is zeroed once per decode, not per launch: with one launch per step a
            // misplacement seen once keeps the cluster on write-through stores for the rest of the decode -- conservative)
            __hip_atomic_fetch_or(cnt + 1, 1 << id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!cluster_barrier(cnt, 16 * (++bar), &S.flag, P.err)) { fail_from(P.step); return; }
        if (tid == 0) S.flag = __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wg_barrier_lds();
        const int mask = S.flag;                               // (the next barrier's flag write sits behind its own workgroup barrier)
        plain = __builtin_popcount((unsigned)mask) == 1 && !P.no_plain;
    }
    for (int step = P.step; step < P.step + P.nsteps; ++step) {
    int stamp_i = 0;
    auto stamp = [&]() {                                       // (diagnostics: end of a phase on this workgroup, before the next barrier)
        if (P.trace && tid == 0 && stamp_i < 64) P.trace[((size_t)step * gridDim.x + blockIdx.x) * 64 + stamp_i] = (long long)wall_clock64();
        ++stamp_i;
    };
    stamp();
    auto fail = [&]() { fail_from(step); };
    if (step == P.test_stall_step && tb == 0 && ct == 3) {     // test hook: see PStep::test_stall_step
        const long long t0 = (long long)wall_clock64(), lim = (long long)P.timeout_k << 11;
        while ((long long)wall_clock64() - t0 < lim) __builtin_amdgcn_s_sleep(32);
    }

    for (int l = 0; l < P.n_layers; ++l) {
        const PLayer& W = P.L[l];
        // 1. q | k | v = LN1(x) Wqkv                                              transformer_layers.py:150-151
        {
            const PGemm G{x, W.qkv_x, W.qkv_b, W.qkv_cs, nullptr, P.qkv, N, 3 * C, 1e-5f, 0, 3 * C, plain};
            if (!pgemm<F32, 4, true, 3>(G, tbg, 3 * ct, S, cnt, 16 * (++bar), P.err)) { fail(); return; }
        }
        stamp();
        // 2. cached self-attention -> a
        if (!pself_phase<KV>(P, W, step, ab, ah0, lane, S, cnt, 16 * (++bar), plain)) { fail(); return; }
        stamp();
        // 3. y = x + fc(a)                                                         transformer_layers.py:152-154
        {
            const PGemm G{P.a, W.wfc_x, W.bfc, nullptr, x, y, N, C, 0.0f, 0, C, plain};
            long long* sub = (P.trace && l == 2) ? P.trace + ((size_t)step * gridDim.x + blockIdx.x) * 64 + 50 : nullptr;
            if (!pgemm<F32, 4, false, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err, sub)) { fail(); return; }
        }
        stamp();
        // 4. q = LN2(y) Wq                                                         transformer_layers.py:156-157
        {
            // (q goes into the first C columns of the image's OWN q|k|v row, pitch 3 C: the launch pipeline packs q rows at pitch C
            // into the same buffer, which there is safe -- every image is past its self-attention -- and here would let one
            // cluster's q overwrite another cluster's q|k|v rows: clusters are not synchronised with each other)
            const PGemm G{y, W.q_x, W.q_b, W.q_cs, nullptr, P.qkv, N, C, 1e-5f, 0, 3 * C, plain};
            long long* sub = (P.trace && l == 2) ? P.trace + ((size_t)step * gridDim.x + blockIdx.x) * 64 + 56 : nullptr;
            if (!pgemm<F32, 4, true, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err, sub)) { fail(); return; }
        }
        stamp();
        // 5. cross-attention against the encoder keys / values -> a
        if (!pcross_phase<KV, TBIG>(P, W, ab, ah0, lane, S, cnt, 16 * (++bar), plain)) { fail(); return; }
        stamp();
        // 6. x = y + fc(a)                                                         transformer_layers.py:158-159
        {
            const PGemm G{P.a, W.wfc2_x, W.bfc2, nullptr, y, x, N, C, 0.0f, 0, C, plain};
            if (!pgemm<F32, 4, false, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err)) { fail(); return; }
        }
        stamp();
        // 7. hidden = gelu(LN3(x) W1 + b1)                                         transformer_layers.py:161-162
        {
            const PGemm G{x, W.w1_x, W.w1_b, W.w1_cs, nullptr, P.hid, N, P.d_inner, 1e-5f, 2, P.d_inner, plain};
            if (!pgemm<F32, 4, true, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err)) { fail(); return; }
        }
        stamp();
        // 8. y = x + W2 hidden + b2                                                transformer_layers.py:162-163
        {
            const PGemm G{P.hid, W.w2_x, W.b2, nullptr, x, y, N, C, 0.0f, 0, C, plain};
            if (!pgemm<F32, KSW2, false, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err)) { fail(); return; }
        }
        stamp();
        float* t = x; x = y; y = t;
    }
    // final LayerNorm (eps 1e-6) folded into the classifier                       nrtr_decoder.py:77,111 + :78
    {
        const PGemm G{x, P.cls_x, P.cls_b, P.cls_cs, nullptr, P.logits, N, P.num_out, 1e-6f, 0, P.num_out, plain};
        if (!pgemm<F32, 4, true, 1>(G, tbg, ct, S, cnt, 16 * (++bar), P.err)) { fail(); return; }
    }
    stamp();
    // soft-max / arg-max of the step, the next step's embedding row (dec_classify_kernel, one wavefront per image)
    if (!cluster_barrier(cnt, 16 * (++bar), &S.flag, P.err)) { fail(); return; }
    if (wv < 2) {
        const int b = tbg * 32 + 2 * ct + wv;
        if (b < N) {
            const int Cc = P.num_out;                          // (<= 128: two classes per lane)
            const float* lg = P.logits + (size_t)b * Cc;
            float v0 = -INFINITY, v1 = -INFINITY;
            if (lane < Cc) v0 = ld4_sys(lg + lane);
            if (lane + kWave < Cc) v1 = ld4_sys(lg + lane + kWave);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
            float* o = P.out + ((size_t)b * P.Lsteps + step) * Cc;
            const bool more = step + 1 < P.Lsteps;
            // the next step's embedding goes to y (which becomes the next launch's x: the host swaps as the launch path does)
            auto embed_next = [&](int tok) {
                if (more) {
                    const float* er = P.emb + (size_t)tok * C;
                    const float* pr = P.pos + (size_t)(step + 1) * C;
                    // (st16_x: an ordinary store when `plain` -- safe because the row's only reader is this cluster's next step,
                    // i.e. the SAME XCD inside the SAME launch (its L2 holds the line); without the placement guarantee, or with
                    // one launch per step on a cluster that was ever seen spread over XCDs, write-through: a plain store would
                    // leave the line in this XCD's L2 and a reader on another XCD could be served a stale copy)
                    for (int c = 4 * lane; c < C; c += 4 * kWave) {
                        const float4 e4 = *reinterpret_cast<const float4*>(er + c), p4 = *reinterpret_cast<const float4*>(pr + c);
                        st16_x(y + (size_t)b * C + c, hf32x4{e4.x + p4.x, e4.y + p4.y, e4.z + p4.z, e4.w + p4.w}, plain);
                    }
                }
            };
            if (!P.greedy) {
                if (lane < Cc) o[lane] = v0;
                if (lane + kWave < Cc) o[lane + kWave] = v1;
                embed_next(P.tokens[(size_t)b * P.Lt + step + 1]);
            } else {
                // first maximum: the same comparison sequence as dec_classify_kernel (ascending classes per lane, then the butterfly)
                float mx = -INFINITY;
                int am = 0x7fffffff;
                if (lane < Cc && v0 > mx) { mx = v0; am = lane; }
                if (lane + kWave < Cc && v1 > mx) { mx = v1; am = lane + kWave; }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(mx, off, kWave);
                    const int oi = __shfl_xor(am, off, kWave);
                    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
                }
                float sum = 0.0f;
                if (lane < Cc) sum += expf(v0 - mx);
                if (lane + kWave < Cc) sum += expf(v1 - mx);
                sum = wave_sum(sum);
                if (lane < Cc) o[lane] = expf(v0 - mx) / sum;
                if (lane + kWave < Cc) o[lane + kWave] = expf(v1 - mx) / sum;
                // (system scope: the next step of THIS launch reads it in the self-attention's <PAD> mask)
                if (lane == 0) __hip_atomic_store(P.tokens + (size_t)b * P.Lt + step + 1, am, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                embed_next(am);
            }
        }
    }
    // the next step of this launch: its first projection stages y's rows behind its cluster barrier
    drain_stores();
    { float* t = x; x = y; y = t; }
    }
}
