// Plane-streaming TPS warp for gfx950: the TPS_PP geometry (many channels, one or two inputs, optional
// attention score), where an image does not fit in LDS but ONE CHANNEL PLANE of each input does.
//
// One workgroup rectifies one image.  Work is split between specialised wavefronts:
//   * kLoaders loader wavefronts stream the channel planes HBM -> LDS with global_load_lds_dwordx4
//     (1 KB per instruction, no register round trip) into a ring of kRing stage slots, always two
//     stages ahead of the consumers; each input byte crosses HBM -> CU exactly once, fully coalesced;
//   * the compute wavefronts first build the sampling grid of their pixels ONCE per image
//     (T-solve by wavefront 0, then one k-ascending FMA chain pair per pixel with the attention
//     score folded in as the reference does: rbf * (score*0.5 + 1), three separately rounded ops),
//     derive the bilinear taps of both inputs once, and then, per stage, read 4 taps per input from
//     LDS and write one fully coalesced output row segment per input.
// One s_barrier per stage hands a landed slot to the consumers and a drained slot back to the
// loaders; the loaders use COUNTED vmcnt waits so that the next stage stays in flight across the
// barrier (a vmcnt(0) there would serialise fetch and compute).
//
// Bound: HBM bandwidth.  Algorithmic bytes per image at the TPS_PP defaults:
// 1,048,576 + 262,144 in, 131,072 score, 256 control points, 2 x 262,144 out = 1,966,336.
//
// Replaces: Attention_Enhanced_TPS.build_P_prime + 2x F.grid_sample, backbones/tps_pp/tps_pp.py:597-615
// (also serves the classic layout for inputs too large for the image-pair kernel).
#include "tpspp_common.h"
#include "tpspp_warp_dev.h"
#include "tpspp_warp_stream.h"

#include <type_traits>

using namespace tpspp_dev;

namespace {

constexpr int kComputeWaves = 8;
constexpr int kLoaders = 4;
constexpr int kRing = 3;
constexpr int kCPS32 = 2;                                   // channels of each input per stage (fp32 planes)
constexpr int kCPS16 = 4;                                   // ... for bf16 planes (same bytes per stage)
constexpr int kThreads = (kComputeWaves + kLoaders) * kWave;
constexpr int kCT = kComputeWaves * kWave;                 // pixel stride between a thread's slots

struct StreamParams {
    const float* in0; int C0, H0, W0;
    const float* in1; int C1, H1, W1;
    const float* ctrl; const float* score; const float* inv_delta_c;
    const float* p_hat; int p_hat_ld; const float* p_xy; const float* p_hat_t;
    int N, n;
    int score_t;               // 1: score is (N, F, n)
    float* out0; float* out1; float* grid; int32_t* idx;
    int pieces0, pieces1;      // 1-KB DMA pieces per plane of in0 / in1 (in1: 0 when absent)
    int per_loader;            // pieces every loader issues per stage (stage padded to a multiple)
    int slot_bytes;            // LDS bytes of one ring slot
    long long* trace;
};

// s_waitcnt vmcnt(N) needs an immediate: N = pieces of the one stage allowed to stay in flight
__device__ __forceinline__ void wait_vmcnt(int n)
{
#define TPSPP_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    switch (n) {
        TPSPP_W(0) TPSPP_W(1) TPSPP_W(2) TPSPP_W(3) TPSPP_W(4) TPSPP_W(5) TPSPP_W(6) TPSPP_W(7)
        TPSPP_W(8) TPSPP_W(9) TPSPP_W(10) TPSPP_W(11) TPSPP_W(12) TPSPP_W(13) TPSPP_W(14) TPSPP_W(15)
        TPSPP_W(16) TPSPP_W(17) TPSPP_W(18) TPSPP_W(19) TPSPP_W(20) TPSPP_W(21) TPSPP_W(22) TPSPP_W(23)
        TPSPP_W(24) TPSPP_W(25) TPSPP_W(26) TPSPP_W(27) TPSPP_W(28) TPSPP_W(29) TPSPP_W(30) TPSPP_W(31)
        TPSPP_W(32)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef TPSPP_W
}

struct TapRegs {               // per pixel, per input: everything the per-channel loop needs
    int o00, o10;              // float offsets inside a plane (o10 already clamped to the last row)
    float nw, ne, sw, se;
    bool inx, iny;
};

__device__ __forceinline__ TapRegs to_regs(const Taps& t)
{
    TapRegs r;
    r.o00 = t.o00; r.o10 = t.o10;
    r.nw = t.nw; r.ne = t.ne; r.sw = t.sw; r.se = t.se;
    r.inx = t.inx; r.iny = t.iny;
    return r;
}

__device__ __forceinline__ float lds_elem(const float* pl, int o) { return pl[o]; }
__device__ __forceinline__ float lds_elem(const unsigned short* pl, int o)
{
    return __builtin_bit_cast(float, (unsigned)pl[o] << 16);          // bf16 -> fp32, exact
}

__device__ __forceinline__ void store_elem(char* base, unsigned pix, float v, float*)
{
    *reinterpret_cast<float*>(base + 4u * pix) = v;
}
__device__ __forceinline__ void store_elem(char* base, unsigned pix, float v, unsigned short*)
{
    unsigned u = __builtin_bit_cast(unsigned, v);                       // round to nearest even
    u += 0x7fffu + ((u >> 16) & 1u);
    *reinterpret_cast<unsigned short*>(base + 2u * pix) = (unsigned short)(u >> 16);
}

template <typename T>
__device__ __forceinline__ float lds_bilerp(const T* pl, const TapRegs& t)
{
    // the east neighbour is read unconditionally (slots end with slack) and masked afterwards
    const float v00 = lds_elem(pl, t.o00);
    float v01 = lds_elem(pl, t.o00 + 1);
    float v10 = lds_elem(pl, t.o10);
    float v11 = lds_elem(pl, t.o10 + 1);
    v01 = t.inx ? v01 : 0.0f;
    v10 = t.iny ? v10 : 0.0f;
    v11 = (t.inx && t.iny) ? v11 : 0.0f;
    float acc = v00 * t.nw;
    acc = fmaf(v01, t.ne, acc);
    acc = fmaf(v10, t.sw, acc);
    acc = fmaf(v11, t.se, acc);
    return acc;
}

// FCT: F at compile time (table row in registers); PPT: pixels per compute thread; B16: the inputs and
// outputs are bf16 in memory (the bf16 configuration: half the bytes, interpolation still in fp32 with the
// fp32 grid, one rounding at the store)
template <int FCT, bool PXY, bool SCORE, int PPT, bool B16 = false>
__global__ void __launch_bounds__(kThreads)
tps_warp_stream_kernel(const StreamParams P)
{
    constexpr int kCPS = B16 ? kCPS16 : kCPS32;
    constexpr int ES = B16 ? 2 : 4;                           // bytes per element of in0 / in1 / out0 / out1
    using elem_t = typename std::conditional<B16, unsigned short, float>::type;
    constexpr int F = FCT;
    constexpr int K = F + 3;
    constexpr int KK = K * K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [ sT: K x float2 (padded to 4 floats) | sInv: K*K (padded) | ring: kRing slots ]
    float2* sT = reinterpret_cast<float2*>(smem);
    float* sInv = smem + ((2 * K + 3) & ~3);
    char* ring = reinterpret_cast<char*>(sInv + ((KK + 3) & ~3));

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int b = blockIdx.x;
    const int HW0 = P.H0 * P.W0, HW1 = P.H1 * P.W1;
    const int stages = (max(P.C0, P.C1) + kCPS - 1) / kCPS;   // kCPS channels of each input per stage

    if (wv == 0) stamp(P.trace, 0);
    if (wv >= kComputeWaves) {
        // ================= loader wavefronts =================
        const int ld = wv - kComputeWaves;
        const char* base0 = reinterpret_cast<const char*>(P.in0);
        const char* base1 = reinterpret_cast<const char*>(P.in1);
        const long long end0 = (long long)P.N * P.C0 * HW0 * ES;
        const long long end1 = (long long)P.N * P.C1 * HW1 * ES;
        const int per_ch = P.pieces0 + P.pieces1;           // slot layout: [in0 c | in1 c] x kCPS
        const int total = per_ch * kCPS;
        auto issue_stage = [&](int s) {
            char* slot = ring + (s % kRing) * P.slot_bytes;
            for (int i = 0; i < P.per_loader; ++i) {
                int piece = ld + i * kLoaders;
                if (piece >= total) piece = total - 1;      // padding: repeat the last piece
                const int sub = piece / per_ch;             // which channel of the stage
                const int q = piece - sub * per_ch;
                const bool second = q >= P.pieces0;
                const int pp = second ? q - P.pieces0 : q;
                const int C = second ? P.C1 : P.C0;
                const int HW = second ? HW1 : HW0;
                const int ch = s * kCPS + sub;
                const int c = ch < C ? ch : C - 1;         // an exhausted input re-reads its last plane
                long long off = ((long long)b * C + c) * HW * ES + (long long)pp * 1024 + lane * 16;
                const long long end = second ? end1 : end0;
                if (off + 16 > end) off = end - 16;        // tail of the tensor: stay inside it
                const char* src = (second ? base1 : base0) + off;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
            }
        };
        // run kRing-1 stages ahead of the consumers
        for (int s = 0; s < kRing - 1 && s < stages; ++s) issue_stage(s);
        lds_only_barrier();                                 // T barrier of the compute wavefronts
        for (int s = 0; s < stages; ++s) {
            // stage s must have landed; the stages issued after it may stay in flight
            const int ahead = min(stages - 1 - s, kRing - 2);
            wait_vmcnt(ahead * P.per_loader);
            lds_only_barrier();                             // A(s): slot s%R ready, slot (s-1)%R drained
            if (s + kRing - 1 < stages) issue_stage(s + kRing - 1);
        }
        return;
    }

    // ================= compute wavefronts =================
    // ---- T-solve inputs (wavefront 0): inv_delta_C read coalesced, one row per lane via LDS ----
    constexpr int NINV = (KK + kWave - 1) / kWave;
    float invv[NINV];
    float cx = 0.0f, cy = 0.0f;
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            invv[i] = P.inv_delta_c[e < KK ? e : KK - 1];
        }
        if (lane < F) {
            const float2 cc = reinterpret_cast<const float2*>(P.ctrl + (size_t)b * F * 2)[lane];
            cx = cc.x; cy = cc.y;
        }
    }

    // ---- table rows (and affine part) of this thread's pixels ----
    int pix[PPT];
    bool live[PPT];
    float rbf[PPT][F];
    float r0[PPT], r1[PPT], r2[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = tid + j * kCT;
        live[j] = p < P.n;
        pix[j] = live[j] ? p : P.n - 1;
        if (PXY) {
            const float2 xy = reinterpret_cast<const float2*>(P.p_xy)[pix[j]];
            r0[j] = 1.0f; r1[j] = xy.x; r2[j] = xy.y;
            if (P.p_hat_t) {
#pragma unroll
                for (int k = 0; k < F; ++k) rbf[j][k] = P.p_hat_t[(size_t)k * P.n + pix[j]];
            } else {
#pragma unroll
                for (int k = 0; k < F; ++k) rbf[j][k] = P.p_hat[(size_t)pix[j] * P.p_hat_ld + k];
            }
        } else {
            if (P.p_hat_t) {
                r0[j] = P.p_hat_t[pix[j]]; r1[j] = P.p_hat_t[(size_t)P.n + pix[j]];
                r2[j] = P.p_hat_t[2 * (size_t)P.n + pix[j]];
#pragma unroll
                for (int k = 0; k < F; ++k) rbf[j][k] = P.p_hat_t[(size_t)(3 + k) * P.n + pix[j]];
            } else {
                const float* ph = P.p_hat + (size_t)pix[j] * P.p_hat_ld;
                r0[j] = ph[0]; r1[j] = ph[1]; r2[j] = ph[2];
#pragma unroll
                for (int k = 0; k < F; ++k) rbf[j][k] = ph[3 + k];
            }
        }
    }

    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NINV; ++i) {
            const int e = lane + i * kWave;
            if (e < KK) sInv[e] = invv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* hrow = sInv + (lane < K ? lane : K - 1) * K;
        float ax = 0.0f, ay = 0.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float hv = hrow[q];
            ax = fmaf(hv, readlane_f(cx, q), ax);
            ay = fmaf(hv, readlane_f(cy, q), ay);
        }
        if (lane < K) sT[lane] = make_float2(ax, ay);
    }
    lds_only_barrier();
    if (wv == 0) stamp(P.trace, 1);                               // T ready

    // ---- sampling grid: k-ascending FMA chains; score folded in as mul, add, mul (tps_pp.py:474) ----
    float gx[PPT], gy[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        float ax = 0.0f, ay = 0.0f;
        {
            const float2 t0 = sT[0], t1 = sT[1], t2 = sT[2];
            ax = fmaf(r0[j], t0.x, ax); ay = fmaf(r0[j], t0.y, ay);
            ax = fmaf(r1[j], t1.x, ax); ay = fmaf(r1[j], t1.y, ay);
            ax = fmaf(r2[j], t2.x, ax); ay = fmaf(r2[j], t2.y, ay);
        }
        const float* srow = SCORE ? P.score + ((size_t)b * P.n + pix[j]) * F : nullptr;
        if (SCORE && P.score_t) {
            // (N, F, n) layout: one coalesced 256-B row segment per k across the wavefront
            const float* scol = P.score + (size_t)b * F * P.n + pix[j];
#pragma unroll
            for (int k = 0; k < F; ++k) {
                float gq = scol[(size_t)k * P.n] * 0.5f;
                gq = gq + 1.0f;
                const float m = rbf[j][k] * gq;
                const float2 tk = sT[3 + k];
                ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
            }
        } else if (SCORE && (F % 4 == 0)) {
#pragma unroll
            for (int k4 = 0; k4 < F / 4; ++k4) {
                const float4 s4 = reinterpret_cast<const float4*>(srow)[k4];
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k4 * 4 + i;
                    float gq = sv[i] * 0.5f;
                    gq = gq + 1.0f;
                    const float m = rbf[j][k] * gq;
                    const float2 tk = sT[3 + k];
                    ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < F; ++k) {
                float m = rbf[j][k];
                if (SCORE) {
                    float gq = srow[k] * 0.5f;
                    gq = gq + 1.0f;
                    m = m * gq;
                }
                const float2 tk = sT[3 + k];
                ax = fmaf(m, tk.x, ax); ay = fmaf(m, tk.y, ay);
            }
        }
        gx[j] = ax; gy[j] = ay;
    }

    // pin the finished grid here (keeps the optimiser from sinking the chains into the stream loop)
#pragma unroll
    for (int j = 0; j < PPT; ++j) asm volatile("" : "+v"(gx[j]), "+v"(gy[j]));

    // ---- taps of both inputs, once per image ----
    TapRegs t0[PPT], t1[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const Taps a = make_taps(gx[j], gy[j], P.H0, P.W0);
        t0[j] = to_regs(a);
        if (live[j]) {
            if (P.grid) reinterpret_cast<float2*>(P.grid)[(size_t)b * P.n + pix[j]] = make_float2(gx[j], gy[j]);
            if (P.idx) reinterpret_cast<int2*>(P.idx)[(size_t)b * P.n + pix[j]] = make_int2(a.x0, a.y0);
        }
        t1[j] = to_regs(make_taps(gx[j], gy[j], P.H1, P.W1));
    }
    if (wv == 0) stamp(P.trace, 2);                               // grid + taps done

    // ---- stream the channel planes ----
    const int off1 = P.pieces0 * 1024;                            // in1's plane inside a slot
    char* o0 = reinterpret_cast<char*>(P.out0) + (size_t)b * P.C0 * P.n * ES;
    char* o1 = P.out1 ? reinterpret_cast<char*>(P.out1) + (size_t)b * P.C1 * P.n * ES : nullptr;
    const size_t row_bytes = (size_t)P.n * ES;
    const int ch_bytes = (P.pieces0 + P.pieces1) * 1024;
    for (int s = 0; s < stages; ++s) {
        lds_only_barrier();                                       // A(s)
        const char* slot = ring + (s % kRing) * P.slot_bytes;
#pragma unroll
        for (int sub = 0; sub < kCPS; ++sub) {
            const int ch = s * kCPS + sub;
            const elem_t* pl0 = reinterpret_cast<const elem_t*>(slot + sub * ch_bytes);
            const elem_t* pl1 = reinterpret_cast<const elem_t*>(slot + sub * ch_bytes + off1);
            if (ch < P.C0) {
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const float r = lds_bilerp(pl0, t0[j]);
                    if (live[j]) store_elem(o0 + ch * row_bytes, (unsigned)pix[j], r, (elem_t*)nullptr);
                }
            }
            if (P.in1 && ch < P.C1) {
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const float r = lds_bilerp(pl1, t1[j]);
                    if (live[j]) store_elem(o1 + ch * row_bytes, (unsigned)pix[j], r, (elem_t*)nullptr);
                }
            }
        }
    }
    if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(P.trace, 4);
    }
}

template <int F, bool PXY, bool SCORE, bool B16>
void launch_ppt(const StreamParams& P, int ppt, size_t lds, hipStream_t st)
{
    const dim3 grid((unsigned)P.N), block(kThreads);
#define TPSPP_LAUNCH(PP)                                                                            \
    {                                                                                               \
        static bool attr_done[tpspp::kMaxDevices] = {};                                                              \
        if (tpspp::first_use_on_device(attr_done)) {                                                                           \
            (void)hipFuncSetAttribute(                                                              \
                reinterpret_cast<const void*>(&tps_warp_stream_kernel<F, PXY, SCORE, PP, B16>),          \
                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
            (void)hipGetLastError();                                                                \
        }                                                                                           \
        hipLaunchKernelGGL((tps_warp_stream_kernel<F, PXY, SCORE, PP, B16>), grid, block, lds, st, P);   \
    }
    if (ppt == 1) TPSPP_LAUNCH(1)
    else TPSPP_LAUNCH(2)
#undef TPSPP_LAUNCH
}

template <int F, bool B16>
void launch_f(const StreamParams& P, int ppt, size_t lds, hipStream_t st)
{
    const bool pxy = P.p_xy != nullptr, sc = P.score != nullptr;
    if (pxy && sc)       launch_ppt<F, true, true, B16>(P, ppt, lds, st);
    else if (pxy && !sc) launch_ppt<F, true, false, B16>(P, ppt, lds, st);
    else if (!pxy && sc) launch_ppt<F, false, true, B16>(P, ppt, lds, st);
    else                 launch_ppt<F, false, false, B16>(P, ppt, lds, st);
}

}  // namespace

namespace tpspp {

bool stream_kernel_applicable(const StreamArgs& a)
{
    if (a.F != 20 && a.F != 32) return false;
    const int n = a.Ho * a.Wo;
    const int ppt = (n + kCT - 1) / kCT;
    if (ppt < 1 || ppt > 2) return false;      // 3 pixels per thread spills (168-VGPR budget)
    if (reinterpret_cast<uintptr_t>(a.in0) % 16 || (a.in1 && reinterpret_cast<uintptr_t>(a.in1) % 16)) return false;
    const int es = a.io_bf16 ? 2 : 4, cps = a.io_bf16 ? kCPS16 : kCPS32;
    if ((a.H0 * a.W0 * es) % 16 || (a.in1 && (a.H1 * a.W1 * es) % 16)) return false;   // 16-B DMA granules
    if ((long long)a.N * a.C0 * a.H0 * a.W0 * es < 1024) return false;
    if (a.in1 && (long long)a.N * a.C1 * a.H1 * a.W1 * es < 1024) return false;
    const int p0 = (a.H0 * a.W0 * es + 1023) / 1024;
    const int p1 = a.in1 ? (a.H1 * a.W1 * es + 1023) / 1024 : 0;
    const int per_loader = ((p0 + p1) * cps + kLoaders - 1) / kLoaders;
    if (per_loader * (kRing - 2) > 32) return false;                            // counted vmcnt range
    const int K = a.F + 3;
    const size_t slot = (size_t)per_loader * kLoaders * 1024 + 16;
    const size_t lds = (size_t)(((2 * K + 3) & ~3) + ((K * K + 3) & ~3)) * 4 + kRing * slot;
    return lds <= 160 * 1024;
}

int launch_stream_kernel(const StreamArgs& a, long long* trace, hipStream_t st)
{
    StreamParams P;
    P.in0 = a.in0; P.C0 = a.C0; P.H0 = a.H0; P.W0 = a.W0;
    P.in1 = a.in1; P.C1 = a.in1 ? a.C1 : 0; P.H1 = a.in1 ? a.H1 : 1; P.W1 = a.in1 ? a.W1 : 1;
    P.ctrl = a.ctrl; P.score = a.score; P.inv_delta_c = a.inv_delta_c;
    P.p_hat = a.p_hat; P.p_hat_ld = a.p_hat_ld; P.p_xy = a.p_xy; P.p_hat_t = a.p_hat_t;
    P.N = a.N; P.n = a.Ho * a.Wo; P.score_t = a.score_t;
    P.out0 = a.out0; P.out1 = a.out1; P.grid = a.grid; P.idx = a.idx;
    const int es = a.io_bf16 ? 2 : 4, cps = a.io_bf16 ? kCPS16 : kCPS32;
    P.pieces0 = (a.H0 * a.W0 * es + 1023) / 1024;
    P.pieces1 = a.in1 ? (a.H1 * a.W1 * es + 1023) / 1024 : 0;
    P.per_loader = ((P.pieces0 + P.pieces1) * cps + kLoaders - 1) / kLoaders;
    P.slot_bytes = P.per_loader * kLoaders * 1024 + 16;
    P.trace = trace;
    const int K = a.F + 3;
    const size_t lds = (size_t)(((2 * K + 3) & ~3) + ((K * K + 3) & ~3)) * 4 + (size_t)kRing * P.slot_bytes;
    const int ppt = (P.n + kCT - 1) / kCT;
    if (a.io_bf16) {
        if (a.F == 20) launch_f<20, true>(P, ppt, lds, st);
        else           launch_f<32, true>(P, ppt, lds, st);
    } else {
        if (a.F == 20) launch_f<20, false>(P, ppt, lds, st);
        else           launch_f<32, false>(P, ppt, lds, st);
    }
    return check_launch("tpspp_warp_fwd(stream)");
}

}  // namespace tpspp
