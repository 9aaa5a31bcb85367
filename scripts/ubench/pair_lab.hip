// Kernel lab, round 4: variants of the image-pair kernel (tpspp_warp_pair.h, template parameter VAR) against the library's
// build of it and against plain copies of the same bytes with the same launch shape, on one / two / three streams.
// BASELINE configs[1]: 512 x 3x32x100 fp32, F = 20; 14 rotating buffer sets (550 MB > the 256 MB Infinity Cache).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I tps_pp_amd/csrc -I include \
//         scripts/ubench/pair_lab.hip -o scripts/ubench/pair_lab -ldl
//   scripts/ubench/pair_lab [consts.bin] [libtpspp_hip.so] [iters] [only]
// Every variant is checked bit for bit against the library's result (grid and tap indices included) before it is timed;
// the table at the end re-times all of them interleaved, so box / clock drift hits every row alike.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "tpspp_warp_pair.h"

using namespace tpspp_pair;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int (*warp_fwd_t)(const float*, int, int, int, const float*, int, int, int, const float*, const float*,
                          const float*, const float*, int, const float*, const float*, int, int, int, int, int,
                          float*, float*, float*, int32_t*, void*);

static const int F = 20, K = 23, C = 3, H = 32, W = 100, n = H * W;
static int N = 512;
static const int SETS = 14;

struct Bufs {
    float* in[SETS]; float* ctrl[SETS]; float* out[SETS];
    float* inv; float* p_hat; float* packed[3];   // packed[QP]
    float* ref[2]; float* refgrid; int32_t* refidx; float* grid; int32_t* idx;
    long long* trace;
};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

// ---- copies with the kernel's launch shape: one workgroup = one image's 38,400 bytes -----------------------------
// MODE 0: registers (16-byte loads, nt stores); MODE 1: the kernel's own data path without the arithmetic -- one
// loader wavefront, LDS-DMA nt, flag, flat 16-byte nt stores from LDS by the other wavefronts
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) copy_img_k(const float* __restrict__ in, float* __restrict__ out, int img_bytes, int late_from)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* src = reinterpret_cast<const char*>(in) + (size_t)blockIdx.x * img_bytes;
    gchar* dst = (gchar*)out + (size_t)blockIdx.x * img_bytes;
    const int n16 = img_bytes >> 4;
    if (MODE == 0) {
        for (int e = tid; e < n16; e += THREADS) {
            const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src) + e);
            store16_nt(dst + 16u * (unsigned)e, x);
        }
        return;
    }
    constexpr int NWC = THREADS / 64 - 1;
    float* sFlag = sm;
    float* sImg = sm + 4;
    if (tid == 0) reinterpret_cast<int*>(sFlag)[0] = 0;
    lds_only_barrier();
    if (wv == NWC) {
        unsigned fl = (unsigned)(size_t)sFlag; int one = 1;
        asm volatile("" : "+v"(fl), "+v"(one));
        if ((int)blockIdx.x >= late_from) __builtin_amdgcn_s_sleep(127);
        const int pieces = (img_bytes + 1023) >> 10;
        for (int k = 0; k < pieces; ++k) {
            const int off = k * 1024 + lane * 16;
            if (off < img_bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sImg) + k * 1024), 16, 0, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fl), "v"(one) : "memory");
        return;
    }
    wait_flag(sFlag, 1);
    for (int e = tid; e < n16; e += NWC * 64) {
        const v4f x = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sImg) + 16 * e);
        store16_nt(dst + 16u * (unsigned)e, x);
    }
}

__global__ void __launch_bounds__(256) copy_flat_k(const v4f* __restrict__ src, char* dst, int n4)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x)
        store16_nt((gchar*)dst + 16u * (unsigned)i, __builtin_nontemporal_load(src + i));
}

// ---- timing -------------------------------------------------------------------------------------------------------
static hipStream_t g_st[4];

// `streams` launches in flight side by side: launch i goes to stream i % streams
template <class L> static float period_us(L launch, int iters, int streams, int warm = 30)
{
    hipEvent_t e0, e1[4]; CK(hipEventCreate(&e0));
    for (int s = 0; s < streams; ++s) CK(hipEventCreate(&e1[s]));
    for (int i = 0; i < warm; ++i) launch(i, g_st[i % streams]);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, g_st[0]));
    for (int s = 1; s < streams; ++s) CK(hipStreamWaitEvent(g_st[s], e0, 0));   // nobody starts before the clock does
    for (int i = 0; i < iters; ++i) launch(i, g_st[i % streams]);
    for (int s = 0; s < streams; ++s) CK(hipEventRecord(e1[s], g_st[s]));
    float worst = 0.0f;
    for (int s = 0; s < streams; ++s) {
        CK(hipEventSynchronize(e1[s]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1[s]));
        worst = std::max(worst, ms);
    }
    CK(hipEventDestroy(e0));
    for (int s = 0; s < streams; ++s) CK(hipEventDestroy(e1[s]));
    return worst * 1e3f / iters;
}

struct Variant {
    std::string name;
    std::function<void(const Bufs&, int, float*, float*, int32_t*, long long*, hipStream_t)> run;
    bool checked;
    int imgs = 1;      // images per workgroup (trace report)
};

static const float* g_p_hat = nullptr;

template <int VAR, bool AUX, bool TRACE>
static void launch_pair_var(const float* in, const float* ctrl, const float* inv, const float* packed, int Nn, float* out,
                            float* grid, int32_t* idx, long long* trace, hipStream_t st)
{
    PairParams P;
    P.in = in; P.ctrl = ctrl; P.inv_delta_c = inv; P.packed = packed; P.N = Nn;
    P.p_hat = g_p_hat; P.p_hat_ld = K;
    P.out = out; P.grid = grid; P.idx = idx; P.trace = trace;
    const size_t lds = pair_lds_bytes<F, C, H, W, H, W>(&P.zero_off, &P.out_off);
    auto kern = tps_warp_pair_kernel<F, C, H, W, H, W, AUX, TRACE, VAR>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
    hipLaunchKernelGGL(kern, dim3((Nn + 1) / 2), dim3((PairGeo<H, W>::NW + kPairLoaders) * 64), lds, st, P);
}

static const float* g_packed = nullptr;

template <int VAR>
static Variant pair_variant(const char* label)
{
    char nm[128]; snprintf(nm, 128, "pair VAR=%d %s", VAR, label);
    return Variant{nm, [](const Bufs& Bf, int s, float* o, float* g, int32_t* ix, long long* tr, hipStream_t st) {
        if (tr) launch_pair_var<VAR, false, true>(Bf.in[s], Bf.ctrl[s], Bf.inv, g_packed, N, o, g, ix, tr, st);
        else if (g || ix) launch_pair_var<VAR, true, false>(Bf.in[s], Bf.ctrl[s], Bf.inv, g_packed, N, o, g, ix, tr, st);
        else launch_pair_var<VAR, false, false>(Bf.in[s], Bf.ctrl[s], Bf.inv, g_packed, N, o, g, ix, tr, st); }, true, 2};
}

// the pair kernel's stamps: t[0..6] = shader-clock ticks since entry (T ready, grid + descriptors, A landed, A staged,
// B landed, B staged, stores retired), t[7] = chip-wide 100 MHz clock at entry
static void trace_report(const Bufs& B, int nblocks, int launches)
{
    std::vector<long long> t((size_t)launches * nblocks * 8);
    CK(hipMemcpy(t.data(), B.trace, t.size() * 8, hipMemcpyDeviceToHost));
    auto pct = [](std::vector<double> v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    const char* nm[7] = {"T ready", "grid+desc", "A landed", "A staged", "B landed", "B staged", "retired"};
    std::vector<double> ph[7];
    for (int b = 0; b < nblocks; ++b) {
        const long long* s = &t[((size_t)(launches - 1) * nblocks + b) * 8];
        for (int i = 0; i < 7; ++i) ph[i].push_back(s[i] / 2400.0);   // shader-clock ticks, at a nominal 2.4 GHz
    }
    printf("      us since entry (ticks / 2400), p10/med/p90:");
    for (int i = 0; i < 7; ++i) printf(" %s %.2f/%.2f/%.2f |", nm[i], pct(ph[i], .1), pct(ph[i], .5), pct(ph[i], .9));
    printf("\n");
    for (int l = std::max(0, launches - 3); l < launches; ++l) {
        long long s0 = 1LL << 62, s1 = 0, e0 = 1LL << 62, e1 = 0;
        for (int b = 0; b < nblocks; ++b) {
            const long long* s = &t[((size_t)l * nblocks + b) * 8];
            const long long st = s[7], en = s[7] + s[6] / 24;
            s0 = std::min(s0, st); s1 = std::max(s1, st); e0 = std::min(e0, en); e1 = std::max(e1, en);
        }
        static long long prev_e1 = 0, prev_s0 = 0;
        printf("      launch %d: starts spread %.2f, first start -> first end %.2f, -> last end %.2f us", l, (s1 - s0) / 100.0, (e0 - s0) / 100.0, (e1 - s0) / 100.0);
        if (l > std::max(0, launches - 3)) printf("; gap after previous launch's last end %.2f us, start-to-start %.2f us", (s0 - prev_e1) / 100.0, (s0 - prev_s0) / 100.0);
        printf("\n");
        prev_e1 = e1; prev_s0 = s0;
    }
}

int main(int argc, char** argv)
{
    const char* consts = argc > 1 ? argv[1] : "scripts/ubench/warp_lab_consts.bin";
    const char* libpath = argc > 2 ? argv[2] : "tps_pp_amd/libtpspp_hip.so";
    const int iters = argc > 3 ? atoi(argv[3]) : 1000;
    const char* only = argc > 4 ? argv[4] : "";
    std::vector<float> hinv(K * K), hphat((size_t)n * K), hident(F * 2);
    {
        FILE* f = fopen(consts, "rb");
        if (!f) { printf("cannot open %s\n", consts); return 1; }
        if (fread(hinv.data(), 4, hinv.size(), f) != hinv.size() || fread(hphat.data(), 4, hphat.size(), f) != hphat.size() ||
            fread(hident.data(), 4, hident.size(), f) != hident.size()) { printf("short consts file\n"); return 1; }
        fclose(f);
    }
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    warp_fwd_t warp_fwd = (warp_fwd_t)dlsym(lib, "tpspp_warp_fwd");
    auto set_tuning = (int (*)(int, int, int, int))dlsym(lib, "tpspp_warp_set_tuning");
    auto prep_floats = (size_t (*)(int, int, int))dlsym(lib, "tpspp_prepared_table_floats");
    auto prep = (int (*)(const float*, int, int, int, int, float*, void*))dlsym(lib, "tpspp_prepare_mirror_table");
    if (!warp_fwd || !set_tuning || !prep_floats || !prep) { printf("symbols missing\n"); return 1; }
    for (int s = 0; s < 4; ++s) CK(hipStreamCreateWithFlags(&g_st[s], hipStreamNonBlocking));

    Bufs B;
    const size_t img_bytes = (size_t)N * C * n * 4, ctrl_bytes = (size_t)N * F * 2 * 4;
    uint32_t seed = 12345;
    std::vector<float> himg((size_t)N * C * n), hctrl((size_t)N * F * 2);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&B.in[s], img_bytes)); CK(hipMalloc(&B.out[s], img_bytes)); CK(hipMalloc(&B.ctrl[s], ctrl_bytes));
        for (auto& x : himg) x = (float)(lcg(seed) >> 8) / 8388608.0f - 1.0f;
        for (size_t i = 0; i < hctrl.size(); ++i)
            hctrl[i] = hident[i % (F * 2)] + 0.05f * ((float)(lcg(seed) >> 8) / 8388608.0f - 1.0f);
        if (s == 1) {   // a nasty set: large perturbations (clamped / out-of-image taps), specials in the image
            for (size_t i = 0; i < hctrl.size(); ++i) hctrl[i] = hident[i % (F * 2)] + 0.8f * ((float)(lcg(seed) >> 8) / 8388608.0f - 1.0f);
            for (size_t i = 0; i < himg.size(); i += 997) himg[i] = -0.0f;
            for (size_t i = 5; i < himg.size(); i += 7919) himg[i] = (i & 1) ? __builtin_inff() : -__builtin_inff();
            for (size_t i = 11; i < himg.size(); i += 10007) himg[i] = __builtin_nanf("");
        }
        CK(hipMemcpy(B.in[s], himg.data(), img_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.ctrl[s], hctrl.data(), ctrl_bytes, hipMemcpyHostToDevice));
        CK(hipMemset(B.out[s], 0xff, img_bytes));
    }
    CK(hipMalloc(&B.inv, K * K * 4)); CK(hipMalloc(&B.p_hat, hphat.size() * 4));
    CK(hipMemcpy(B.inv, hinv.data(), K * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.p_hat, hphat.data(), hphat.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&B.ref[i], img_bytes));
    CK(hipMalloc(&B.refgrid, (size_t)N * n * 8)); CK(hipMalloc(&B.refidx, (size_t)N * n * 8));
    CK(hipMalloc(&B.grid, (size_t)N * n * 8)); CK(hipMalloc(&B.idx, (size_t)N * n * 8));
    CK(hipMalloc(&B.trace, 8 * 4096 * 8 * 8)); CK(hipMemset(B.trace, 0, 8 * 4096 * 8 * 8));

    // reference: the library's image-pair kernel (round 2's production path), forced
    float* prepared = nullptr;
    CK(hipMalloc(&prepared, prep_floats(H, W, F) * 4));
    if (prep(B.p_hat, K, H, W, F, prepared, nullptr)) { printf("prepare failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    g_packed = prepared + (size_t)K * n;   // second part of the prepared table (tpspp_warp.hip)
    g_p_hat = B.p_hat;
    set_tuning(0, 0, 5, 0);
    auto pairk = [&](int set, float* out, float* grid, int32_t* idx, hipStream_t st) {
        int rc = warp_fwd(B.in[set], C, H, W, nullptr, 0, 0, 0, B.ctrl[set], nullptr, B.inv, B.p_hat, K, nullptr,
                          prepared, 1 | 8, N, F, H, W, out, nullptr, grid, idx, st);
        if (rc) { printf("tpspp_warp_fwd rc=%d\n", rc); exit(1); }
    };
    pairk(0, B.ref[0], nullptr, nullptr, nullptr);
    pairk(1, B.ref[1], B.refgrid, B.refidx, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<float> href[2] = {std::vector<float>((size_t)N * C * n), std::vector<float>((size_t)N * C * n)};
    std::vector<float> hrefgrid((size_t)N * n * 2); std::vector<int32_t> hrefidx((size_t)N * n * 2);
    for (int i = 0; i < 2; ++i) CK(hipMemcpy(href[i].data(), B.ref[i], img_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hrefgrid.data(), B.refgrid, hrefgrid.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hrefidx.data(), B.refidx, hrefidx.size() * 4, hipMemcpyDeviceToHost));

    const double bytes = (double)N * (2.0 * C * n * 4 + F * 2 * 4);
    auto fr = [&](float us) { return bytes / us / 1e6 / 8.0; };

    std::vector<Variant> vars;
    vars.push_back({"library at argv[2] (reference for the bit-exact check)", [&](const Bufs&, int s, float* o, float* g, int32_t* ix, long long*, hipStream_t st) { pairk(s, o, g, ix, st); }, true});
    vars.push_back(pair_variant<0>("this header, as the library builds it"));
    vars.push_back(pair_variant<1>("packed chains"));
    vars.push_back(pair_variant<2>("grid chains on the matrix pipe (4x4x1 f32)"));
    const int img1 = C * n * 4;
    CK(hipFuncSetAttribute((const void*)copy_img_k<1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    vars.push_back({"copy, same shape: 512 x 512 thr, regs, nt ld/st", [=](const Bufs& Bf, int s, float* o, float*, int32_t*, long long*, hipStream_t st) {
        hipLaunchKernelGGL((copy_img_k<0, 512>), dim3(N), dim3(512), 0, st, Bf.in[s], o, img1, N); }, false});
    vars.push_back({"copy, same shape: 512 x 512 thr, LDS-DMA + flat nt st", [=](const Bufs& Bf, int s, float* o, float*, int32_t*, long long*, hipStream_t st) {
        hipLaunchKernelGGL((copy_img_k<1, 512>), dim3(N), dim3(512), 41 * 1024, st, Bf.in[s], o, img1, N); }, false});
    vars.push_back({"copy, same shape, LDS-DMA, second half delayed", [=](const Bufs& Bf, int s, float* o, float*, int32_t*, long long*, hipStream_t st) {
        hipLaunchKernelGGL((copy_img_k<1, 512>), dim3(N), dim3(512), 41 * 1024, st, Bf.in[s], o, img1, N / 2); }, false});
    vars.push_back({"copy, flat 2048 x 256 grid-stride, nt ld/st", [=](const Bufs& Bf, int s, float* o, float*, int32_t*, long long*, hipStream_t st) {
        hipLaunchKernelGGL(copy_flat_k, dim3(2048), dim3(256), 0, st, (const v4f*)Bf.in[s], (char*)o, (int)(img_bytes / 16)); }, false});

    std::vector<float> hout((size_t)N * C * n), hgrid((size_t)N * n * 2); std::vector<int32_t> hidx((size_t)N * n * 2);
    for (auto& v : vars) {
        if (*only && v.name.find(only) == std::string::npos) continue;
        printf("-- %s\n", v.name.c_str()); fflush(stdout);
        bool ok = true; size_t bad = 0;
        for (int s = 0; s < 2 && v.checked; ++s) {
            CK(hipMemset(B.out[s], 0xff, img_bytes));
            if (s == 1) { CK(hipMemset(B.grid, 0xff, (size_t)N * n * 8)); CK(hipMemset(B.idx, 0xff, (size_t)N * n * 8)); }
            CK(hipDeviceSynchronize());   // the memsets run on the null stream, the variant on a non-blocking one
            v.run(B, s, B.out[s], s ? B.grid : nullptr, s ? B.idx : nullptr, nullptr, g_st[0]);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hout.data(), B.out[s], img_bytes, hipMemcpyDeviceToHost));
            if (memcmp(hout.data(), href[s].data(), img_bytes) != 0) {
                ok = false;
                for (size_t i = 0; i < hout.size(); ++i) if (memcmp(&hout[i], &href[s][i], 4)) { if (!bad) printf("   first diff set %d at %zu: %g vs %g\n", s, i, hout[i], href[s][i]); ++bad; }
            }
            if (s == 1) {
                CK(hipMemcpy(hgrid.data(), B.grid, hgrid.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hidx.data(), B.idx, hidx.size() * 4, hipMemcpyDeviceToHost));
                if (memcmp(hgrid.data(), hrefgrid.data(), hgrid.size() * 4) || memcmp(hidx.data(), hrefidx.data(), hidx.size() * 4)) { ok = false; printf("   grid/idx differ\n"); }
            }
        }
        if (!v.checked) {   // a copy must copy
            CK(hipMemset(B.out[0], 0xff, img_bytes)); CK(hipDeviceSynchronize());
            v.run(B, 0, B.out[0], nullptr, nullptr, nullptr, g_st[0]); CK(hipDeviceSynchronize());
            CK(hipMemcpy(hout.data(), B.out[0], img_bytes, hipMemcpyDeviceToHost));
            std::vector<float> hin((size_t)N * C * n); CK(hipMemcpy(hin.data(), B.in[0], img_bytes, hipMemcpyDeviceToHost));
            ok = memcmp(hout.data(), hin.data(), img_bytes) == 0;
        }
        for (int streams : {1, 2, 3}) {
            const float us = period_us([&](int i, hipStream_t st) { v.run(B, i % SETS, B.out[i % SETS], nullptr, nullptr, nullptr, st); }, iters, streams);
            printf("   %d stream%s: %7.2f us/launch  %6.3f TB/s  frac %.3f  %s\n", streams, streams > 1 ? "s" : " ", us, bytes / us / 1e6, fr(us),
                   v.checked ? (ok ? "[bit-exact]" : "[MISMATCH]") : (ok ? "[copies]" : "[COPY WRONG]"));
            fflush(stdout);
        }
        if (!ok && bad) printf("   %zu differing output words\n", bad);
        if (v.name.compare(0, 4, "pair") == 0) {
            const int nb = (N + v.imgs - 1) / v.imgs;
            for (int streams : {1, 2}) {
                const int Ln = 6;
                CK(hipMemset(B.trace, 0, (size_t)Ln * N * 8 * 8)); CK(hipDeviceSynchronize());
                for (int l = 0; l < 20; ++l) v.run(B, l % SETS, B.out[l % SETS], nullptr, nullptr, nullptr, g_st[l % streams]);
                for (int l = 0; l < Ln; ++l) v.run(B, (l + 3) % SETS, B.out[(l + 3) % SETS], nullptr, nullptr, B.trace + (size_t)l * nb * 8, g_st[l % streams]);
                CK(hipDeviceSynchronize());
                printf("    trace, %d stream%s:\n", streams, streams > 1 ? "s" : "");
                trace_report(B, nb, Ln);
            }
        }
    }
    // ---- interleaved re-timing ----
    {
        const int rounds = 7, it = 600;
        std::vector<std::vector<float>> t1(vars.size()), t2(vars.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t k = 0; k < vars.size(); ++k) {
                if (*only && vars[k].name.find(only) == std::string::npos) continue;
                t1[k].push_back(period_us([&](int i, hipStream_t st) { vars[k].run(B, i % SETS, B.out[i % SETS], nullptr, nullptr, nullptr, st); }, it, 1, 20));
                t2[k].push_back(period_us([&](int i, hipStream_t st) { vars[k].run(B, i % SETS, B.out[i % SETS], nullptr, nullptr, nullptr, st); }, it, 2, 20));
            }
        printf("interleaved, %d rounds x %d launches, us per launch min / median (fraction of 8 TB/s at the median):\n", rounds, it);
        printf("  %-56s %-28s %-28s\n", "", "one stream", "two streams");
        for (size_t k = 0; k < vars.size(); ++k) {
            if (t1[k].empty()) continue;
            std::sort(t1[k].begin(), t1[k].end()); std::sort(t2[k].begin(), t2[k].end());
            const float m1 = t1[k][t1[k].size() / 2], m2 = t2[k][t2[k].size() / 2];
            printf("  %-56s %6.2f / %6.2f  (%.3f)      %6.2f / %6.2f  (%.3f)\n", vars[k].name.c_str(), t1[k][0], m1, fr(m1), t2[k][0], m2, fr(m2));
        }
    }
    return 0;
}
