// Launcher of the run-time-geometry in-place warp kernel (tpspp_warp_geo.h): geometry planning + the (C, QP, AUX)
// instantiations.  Its own translation unit: compiled with -fno-slp-vectorize (tps_pp_amd/build.py) -- the vectoriser
// packs the (x, y) pairs of the grid chains into v_pk_fma_f32, which is slower here (profiles/r04_warp_lab.txt: a
// packed FMA occupies the SIMD for two passes and the kernel is bound by VALU time, not by issue slots).
#include "tpspp_common.h"
#include "tpspp_warp_geo.h"
#include "tpspp_warp_span.h"
#include "tpspp_warp_geo_launch.h"

namespace tpspp {

using tpspp_dev::kWave;

// the packed table's quadrant pixels per thread for an output geometry (0: no packed form).  The smallest divisor of the
// row groups that leaves at most 13 compute wavefronts per image -- the layout the image-pair kernel and the
// instantiated in-place kernels (tpspp_warp_img.h) were built for --, else the largest divisor <= 4: the run-time
// kernel then splits an image over several workgroups ("bands").
int geo_qp(int Ho, int Wo)
{
    if (Ho <= 0 || Wo <= 0 || Wo % 4 != 0 || Ho % 16 != 0) return 0;
    const int BW = tpspp_img::img_block_w(Wo), BH = 32 / BW;
    const int CG = ((Wo / 2) + BW - 1) / BW, RG = (Ho / 2) / BH;
    if (CG * BW > Wo || RG < 1) return 0;
    for (int qp = 1; qp <= RG && qp <= 4; ++qp)
        if (RG % qp == 0 && (CG * (RG / qp) * 32 + kWave - 1) / kWave <= 13) return qp;
    for (int qp = 4; qp >= 1; --qp)
        if (RG % qp == 0) return qp;
    return 0;
}

int geo_nthr(int Ho, int Wo, int QP)
{
    const int BW = tpspp_img::img_block_w(Wo), BH = 32 / BW;
    return (((Wo / 2) + BW - 1) / BW) * (((Ho / 2) / BH) / QP) * 32;
}

// the span kernel's own packed copy (QP = 1: one quadrant pixel per thread) exists for the geometries whose quadrant one
// workgroup of the in-place kernels cannot cover with <= 2 pixels per thread: wavefronts of that copy, 0 = none
int span_table_waves(int Ho, int Wo)
{
    if (geo_qp(Ho, Wo) < 3) return 0;
    const int BW = tpspp_img::img_block_w(Wo), BH = 32 / BW;
    const int CG = ((Wo / 2) + BW - 1) / BW, RG = (Ho / 2) / BH;
    return (CG * RG * 32 + kWave - 1) / kWave;
}

namespace {

struct Plan { int QP, BW, CG, RGB, bands, nthr, NW, nload, imgs; size_t lds; };
struct SpanPlan { int BW, CG, RG, bands, nthr, NWv, span_rows, chunk_floats, margin, spec; size_t lds; };

int g_span_bands = 0;        // lab knobs (tpspp_warp_set_tuning with kernel_choice 8): workgroups per image, 0 = heuristic;
int g_span_gather = 0;       // every workgroup on the global-memory path;
int g_span_lds_kb = 0;       // LDS budget per workgroup in KB, 0 = 38 (four workgroups per CU)
int g_span_no_spec = 0;      // the measured-span form (rounds 5's first version) instead of the windows requested at launch

// Row bands with span staging (tpspp_warp_span.h): workgroups per image (a divisor of the quadrant's row groups) and the
// staging buffer's size; the buffer then takes whatever span fits (the rows the band's taps reach, per region).
bool plan_span(int C, int H, int W, int F, SpanPlan* p)
{
    if (!(C == 1 || C == 3 || C == 4) || F != 20 || span_table_waves(H, W) == 0 || (H * W) % 4 != 0) return false;
    p->BW = tpspp_img::img_block_w(W);
    const int BH = 32 / p->BW;
    p->CG = ((W / 2) + p->BW - 1) / p->BW;
    p->RG = (H / 2) / BH;
    const int K = F + 3;
    // Measured at batch 512 (scripts/debug/bench_span_knobs.py): the kernel is bound by vector-ALU time per SIMD, so (a) a
    // workgroup whose wavefronts spread evenly over the four SIMDs wins (48x160: 12 wavefronts 32.8 us, 9 wavefronts 39.4),
    // (b) small workgroups with a small buffer -- four per CU -- beat two large ones (64x200: 7 wavefronts / 38 KB 48.7 us,
    // 7 / 78 KB 59.4, 13 / any 64.3).  Preference: whole multiples of four wavefronts (the most, <= 12), else the most
    // wavefronts <= 8, else anything <= 13; buffer 38 KB, more only when the band's own rows + 3 do not fit.
    const size_t kb = (size_t)(g_span_lds_kb > 0 ? g_span_lds_kb : 38);
    const size_t budgets[3] = {kb * 1024, (size_t)78 * 1024, (size_t)158 * 1024};
    p->margin = 0; p->spec = 0;
    for (size_t budget : budgets)
        for (int cls = 0; cls < 3; ++cls) {
            int bestB = 0, bestNW = 0;
            for (int B = 1; B <= p->RG; ++B) {
                if (p->RG % B) continue;
                if (g_span_bands > 0 && B != g_span_bands) continue;
                const int nthr = p->CG * (p->RG / B) * 32, NWv = (nthr + kWave - 1) / kWave;
                const bool ok = cls == 0 ? (nthr % 256 == 0 && NWv <= 12) : cls == 1 ? NWv <= 8 : NWv <= 13;
                if (!ok || NWv <= bestNW) continue;
                const int rows = (p->RG / B) * BH;
                const size_t fixed = (size_t)(tpspp_span::span_stage_off(K) + W + 4) * 4;
                if (budget <= fixed) continue;
                // the largest span whose chunks (whole 1-KB pieces per channel) fit the budget
                int span_rows = (int)(((budget - fixed) / C / 1024) * 1024 / ((size_t)W * 4));
                if (span_rows > H) span_rows = H;
                if (span_rows < rows + 3 && span_rows < H) continue;
                if (tpspp_span::span_lds_bytes(K, C, W, span_rows, rows) > 160 * 1024) continue;
                bestB = B; bestNW = NWv;
                p->bands = B; p->nthr = nthr; p->NWv = NWv; p->span_rows = span_rows;
                p->chunk_floats = tpspp_span::span_chunk_floats(span_rows, W);
                p->lds = tpspp_span::span_lds_bytes(K, C, W, span_rows, rows);
            }
            if (bestB) {
                // windows requested at launch (SPEC) for the SAME decomposition: both regions' windows (band rows + 2 x margin)
                // at once; margin 2 rows (TPSPP_SPAN_MARGIN overrides: 1 -> more wavefronts on the global-memory path at
                // +-1.6 rows of displacement, 3 -> three workgroups per CU; both slower) -- only where four workgroups still
                // share a CU (<= 40 KB: 64x200 45.4 against 48.8 us, 48x160 26.4 against 29.9; 64x256 would need 50 KB and
                // measures 56.8 against 52.5 -- or fewer, smaller workgroups: worse --: it keeps the measured spans)
                if (!g_span_no_spec) {
                    const char* mv = getenv("TPSPP_SPAN_MARGIN");
                    int mg = mv ? atoi(mv) : 2;
                    if (mg < 0) mg = 0;
                    const int rows = (p->RG / p->bands) * BH;
                    int win = rows + 2 * mg;
                    if (win > H) win = H;
                    const int chunk = win * W;           // exact: no rounding to whole DMA pieces (see the kernel)
                    size_t stage = (size_t)2 * C * chunk;
                    const size_t outb = (size_t)2 * C * rows * W;
                    if (outb > stage) stage = outb;
                    const size_t lds = (size_t)(tpspp_span::span_stage_off(K) + stage + W + 4) * 4;
                    const size_t cap = (size_t)(g_span_lds_kb > 0 ? g_span_lds_kb : 40) * 1024;
                    if (rows * W * 4 >= 1008 && lds <= cap) {   // (a piece straddles at most two channels, also in a clipped window)
                        p->spec = 1; p->margin = mg; p->span_rows = win; p->chunk_floats = chunk; p->lds = lds;
                    }
                }
                return true;
            }
        }
    return false;
}

template <int C, bool AUX, bool SPEC, typename G = tpspp_span::SpanRT>
void launch_span_one(const tpspp_span::SpanParams& P, const SpanPlan& pl, hipStream_t st)
{
    auto kern = tpspp_span::tps_warp_span_kernel<20, C, AUX, SPEC, G>;
    static bool attr_done[kMaxDevices] = {};
    if (first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const unsigned blocks = (unsigned)((P.N + 7) / 8) * 8u * (unsigned)pl.bands;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3((unsigned)(pl.NWv * kWave)), pl.lds, st, P);
}

int g_span_no_fixed = 0;     // lab knob (TPSPP_SPAN_RUNTIME=1): the run-time-geometry form for the instantiated geometries as well

// the instantiated geometries (C = 3, no auxiliary outputs): taken when the plan is exactly the one they were compiled for
template <typename G, bool SPEC>
bool span_fixed_matches(const tpspp_span::SpanParams& P, const SpanPlan& pl)
{
    return P.H == G::H && P.W == G::W && pl.BW == G::BW && pl.CG == G::CG && pl.RG == G::RG && pl.bands == G::BANDS &&
           pl.nthr == G::NTHR && (pl.spec != 0) == SPEC && (!SPEC || pl.margin == G::MARGIN);
}

template <int C>
void launch_span_aux(const tpspp_span::SpanParams& P, const SpanPlan& pl, hipStream_t st)
{
    if constexpr (C == 3) {
        static const bool rt = getenv("TPSPP_SPAN_RUNTIME") != nullptr;
        if (!rt && !g_span_no_fixed && !P.grid && !P.idx && !P.force_gather) {
            using G64x200 = tpspp_span::SpanFix<64, 200, 8, 13, 8, 8, 2>;
            using G48x160 = tpspp_span::SpanFix<48, 160, 16, 5, 12, 4, 2>;
            using G64x256 = tpspp_span::SpanFix<64, 256, 32, 4, 32, 8, 0>;
            if (span_fixed_matches<G64x200, true>(P, pl)) { launch_span_one<3, false, true, G64x200>(P, pl, st); return; }
            if (span_fixed_matches<G48x160, true>(P, pl)) { launch_span_one<3, false, true, G48x160>(P, pl, st); return; }
            if (span_fixed_matches<G64x256, false>(P, pl)) { launch_span_one<3, false, false, G64x256>(P, pl, st); return; }
        }
    }
    if (pl.spec) {
        if (P.grid || P.idx) launch_span_one<C, true, true>(P, pl, st);
        else launch_span_one<C, false, true>(P, pl, st);
        return;
    }
    if (P.grid || P.idx) launch_span_one<C, true, false>(P, pl, st);
    else launch_span_one<C, false, false>(P, pl, st);
}

int g_geo_pair = 1;          // lab knob: 0 = never an image pair per workgroup
int g_geo_force_bands = 0;   // lab knob (tpspp_warp_set_tuning's `bands` with kernel_choice 7): 0 = heuristic

bool plan_geo(int C, int H, int W, int F, Plan* p)
{
    if (!(C == 1 || C == 3 || C == 4) || F != 20) return false;
    const int QP = geo_qp(H, W);
    if (QP == 0 || (H * W) % 4 != 0) return false;
    p->QP = QP;
    p->BW = tpspp_img::img_block_w(W);
    const int BH = 32 / p->BW;
    p->CG = ((W / 2) + p->BW - 1) / p->BW;
    p->RGB = ((H / 2) / BH) / QP;
    p->imgs = 1;
    p->lds = tpspp_geo::geo_lds_bytes(F + 3, C, H, W, 1);
    if (p->lds > 160 * 1024) return false;
    // loaders: one per ~40 KB of image; workgroup size: <= 16 wavefronts, <= 12 from QP = 3 on (register budget)
    p->nload = (int)((size_t)C * H * W * 4 / (40 * 1024)) + 1;
    if (p->nload > 3) p->nload = 3;
    const int max_waves = QP >= 3 ? 12 : 16;
    bool found = false;
    for (int B = 1; B <= p->RGB && !found; ++B) {
        if (p->RGB % B) continue;
        if (g_geo_force_bands > 0 && B != g_geo_force_bands && B < p->RGB) continue;
        const int nthr = p->CG * (p->RGB / B) * 32;
        const int NW = (nthr + kWave - 1) / kWave;
        if (NW + p->nload <= max_waves && NW <= 13) { p->bands = B; p->nthr = nthr; p->NW = NW; found = true; }
    }
    if (!found) return false;
    // image pair per workgroup: one workgroup covers the quadrant, QP x C <= 6 results per mirror pixel (registers),
    // two images fit the LDS, and every loader has >= kGeoKB pieces of image B to issue behind image A's
    const int pieces1 = (C * H * W * 4) / 1024;
    int nload2 = p->nload < 3 && pieces1 / (p->nload + 1) >= tpspp_geo::kGeoKB ? p->nload + 1 : p->nload;
    // (NW >= 2: compute wavefront g solves image g's T -- with a single compute wavefront, wavefront 1 is a loader and image
    // B's T would never be published)
    // (QP <= 2: launch_qp has the pair form for one and two quadrant pixels per thread only -- a plan that said "pair" for QP 3 / 4
    // was launched as the one-image kernel with the pair's LDS layout: 64x160 C = 1 faulted once its quadrant fitted one workgroup)
    if (g_geo_pair && p->bands == 1 && p->NW >= 2 && QP <= 2 && QP * C <= 6 && pieces1 / nload2 >= tpspp_geo::kGeoKB && p->NW + nload2 <= 16 &&
        tpspp_geo::geo_lds_bytes(F + 3, C, H, W, 2) <= 160 * 1024) {
        p->imgs = 2; p->nload = nload2;
        p->lds = tpspp_geo::geo_lds_bytes(F + 3, C, H, W, 2);
    }
    return true;
}

template <int C, int QP, int IMGS, bool AUX>
void launch_one(const tpspp_geo::GeoParams& P, const Plan& pl, hipStream_t st)
{
    auto kern = tpspp_geo::tps_warp_geo_kernel<20, C, QP, IMGS, AUX>;
    static bool attr_done[kMaxDevices] = {};
    if (first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const unsigned groups = (unsigned)((P.N + IMGS - 1) / IMGS);
    hipLaunchKernelGGL(kern, dim3(groups * (unsigned)pl.bands), dim3((unsigned)((pl.NW + pl.nload) * kWave)), pl.lds, st, P);
}

template <int C, int QP, int IMGS>
void launch_aux(const tpspp_geo::GeoParams& P, const Plan& pl, hipStream_t st)
{
    if (P.grid || P.idx) launch_one<C, QP, IMGS, true>(P, pl, st);
    else launch_one<C, QP, IMGS, false>(P, pl, st);
}

template <int C>
void launch_qp(const tpspp_geo::GeoParams& P, const Plan& pl, hipStream_t st)
{
    switch (pl.QP) {
    case 1: if (pl.imgs == 2) launch_aux<C, 1, 2>(P, pl, st); else launch_aux<C, 1, 1>(P, pl, st); break;
    case 2:
        if constexpr (2 * C <= 6) { if (pl.imgs == 2) { launch_aux<C, 2, 2>(P, pl, st); break; } }
        launch_aux<C, 2, 1>(P, pl, st); break;
    case 3: launch_aux<C, 3, 1>(P, pl, st); break;
    default: launch_aux<C, 4, 1>(P, pl, st); break;
    }
}

}  // namespace

void geo_set_bands(int bands) { g_geo_force_bands = bands & 7; g_geo_pair = (bands & 8) ? 0 : 1; }
void span_set_tuning(int bands, int gather, int lds_kb, int no_spec) { g_span_bands = bands; g_span_gather = gather; g_span_lds_kb = lds_kb; g_span_no_spec = no_spec; }

bool geo_kernel_single_workgroup(int C, int H, int W, int F)
{
    Plan pl;
    return plan_geo(C, H, W, F, &pl) && pl.bands == 1;
}

bool span_kernel_applicable(int C, int H, int W, int F)
{
    SpanPlan pl;
    return plan_span(C, H, W, F, &pl);
}

bool launch_span_kernel(int C, int H, int W, int F, const float* in, const float* ctrl, const float* inv_delta_c,
                        const float* span_packed, int N, float* out, float* grid, int32_t* idx, hipStream_t st)
{
    SpanPlan pl;
    if (!plan_span(C, H, W, F, &pl)) return false;
    tpspp_span::SpanParams P;
    P.in = in; P.ctrl = ctrl; P.inv_delta_c = inv_delta_c; P.packed = span_packed; P.N = N;
    P.out = out; P.grid = grid; P.idx = idx;
    P.H = H; P.W = W; P.BW = pl.BW; P.CG = pl.CG; P.RG = pl.RG; P.bands = pl.bands; P.nthr = pl.nthr;
    P.lg_bw = __builtin_ctz((unsigned)pl.BW);
    P.span_rows = pl.span_rows; P.chunk_floats = pl.chunk_floats; P.stage_off = tpspp_span::span_stage_off(F + 3);
    P.force_gather = g_span_gather;
    P.margin = pl.margin;
    if (C == 1) launch_span_aux<1>(P, pl, st);
    else if (C == 3) launch_span_aux<3>(P, pl, st);
    else launch_span_aux<4>(P, pl, st);
    return true;
}

bool geo_kernel_applicable(int C, int H, int W, int F)
{
    Plan pl;
    return plan_geo(C, H, W, F, &pl);
}

bool launch_geo_kernel(int C, int H, int W, int F, const float* in, const float* ctrl, const float* inv_delta_c,
                       const float* packed, int N, float* out, float* grid, int32_t* idx, hipStream_t st)
{
    Plan pl;
    if (!plan_geo(C, H, W, F, &pl)) return false;
    tpspp_geo::GeoParams P;
    P.in = in; P.ctrl = ctrl; P.inv_delta_c = inv_delta_c; P.packed = packed; P.N = N;
    P.out = out; P.grid = grid; P.idx = idx;
    P.H = H; P.W = W; P.BW = pl.BW; P.CG = pl.CG; P.RGB = pl.RGB; P.bands = pl.bands; P.nthr = pl.nthr; P.NW = pl.NW;
    P.img_off = tpspp_geo::geo_img_off(F + 3, pl.imgs);
    if (C == 1) launch_qp<1>(P, pl, st);
    else if (C == 3) launch_qp<3>(P, pl, st);
    else launch_qp<4>(P, pl, st);
    return true;
}

}  // namespace tpspp
