// Internal interface between the dispatcher (tpspp_warp.hip) and the run-time-geometry in-place kernel
// (tpspp_warp_geo.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tpspp {

// quadrant pixels per thread of the packed table for an output geometry (0: none) and the compute threads of that mapping
int geo_qp(int Ho, int Wo);
int geo_nthr(int Ho, int Wo, int QP);
// does (C, H = Ho, W = Wo, F) fit the kernel: C in {1, 3, 4}, F = 20, a packed table exists, the image fits the LDS
bool geo_kernel_applicable(int C, int H, int W, int F);
// lab knob: force the number of workgroups per image (0 = heuristic)
void geo_set_bands(int bands);
// enqueue it; false (nothing launched) when not applicable
bool launch_geo_kernel(int C, int H, int W, int F, const float* in, const float* ctrl, const float* inv_delta_c,
                       const float* packed, int N, float* out, float* grid, int32_t* idx, hipStream_t st);

// ---- row bands with span staging (tpspp_warp_span.h): the large geometries ----
// wavefronts of the span kernel's own packed copy of the table (QP = 1 layout, third section of the prepared table); 0: none
int span_table_waves(int Ho, int Wo);
// does one workgroup of the in-place kernel cover the image (no bands)
bool geo_kernel_single_workgroup(int C, int H, int W, int F);
bool span_kernel_applicable(int C, int H, int W, int F);
// lab knobs: workgroups per image (0 = heuristic), every workgroup on the global-memory path, LDS budget in KB (0 = 38)
void span_set_tuning(int bands, int gather, int lds_kb, int no_spec);
bool launch_span_kernel(int C, int H, int W, int F, const float* in, const float* ctrl, const float* inv_delta_c,
                        const float* span_packed, int N, float* out, float* grid, int32_t* idx, hipStream_t st);

}  // namespace tpspp
