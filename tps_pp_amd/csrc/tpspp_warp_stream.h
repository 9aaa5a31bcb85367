// Internal interface between the dispatcher (tpspp_warp.hip) and the plane-streaming kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tpspp {

struct StreamArgs {
    const float* in0; int C0, H0, W0;
    const float* in1; int C1, H1, W1;
    const float* ctrl; const float* score; const float* inv_delta_c;
    const float* p_hat; int p_hat_ld; const float* p_xy; const float* p_hat_t;
    int N, F, Ho, Wo;
    int score_t;               // 1: score is (N, F, n)
    int io_bf16;               // 1: in0 / in1 / out0 / out1 are bf16 in memory (pointers reinterpreted)
    float* out0; float* out1; float* grid; int32_t* idx;
};

// Shape / alignment / LDS-budget test for the plane-streaming kernel.
bool stream_kernel_applicable(const StreamArgs& a);
// Enqueue it (only call when applicable).  Returns TPSPP_OK / TPSPP_EIO.
int launch_stream_kernel(const StreamArgs& a, long long* trace, hipStream_t st);

}  // namespace tpspp
