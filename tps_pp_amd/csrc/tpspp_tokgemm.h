// Internal interface between the recogniser head (tpspp_head.hip) and the token GEMM on the bf16 matrix cores
// (tpspp_tokgemm.hip): out (Co, M) = act(W^T X + bias) [+ res] for channel-major activations, M = images x tokens.
#pragma once
#include <hip/hip_runtime.h>

namespace tpspp {

struct TokGemmArgs {
    const float* X;            // (K, M) fp32, channel-major
    const void* W;             // the weight as tpspp_conv2d_bf16_fwd takes it for a 1x1 kernel (ops.prep_conv_weight_bf16):
                               // [Co / 64][K / kc][hi | lo][kc / 8][64][8] bf16 (the [hi | lo] level only with x3)
    const float* bias;         // (Co) or null
    const float* res;          // (Co, M) fp32 or null
    void* out;                 // (Co, M) fp32 or bf16
    int out_f32;
    int K, Co, M;
    int act;                   // 0 none, 2 GELU (erf)
    int x3;                    // three-term split: X and W as hi + lo bf16 halves, products hi*hi + hi*lo + lo*hi
    int kgc;                   // k groups of 8 per weight chunk (kc / 8)
};

// K % 32 == 0, K % kc == 0, Co % 128 == 0, M % 4 == 0, 16-byte aligned X
bool tok_gemm_applicable(const TokGemmArgs& a);
void launch_tok_gemm(const TokGemmArgs& a, hipStream_t st);

}  // namespace tpspp
