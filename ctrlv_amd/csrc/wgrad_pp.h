// Internal interface of wgrad_pp.hip (the LDS-DMA weight-gradient kernel) for backward.hip's ctrlv_gemm_wgrad.
#pragma once
#include "common.h"

struct ctrlv_wgrad_pp_plan_t {
  int ntiles, ktiles, slabs, rows_per_slab;
};
// does the kernel serve this layer (mode 0, stride-1 3x3, temporal conv; N, Cin multiples of 64; no concat operand)?
bool ctrlv_wgrad_pp_serves(const ctrlv_gemm_desc& d, const void* dY, int ldy);
// grid decomposition: (n tile of 320, k tile of 256, row slab); a function of the layer only
void ctrlv_wgrad_pp_plan(const ctrlv_gemm_desc& d, ctrlv_wgrad_pp_plan_t* p);
// the main launch; `part` as in wgrad_kernel (slab partials [slabs][N][K] + [slabs][N] bias sums, or null: fp32 atomics)
int ctrlv_wgrad_pp_launch(const ctrlv_gemm_desc& d, const void* dY, int ldy, float* dW, float* dbias, float scale,
                          int torch_layout, float* part, const ctrlv_wgrad_pp_plan_t& p, ctrlv_stream_t stream);
