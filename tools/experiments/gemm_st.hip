// Streamed 4-wave gather-GEMM (two workgroups per CU) for the short-K nn.Linear layers: instantiations + dispatcher.
#include "gemm_st_kernel.h"

bool ctrlv_gemm_pp_supports(const ctrlv_gemm_desc& d);   // gemm_pp_m0.hip: 32-bit offset limits, epilogue operand sets

// tile 9: 256 x 160, mode 0, one A source, K a multiple of 32 and >= 96
bool ctrlv_gemm_st_supports(const ctrlv_gemm_desc& d) {
  if (d.mode != 0 || d.A2 || d.raw_out || (d.Cin & 31) || d.Cin < 96) return false;
  if (d.geglu) return true;
  return pp_epi_of(d) >= 0;
}

int ctrlv_gemm_launch_st(const ctrlv_gemm_desc& d, hipStream_t stream) {
  if (d.geglu) return launch_st_one<160, true, 0>(d, stream);
  switch (pp_epi_of(d)) {
    case 0: return launch_st_one<160, false, 0>(d, stream);
    case 1: return launch_st_one<160, false, 1>(d, stream);
    case 2: return launch_st_one<160, false, 2>(d, stream);
    case 3: return launch_st_one<160, false, 3>(d, stream);
    case 6: return launch_st_one<160, false, 6>(d, stream);
    default: break;
  }
  ctrlv_set_error("ctrlv_gemm: epilogue operand combination not served by the streamed kernel");
  return CTRLV_E_BAD_ARG;
}

// Diagnostic (tests): resident workgroups per CU of the streamed GEGLU kernel at its launch configuration -- the
// schedule only pays when TWO fit (LDS <= 80 KB, <= 256 registers).
extern "C" int ctrlv_gemm_st_occupancy() {
  constexpr int smem = 3 * (256 + 160) * 64 + 160 * 4 + 1024;
  auto kfn = gemm_st_kernel<160, true, 0>;
  if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return -1;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kfn, 256, smem) != hipSuccess) return -1;
  return n;
}
