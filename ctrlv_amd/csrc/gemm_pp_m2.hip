// Ping-pong gather-GEMM instantiations: mode 2 (temporal Conv3d (3,1,1)).
#include "gemm_pp_kernel.h"

int ctrlv_gemm_launch_pp_temporal(const ctrlv_gemm_desc& d, int tile, bool persistent, hipStream_t stream) {
  if (tile == 10) return launch_epi<128, 4, 2, 2>(d, persistent, stream);
  if (tile == 5 || tile == 7) return launch_epi<256, 2, 4, 2>(d, persistent, stream);
  return launch_epi<320, 4, 2, 2>(d, persistent, stream);
}

CTRLV_CLOCK_READER(pp_m2)
