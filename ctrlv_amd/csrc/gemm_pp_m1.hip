// Ping-pong gather-GEMM instantiations: mode 1 (3x3 Conv2d: stride 1 / 2, nearest-x2 upsample fused).
#include "gemm_pp_kernel.h"

int ctrlv_gemm_launch_pp_conv(const ctrlv_gemm_desc& d, int tile, bool persistent, hipStream_t stream) {
  if (tile == 10) return launch_epi<128, 4, 2, 1>(d, persistent, stream);     // N <= 128 (the VAE decoder's top level)
  if (tile == 5 || tile == 7) return launch_epi<256, 2, 4, 1>(d, persistent, stream);
  return launch_epi<320, 4, 2, 1>(d, persistent, stream);
}

CTRLV_CLOCK_READER(pp_m1)
