// Instantiations and dispatch of the four-waves-per-SIMD 16x16x32 gather-GEMM core (gemm_w16_kernel.h).
#include "gemm_w16_kernel.h"

// tile 12: 256 x 256 (wave tile 64 x 64; GEGLU and N % 256 == 0 layers), tile 13: 256 x 320 (wave tile 64 x 80).
template <int BN, int MODE>
static int launch_w16_epi(const ctrlv_gemm_desc& d, hipStream_t stream) {
  switch (pp_epi_of(d)) {
    case 0: return launch_w16<BN, MODE, false, 0>(d, stream);
    case 1: return launch_w16<BN, MODE, false, 1>(d, stream);
    case 2: return launch_w16<BN, MODE, false, 2>(d, stream);
    case 3:
      if constexpr (MODE == 0) return launch_w16<BN, MODE, false, 3>(d, stream);
      break;
    case 6:
      if constexpr (MODE == 0) return launch_w16<BN, MODE, false, 6>(d, stream);
      break;
    default: break;
  }
  ctrlv_set_error("ctrlv_gemm: epilogue operand combination not served by the 16x16x32 core");
  return CTRLV_E_BAD_ARG;
}

// what the core can run at all (the launcher's conditions; ctrlv_gemm_w16_serves adds the policy: which LAYERS it is given)
bool ctrlv_gemm_w16_supports(const ctrlv_gemm_desc& d, int tile) {
  const int bn = tile == 12 ? 256 : 320;
  const long lim = 0xFFFFFFF0L;
  if (d.A2 || (d.raw_out && (!d.geglu || d.ld_raw < d.N || d.ld_raw % 8 != 0 || (long)d.M * d.ld_raw * 2 > lim)) || d.gn_partials || d.ksplit || d.act || (d.out_f32 & 1) || d.R1_lo || d.R2_lo || d.out_lo) return false;
  if ((long)d.M * d.ldo * 2 > lim || (d.R1 && (long)d.M * d.ldr1 * 2 > lim) || (d.R2 && (long)d.M * d.ldr2 * 2 > lim)) return false;
  if (d.vmode && pp_vtable_rows(d) * d.ldv * 4 > lim) return false;
  {
    const long a_rows = d.mode == 1 ? (long)(d.M / (d.Ho * d.Wo)) * d.H * d.Wd : (long)d.M;
    const long bias_rows = d.mode == 1 ? d.Wd + 1 : (d.mode == 2 ? d.S : 0);
    if ((a_rows + 2 * bias_rows) * d.lda * 2 > lim) return false;
    if ((long)d.N * d.taps * d.Cin * 2 > lim) return false;
  }
  if (d.taps * (d.Cin >> 5) < 4 || d.Cin % 32 != 0) return false;
  if (d.n_store % 8 != 0 || d.ldo % 8 != 0 || (d.R1 && d.ldr1 % 8 != 0) || (d.R2 && d.ldr2 % 8 != 0) || (d.vmode && d.ldv % 8 != 0)) return false;
  if (d.geglu) return d.mode == 0 && bn == 256 && d.N % 64 == 0 && d.n_scale2 == 0;
  if (d.n_scale2 % 32 != 0 || (bn == 320 && d.n_scale2 % 80 != 0 && d.n_scale2 != 0)) return false;
  if (bn == 320 && d.N % 16 != 0) return false;
  const int e = pp_epi_of(d);
  if (e < 0 || e == 8) return false;
  return d.mode == 0 || e <= 2;
}

int ctrlv_gemm_launch_w16(const ctrlv_gemm_desc& d, int tile, hipStream_t stream) {
  if (!ctrlv_gemm_w16_supports(d, tile)) {
    ctrlv_set_error("ctrlv_gemm: launch not served by the 16x16x32 core (tile %d)", tile);
    return CTRLV_E_BAD_ARG;
  }
  if (tile == 12) {
    if (d.geglu) return launch_w16<256, 0, true, 0>(d, stream);
    if (d.mode == 0) return launch_w16_epi<256, 0>(d, stream);
    if (d.mode == 1) return launch_w16_epi<256, 1>(d, stream);
    return launch_w16_epi<256, 2>(d, stream);
  }
  if (d.mode == 0) return launch_w16_epi<320, 0>(d, stream);
  if (d.mode == 1) return launch_w16_epi<320, 1>(d, stream);
  return launch_w16_epi<320, 2>(d, stream);
}

CTRLV_CLOCK_READER(w16)
