// Ping-pong gather-GEMM instantiations: mode 0 (nn.Linear / 1x1 conv), incl. the GEGLU variant, and the dispatcher.
#include "gemm_pp_kernel.h"

int ctrlv_gemm_launch_pp_conv(const ctrlv_gemm_desc& d, int tile, bool persistent, hipStream_t stream);      // gemm_pp_m1.hip
int ctrlv_gemm_launch_pp_temporal(const ctrlv_gemm_desc& d, int tile, bool persistent, hipStream_t stream);  // gemm_pp_m2.hip

bool ctrlv_conv_halo_order(const ctrlv_gemm_desc& d) {
  return ctrlv_debug().conv_halo != 0 && conv_halo_geometry(d);     // (0: per-tap gather, tap-major K order of round 3)
}

// does the ping-pong family serve this descriptor's epilogue? (gemm.hip falls back to the 128x128 kernel otherwise)
bool ctrlv_gemm_pp_supports(const ctrlv_gemm_desc& d) {
  // the epilogue addresses out / R1 / R2 / V through buffer descriptors: 32-bit byte offsets
  const long lim = 0xFFFFFFF0L;
  if ((long)d.M * d.ldo * 2 > lim) return false;
  if (d.R1 && (long)d.M * d.ldr1 * 2 > lim) return false;
  if (d.R2 && (long)d.M * d.ldr2 * 2 > lim) return false;
  if (d.vmode && pp_vtable_rows(d) * d.ldv * 4 > lim) return false;
  {  // sources: 32-bit offsets as well (row offset + the tap-displacement bias the descriptor base is moved down by)
    const long a_rows = d.mode == 1 ? (long)(d.M / (d.Ho * d.Wo)) * d.H * d.Wd : (long)d.M;
    const long bias_rows = d.mode == 1 ? d.Wd + 1 : (d.mode == 2 ? d.S : 0);
    const long ld = d.lda > d.lda2 ? d.lda : d.lda2;
    if ((a_rows + 2 * bias_rows) * ld * 2 > lim) return false;
    if ((long)d.N * d.taps * d.Cin * 2 > lim) return false;
  }
  if (d.taps * (d.Cin >> 5) < 4) return false;       // the DMA ring runs three half-steps ahead inside one tile
  // a second A source (skip concat) is instantiated for the plain GEMM with a bias-only epilogue (1x1 shortcut convs)
  if (d.A2 && !(d.mode == 0 && !d.geglu && pp_epi_of(d) == 0)) return false;
  if (d.raw_out && ((long)d.M * d.ld_raw * 2 > lim || d.ld_raw % 8 != 0)) return false;
  if (d.geglu) return d.mode == 0 && !pp_split_io(d);
  const int e = pp_epi_of(d);
  if (e < 0) return false;
  if (pp_split_io(d)) {   // split trunk planes (launch_epi_split): fp16 elements, N a multiple of 320 (tile 6)
    if (CTRLV_ELEM_DTYPE != 1 || d.N % 320 != 0 || d.raw_out || d.n_scale2) return false;
    // (with GroupNorm partials: the two {R1} trunk writers that feed a norm -- row-halo conv2, temporal conv2)
    if (d.gn_partials) return e == 2 && (d.mode == 2 || (d.mode == 1 && ctrlv_conv_halo_order(d)));
    if (d.mode == 0) return e == 0 || e == 2 || e == 3 || e == 6;
    if (d.mode == 1) return e == 0 || e == 2;
    return e == 2;
  }
  if (e == 8) return (long)d.ksplit * d.Cin == d.w_cin;                   // K slices (any gather mode)
  return d.mode == 0 || e <= 2;
}

// Does a launch of this descriptor write GroupNorm chunk partials when gn_partials is set (gemm_epilogue_lds, GNS)?  A
// function of the layer's shape only -- never of the batch size -- so a clip is normalised through the same path alone
// and in a batch: ctrlv_gemm serves such a launch on the 256x320 tile whatever M is.
extern "C" int ctrlv_gemm_gn_partials_serves(const ctrlv_gemm_desc* dp) {
  if (!dp) return 0;
  const ctrlv_gemm_desc& d = *dp;
  if (!ctrlv_debug().gn_fused) return 0;
  const int cpg = d.N / 32;
  if (d.N <= 0 || d.N % 320 != 0 || !(cpg == 10 || cpg == 20 || cpg == 40)) return 0;      // 160-column wave tiles hold whole groups
  if (d.n_store != d.N || d.ldo % 8 != 0 || d.geglu || d.A2 || d.raw_out || d.n_scale2) return 0;
  if (pp_split_io(d) && pp_epi_of(d) != 2) return 0;        // split planes: the {R1} writers only (launch_epi_split)
  if ((d.R1 && d.ldr1 % 8 != 0) || (d.vmode && d.ldv % 8 != 0) || d.Cin % 64 != 0) return 0;
  const int e = pp_epi_of(d);
  long S = 0;
  if (d.mode == 1) {
    if (!(e == 1 || e == 2) || !ctrlv_conv_halo_order(d)) return 0;
    S = (long)d.Ho * d.Wo;
  } else if (d.mode == 2) {
    const int cross = ctrlv_debug().gn_cross;   // (0: the {R1} temporal conv -- a res block's last GEMM, feeding the transformer's norm -- does not serve)
    if (!(e == 1 || (e == 2 && cross))) return 0;
    S = d.S;
  } else {
    return 0;
  }
  if (S <= 0 || S % 64 != 0 || d.M <= 0 || d.M % S != 0) return 0;      // a 64-row wave tile never straddles two images
  if (d.vmode && !(d.vmode == 1 && d.vdiv % 64 == 0)) return 0;          // ... nor two rows of the row-vector table
  return ctrlv_gemm_pp_supports(d) ? 1 : 0;
}

// tile 5: 256x256 (waves 2x4); tile 6: 256x320 (waves 4x2); tiles 7 / 8: the same kernels launched with one
// workgroup per output tile instead of one persistent workgroup per CU; tile 10: 256x128 (waves 4x2; 3x3 and temporal
// convs with N <= 128 -- on the 256-wide tile half of every MFMA would be padding).
int ctrlv_gemm_launch_pp(const ctrlv_gemm_desc& d, int tile, hipStream_t stream) {
  const bool persistent = tile <= 6 || tile == 10;
  if (d.mode == 1) return ctrlv_gemm_launch_pp_conv(d, tile, persistent, stream);
  if (d.mode == 2) return ctrlv_gemm_launch_pp_temporal(d, tile, persistent, stream);
  if (tile == 10) {
    ctrlv_set_error("ctrlv_gemm: tile 10 (256x128) is instantiated for the 3x3 and temporal convs only");
    return CTRLV_E_BAD_ARG;
  }
  if (d.geglu) {
    if (d.raw_out) {      // training forward: the raw projection is written too
      if (tile == 5 || tile == 7) return launch_one<256, 2, 4, 0, true, 0, false, true>(d, persistent, stream);
      return launch_one<320, 4, 2, 0, true, 0, false, true>(d, persistent, stream);
    }
    if (tile == 5 || tile == 7) return launch_one<256, 2, 4, 0, true, 0>(d, persistent, stream);
    return launch_one<320, 4, 2, 0, true, 0>(d, persistent, stream);
  }
  if (tile == 5 || tile == 7) return launch_epi<256, 2, 4, 0>(d, persistent, stream);
  return launch_epi<320, 4, 2, 0>(d, persistent, stream);
}

CTRLV_CLOCK_READER(pp_m0)
