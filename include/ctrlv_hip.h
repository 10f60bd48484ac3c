/*
 * ctrlv_hip.h -- C ABI of libctrlv_hip.so: the MI355X (gfx950) implementation of Ctrl-V's denoising hot path.
 *
 * The reference (oooolga/Ctrl-V) has NO native boundary: its hot path is two Python callables whose arithmetic lives
 * in diffusers==0.27.2 torch modules.  Each entry point below names the reference interface (file:line under
 * /root/reference, or the diffusers block it instantiates) that it replaces.  All functions:
 *   - take plain device pointers + sizes (no torch types), enqueue on the caller's hipStream_t, never allocate
 *     device memory, never synchronise the host;
 *   - return 0 on success, <0 on error (CTRLV_E_*); the message is available through ctrlv_last_error();
 *   - read inputs only; outputs are caller-owned buffers.
 *
 * Activations are CHANNELS-LAST rows: a tensor the reference holds as (N, C, H, W) is the row-major matrix
 * [N*H*W, C] of 16-bit ELEMENTS here (row = (n, y, x), n = b*F + f).  fp32 is used for all accumulation and statistics.
 *
 * ELEMENT TYPE.  The same sources and the same ABI are built twice:
 *   libctrlv_hip.so      elements are bf16  (ctrlv_elem_dtype() == 2) -- BASELINE.json's dtype; inference + training step
 *   libctrlv_hip_f16.so  elements are fp16  (ctrlv_elem_dtype() == 1) -- the dtype the reference itself evaluates in (fp16
 *                        autocast: config/a100l.yaml:9, tools/eval_video_controlnet.py:110-118); inference plans of fp16 models
 * Wherever this header says "bf16" for an activation / packed weight / residual buffer, read "the library's element
 * type".  The dtype CODES of the model boundary (0 fp32, 1 fp16, 2 bf16: sample, ehs, parameters, outputs) mean the same
 * in both libraries.  A process may load both (the host layer selects by the model's dtype).
 */
#ifndef CTRLV_HIP_H
#define CTRLV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ctrlv_stream_t; /* hipStream_t */

enum {
  CTRLV_OK = 0,
  CTRLV_E_BAD_ARG = -1,
  CTRLV_E_BAD_SHAPE = -2,
  CTRLV_E_HIP = -3,
  CTRLV_E_BAD_DTYPE = -4,  /* a dtype code outside {0 fp32, 1 fp16, 2 bf16} at the model boundary */
  CTRLV_E_WORKSPACE = -5,  /* caller-supplied workspace smaller than ctrlv_plan_workspace_bytes() */
};

/* Library ABI version (bumped on any signature change). */
int ctrlv_abi_version(void);
/* dtype code (1 fp16 / 2 bf16) of the element type this library was built for (see above). */
int ctrlv_elem_dtype(void);
/* Copies the build id (hash of the kernel sources + this header the library was compiled from) into buf; returns its
 * length.  The Python host layer compares it with the sources it sits next to and refuses a stale library. */
int ctrlv_build_id(char* buf, size_t n);
/* Copies the last error message of the calling thread into buf (NUL terminated); returns its length. */
int ctrlv_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------------------------------------------------------
 * Gather-GEMM:  out[m, :] = epilogue( sum_taps A_tap[m, :] . W[:, tap, :]^T )
 *
 * One kernel family replaces every dense contraction of the path:
 *   mode 0  nn.Linear / 1x1 Conv2d ........ to_q/k/v/out, proj_in/out, GEGLU FF, time MLPs, conv_shortcut,
 *                                            controlnet_down_blocks[i] / controlnet_mid_block zero-convs
 *                                            (src/ctrlv/models/controlnet.py:148-185,331-344; `conditioning_scale`
 *                                            of :343-344 is folded into s_acc)
 *   mode 1  Conv2d 3x3 pad 1 .............. ResnetBlock2D.conv1/conv2, Downsample2D (stride 2), Upsample2D
 *                                            (nearest x2 fused: up=1), conv_in / conv_out
 *                                            (unet_spatio_temporal_condition.py:97,163; controlnet.py:297-298)
 *   mode 2  Conv3d (3,1,1) pad (1,0,0) .... TemporalResnetBlock.conv1/conv2 (3 taps along the frame axis)
 *
 * A, A2, R1, R2, out(bf16) are bf16; W is bf16 [N][taps*Cin] (K contiguous, tap-major); bias, V are fp32.
 * Epilogue, in fp32:  v = s_acc*(acc + bias[n]) + s1*R1[m,n] + s2*R2[m,n] + V[vidx(m), n];  v = silu(v) if act;
 * GEGLU (geglu=1): weight rows are interleaved in blocks of 16 (value block, gate block); out[m, j] = a_j * gelu_erf(g_j)
 * (erf-GELU x*Phi(x) evaluated in fp32 by a polynomial, |absolute error| <= 2.1e-6 -- csrc/common.h),
 * out has N/2 columns.  This fuses AlphaBlender, residual adds, the temb broadcast add, the frame
 * positional embedding and the degenerate 1-key CLIP cross-attention (a row vector per clip) into the GEMM.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ctrlv_gemm_desc {
  const void* A;      /* [rows, lda] bf16 */
  const void* A2;     /* optional second source for channels >= c_split (skip-concat, torch.cat dim=1) or NULL;
                         every mode accepts it (fast path: plain GEMM with a bias-only epilogue, the 1x1 shortcuts) */
  const void* W;      /* [N, taps*Cin] bf16 */
  void* out;          /* [M, ldo] bf16 (or fp32 if out_f32) */
  const float* bias;  /* [N] or NULL */
  const void* R1;     /* [M, ldr1] bf16 or NULL */
  const void* R2;     /* [M, ldr2] bf16 or NULL */
  const float* V;     /* [*, ldv] fp32 row-vector table or NULL */
  int32_t M, N, Cin, taps;
  int32_t lda, lda2, c_split;
  int32_t mode;                       /* 0 plain, 1 conv2d 3x3, 2 temporal (3,1,1) */
  int32_t H, Wd, Ho, Wo, stride, up;  /* mode 1: input H x Wd (before upsample), output Ho x Wo */
  int32_t F, S;                       /* mode 2: frames per clip, pixels per frame.  mode 0: S = optional rows-per-image
                                         hint (0 = unknown), the shape key of the split plan (ctrlv_gemm_splitk_ws_bytes) */
  int32_t ldo, n_store;               /* output leading dimension; columns >= n_store are not written */
  int32_t ldr1, ldr2;
  float s_acc, s1, s2;
  int32_t vmode;                      /* 0 none; 1: vidx = (m / vdiv) % vmod; 2: vidx = ((m / vdiv) * vS + m % vS) % vmod */
  int32_t vdiv, vmod, vS, ldv;
  int32_t act;                        /* 0 none, 1 SiLU */
  int32_t geglu;
  int32_t out_f32;
  int32_t tile;                       /* 0 auto, else forces a tile configuration (testing) */
  int32_t ld_raw;                     /* leading dimension of raw_out (elements) */
  void* raw_out;                      /* GEGLU only, optional (training forward): the projection BEFORE the gate,
                                         [M, ld_raw] bf16 in the packed (16 value | 16 gate) column-block order -- what
                                         ctrlv_geglu_bwd consumes -- written by the same launch (ping-pong tiles) */
  int32_t n_scale2;                   /* output columns [0, n_scale2) take s_acc2 in place of s_acc (multiple of 32; 0 = */
  float s_acc2;                       /* none).  The fused q|k|v projection scales its q block by (1/sqrt(64)) log2(e)
                                         here, in fp32 before the one bf16 rounding: ctrlv_attention_spatial_prescaled */
  float* gn_partials;                 /* optional: GroupNorm(32) chunk partials of `out`, written by the same launch --
                                         [M / 64][32 groups][mean, M2] fp32 over 64-row chunks (what the statistics pass
                                         of ctrlv_groupnorm would compute by re-reading `out`); consumed by
                                         ctrlv_groupnorm_from_partials.  Only where ctrlv_gemm_gn_partials_serves(d) */
  void* splitk_ws;                    /* optional scratch of ctrlv_gemm_splitk_ws_bytes(d) bytes: lets ctrlv_gemm split the
                                         contraction of a small-image, long-K conv into K slices (one launch for all
                                         slices + a streaming sum / epilogue kernel).  NULL = never split */
  int32_t ksplit, w_cin;              /* internal (set by ctrlv_gemm for its slice launch; callers leave 0): number of K
                                         slices; channels per tap of W when Cin is a slice's channel count */
  /* SPLIT ("fp16x2") residual trunk -- libctrlv_hip_f16.so only (round 5).  A tensor of the residual stream may carry a
   * second plane of the same shape and pitch IN ELEMENTS: hi = rne_fp16(v), lo = rne_e5m2(v - hi) -- ABI 20: the lo plane
   * holds ONE BYTE per element (e5m2 / "bf8": fp16's sign and exponent, two mantissa bits; row pitch in bytes = the hi
   * plane's pitch in elements), hi + lo keeps ~15 significant bits in 3 bytes per element (ABI 18-19: a 16-bit lo plane; on
   * the oracle both give the same model-level error, tests/trunk_precision_study.py).  R1_lo / R2_lo (optional, need R1 /
   * R2): the operand is R1 + R1_lo (R2 + R2_lo); out_lo (optional): `out` receives hi, out_lo receives lo.  MFMA operands
   * (A, A2) always read the hi plane: it is the element-rounded tensor.  Not with GEGLU / SiLU / fp32 output / raw_out. */
  const void* R1_lo;
  const void* R2_lo;
  void* out_lo;
} ctrlv_gemm_desc;

int ctrlv_gemm(const ctrlv_gemm_desc* d, ctrlv_stream_t stream);
/* Bytes of splitk_ws a launch of `d` wants (0: the launch is not split).  A function of the layer's shape -- pixels per
 * image, N, Cin, taps -- and of M only through the size of the scratch: a clip is computed with the same summation order
 * alone and in a batch.  A caller that passes the scratch for one batch size must pass it for every batch size. */
size_t ctrlv_gemm_splitk_ws_bytes(const ctrlv_gemm_desc* d);
/* 1 if a launch of `d` can write gn_partials: a 3x3 conv (row-halo eligible geometry) or a temporal conv with a {V} or
 * {R1} epilogue (ResnetBlock2D.conv1 / conv2, TemporalResnetBlock.conv1 / conv2: the producers of norm2, temporal norm1,
 * temporal norm2 and -- conv2 with the AlphaBlender epilogue -- of the GroupNorm that opens the transformer behind the
 * block), N = 320 / 640 / 1280, image size a multiple of 64 pixels.  Depends on the layer's shape only, never on the
 * batch size. */
int ctrlv_gemm_gn_partials_serves(const ctrlv_gemm_desc* d);

/* ------------------------------------------------------------------------------------------------------------------
 * GroupNorm(32) (+SiLU), channels-last.  Replaces nn.GroupNorm + SiLU of ResnetBlock2D.norm1/norm2,
 * TemporalResnetBlock.norm1/norm2 (5-D: statistics over (C/32, F, H, W)), TransformerSpatioTemporalModel.norm and
 * conv_norm_out (unet_spatio_temporal_condition.py:161-162).
 *   x  : [n_img*S, C] bf16, optionally the channel-concat of (x [.., c_split], x2 [.., C-c_split]) (torch.cat dim=1)
 *   statistics are taken per (img / imgs_per_stat, group): imgs_per_stat = 1 (4-D) or F (5-D)
 *   stats pass writes per-chunk fp32 (mean, M2 = sum (x - mean)^2) per group (accumulated about per-channel pilot
 *   values, so |mean| >> std does not cancel), then combines them in fp64 into (mean, rstd) per (statistics row, group),
 *   stored behind the chunk partials; the apply pass streams y = [silu](x * a_c + b_c).
 * `partials` must hold (n_img * n_chunks + n_img / imgs_per_stat) * 64 floats, n_chunks = ctrlv_groupnorm_chunks().
 * ------------------------------------------------------------------------------------------------------------------ */
int ctrlv_groupnorm_chunks(int n_img, int S, int C, int imgs_per_stat);
int ctrlv_groupnorm_stats(const void* x, const void* x2, int c_split, int n_img, int S, int C, int imgs_per_stat,
                          float eps, float* partials, ctrlv_stream_t stream);
int ctrlv_groupnorm_apply(const void* x, const void* x2, int c_split, int n_img, int S, int C, int imgs_per_stat,
                          const float* partials, const float* gamma, const float* beta, int silu, void* y,
                          ctrlv_stream_t stream);
/* GroupNorm(+SiLU) of a tensor whose producer wrote the chunk partials (ctrlv_gemm_desc.gn_partials): combines them
 * into (mean, rstd) and streams y -- 1 read + 1 write of the tensor.  `partials` holds (n_img * S / 64 + n_img /
 * imgs_per_stat) * 64 floats (the producer's part first); S must be a multiple of 64. */
int ctrlv_groupnorm_from_partials(const void* x, int n_img, int S, int C, int imgs_per_stat, float eps, float* partials,
                                  const float* gamma, const float* beta, int silu, void* y, ctrlv_stream_t stream);
/* The same on a SPLIT tensor (ABI 18): x_lo = the lo plane the producing launch wrote beside `x` (ctrlv_gemm_desc.out_lo together
 * with gn_partials: the {R1} row-halo 3x3 and temporal convs, ctrlv_gemm_gn_partials_serves); NULL = plain. */
int ctrlv_groupnorm_from_partials_split(const void* x, const void* x_lo, int n_img, int S, int C, int imgs_per_stat, float eps,
                                        float* partials, const float* gamma, const float* beta, int silu, void* y,
                                        ctrlv_stream_t stream);
/* The statistics and apply passes on a SPLIT input (ctrlv_gemm_desc.out_lo): x_lo / x2_lo are the lo planes of x / x2 (same shapes and
 * pitches; either may be NULL = that half has no lo plane); the normalised value is x + x_lo.  y is a plain tensor. */
int ctrlv_groupnorm_stats_split(const void* x, const void* x_lo, const void* x2, const void* x2_lo, int c_split, int n_img,
                                int S, int C, int imgs_per_stat, float eps, float* partials, ctrlv_stream_t stream);
int ctrlv_groupnorm_apply_split(const void* x, const void* x_lo, const void* x2, const void* x2_lo, int c_split, int n_img,
                                int S, int C, int imgs_per_stat, const float* partials, const float* gamma,
                                const float* beta, int silu, void* y, ctrlv_stream_t stream);

/* LayerNorm over the channel axis of [M, C] bf16 rows (BasicTransformerBlock.norm1/3,
 * TemporalBasicTransformerBlock.norm_in/1/3).  If V != NULL, normalises x[m,:] + V[(m / vdiv) % vmod, :]
 * (the frame positional embedding add of TransformerSpatioTemporalModel). */
int ctrlv_layernorm(const void* x, int M, int C, const float* gamma, const float* beta, float eps,
                    const float* V, int vdiv, int vmod, int ldv, void* y, ctrlv_stream_t stream);
/* ... of a SPLIT input: normalises x + x_lo (+ V) (x_lo: the lo plane of x, ctrlv_gemm_desc.out_lo). */
int ctrlv_layernorm_split(const void* x, const void* x_lo, int M, int C, const float* gamma, const float* beta, float eps,
                          const float* V, int vdiv, int vmod, int ldv, void* y, ctrlv_stream_t stream);

/* Fused feed-forward pair at C = 320:  out = epilogue( GEGLU(x . W1^T + b1) . W2^T )  with the second projection's
 * epilogue operands taken from `out_desc` (out, bias = b2, R1 / R2 / V, s_acc, s1, s2, M, N = 320, Cin = 1280, mode 0):
 * the two ctrlv_gemm launches of BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff_in / .ff [DIFF-0.27.2
 * attention.py FeedForward(GEGLU)] without the 4C-wide intermediate in HBM.  x: bf16 [M][ldx]; w1f / w2f: the
 * fragment-major forms written by ctrlv_ff_fused_pack from the packed weights ([2560][320] in the GEGLU-interleaved row
 * order, [320][1280]) and b1 (fp32 [2560], the same row order; it travels inside w1f as a 21st K step).  w1f holds
 * ctrlv_ff_fused_w1f_bytes() bytes, w2f 320 * 1280 * 2.  Results: the arithmetic of the two launches up to the
 * summation order of the second projection and 2^-17 |b1| (csrc/ff_fused.hip). */
int ctrlv_ff_fused_w1f_bytes(void);
int ctrlv_ff_fused_pack(const void* w1_packed, const float* b1, const void* w2_packed, void* w1f, void* w2f,
                        ctrlv_stream_t stream);
int ctrlv_ff_fused(const void* x, int ldx, const void* w1f, const void* w2f, const ctrlv_gemm_desc* out_desc,
                   ctrlv_stream_t stream);
/* The same with the LayerNorm in front of the feed-forward folded in (norm3 / norm_in of the transformer blocks):
 * the kernel's input rows are x' = LayerNorm(x + ln_V[(m / ln_vdiv) % ln_vmod]) * gamma + beta, rounded to bf16 like
 * ctrlv_layernorm's output (ln_V may be null; ln_gamma = null: no LayerNorm, = ctrlv_ff_fused). */
int ctrlv_ff_fused_ln(const void* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                      const float* ln_V, int ln_vdiv, int ln_vmod, int ln_ldv, const void* w1f, const void* w2f,
                      const ctrlv_gemm_desc* out_desc, ctrlv_stream_t stream);
/* 1 if ctrlv_ff_fused serves this second-projection descriptor with input rows of pitch ldx -- N = 320, Cin = 1280,
 * n_store 320, every operand within 32-bit byte offsets, pitches multiples of 8; epilogue bias, +R1 or +R1+R2; a row-vector
 * operand (vmode 1 / 2) with bias or +R1, and with +R1+R2 only in the per-tile form (vmode 1, vdiv a multiple of 256,
 * s_acc == 1) -- 0 = use the two ctrlv_gemm launches.  The launcher applies exactly these conditions. */
int ctrlv_ff_fused_serves(const ctrlv_gemm_desc* out_desc, int ldx);

/* Fused temporal self-attention block at C = 320 (ABI 18; csrc/temporal_fused.hip):
 *     out = R1 + to_out( softmax_f( q k^T / 8 ) v ) + bias + V[clip],     (q | k | v) = x . W_qkv^T
 * over the F <= 32 frames of every pixel -- `TemporalBasicTransformerBlock.attn1` with its residual and the one-key
 * cross-attention vector [DIFF-0.27.2; instantiated by get_down_block / UNetMidBlockSpatioTemporal,
 * src/ctrlv/models/controlnet.py:157-170,186-192] in ONE launch instead of ctrlv_gemm (q|k|v) + ctrlv_attention_temporal +
 * ctrlv_gemm (to_out) [+ ctrlv_layernorm with ln_gamma]: neither the 3C-wide q|k|v tensor nor the attention output (nor the
 * normalised rows) reaches HBM.  x = LayerNorm(R1) rows -- or, with ln_gamma / ln_beta, the raw rows (= R1) --
 * [B F S][ldx] ordered (b, f, s); wf = ctrlv_temporal_fused_weight_bytes() bytes written by ctrlv_temporal_fused_pack from the
 * packed [3C][ld_qkv] q|k|v weight and the packed [C][ld_o] output projection; bias fp32 [C] or NULL; V: row-vector table
 * as in ctrlv_gemm_desc (vmode 1: vdiv = F S -- one row per clip; vmode 2: also vS = S); R1_lo / out_lo: SPLIT trunk planes
 * (fp16 element library).  Results: the arithmetic of the three launches up to summation orders (same rounding points:
 * q, k, v, P and the attention output are rounded to the element type). */
typedef struct ctrlv_temporal_fused_desc {
  const void* x; int32_t ldx;
  const void* wf;
  const float* bias;
  const void* R1; const void* R1_lo; int32_t ldr1;
  const float* V; int32_t vmode, vdiv, vmod, vS, ldv;
  void* out; void* out_lo; int32_t ldo;
  int32_t B, F, S, C;
  const float* ln_gamma; const float* ln_beta; float ln_eps;   /* optional: x holds the RAW rows and the kernel normalises them
                                                                  first (LayerNorm over C, ctrlv_layernorm's arithmetic and
                                                                  rounding; with a split trunk x is the hi plane: the branch
                                                                  input is the element-rounded value, the residual operand
                                                                  R1 + R1_lo stays exact).  NULL: x is used as it is */
} ctrlv_temporal_fused_desc;
size_t ctrlv_temporal_fused_weight_bytes(void);
int ctrlv_temporal_fused_pack(const void* wqkv_packed, int ld_qkv, const void* wo_packed, int ld_o, void* wf,
                              ctrlv_stream_t stream);
int ctrlv_temporal_fused(const ctrlv_temporal_fused_desc* d, ctrlv_stream_t stream);
/* 1 if ctrlv_temporal_fused serves this descriptor (C = 320, F <= 32, pitches multiples of 8, 32-bit byte offsets, a per-clip
 * row vector) -- 0 = use the three launches.  The launcher applies exactly these conditions. */
int ctrlv_temporal_fused_serves(const ctrlv_temporal_fused_desc* d);

/* Row softmax of fp32 scores into bf16 probabilities: probs[r, :cols] = softmax(scores[r, :cols]) (cols a multiple of 4,
 * <= 16384).  The VAE mid block's single-head attention (head dim 512; AutoencoderKLTemporalDecoder, called by
 * pipeline_video_control.py:235,278,346) runs as scores GEMM -> this -> P.V GEMM. */
int ctrlv_softmax_rows(const float* scores, int rows, int cols, long ld_scores, void* probs, long ld_probs,
                       ctrlv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Self-attention cores (diffusers AttnProcessor2_0 = F.scaled_dot_product_attention, head_dim 64, no mask).
 * qkv: [rows, 3*C] bf16 with q | k | v column blocks (fused to_q/to_k/to_v output), heads = C/64; out: [rows, C].
 *   spatial : rows = n_img*S, attention over the S tokens of each image (BasicTransformerBlock.attn1)
 *   temporal: rows = B*F*S ordered (b, f, s); attention over the F frames of each (b, s) -- the
 *             (b f) s c <-> (b s) f c permutes of TemporalBasicTransformerBlock are index math, F <= 32.
 * ------------------------------------------------------------------------------------------------------------------ */
int ctrlv_attention_spatial(const void* qkv, void* out, int n_img, int S, int C, ctrlv_stream_t stream);
/* The same core for q columns that are ALREADY scaled by (1/sqrt(64)) * log2(e) = 0.18033688 (ctrlv_gemm_desc.n_scale2 /
 * s_acc2 on the q|k|v projection): softmax_2(q' k^T) v.  The scaled scores then come out of the K.Q^T MFMAs and the
 * running max is subtracted through their C operand -- no per-score multiply-add on the vector ALU (the inference
 * plans use this entry; training keeps the unscaled form and its LSE convention). */
int ctrlv_attention_spatial_prescaled(const void* qkv, void* out, int n_img, int S, int C, ctrlv_stream_t stream);
int ctrlv_attention_temporal(const void* qkv, void* out, int B, int F, int S, int C, ctrlv_stream_t stream);
/* Training forward of the spatial core: additionally writes L = m + log2(l) of every softmax row (log2 domain of the
 * scaled scores), lse [n_img][C/64][S] fp32 (NULL = ctrlv_attention_spatial). */
int ctrlv_attention_spatial_lse(const void* qkv, void* out, float* lse, int n_img, int S, int C, ctrlv_stream_t stream);
/* Backward of the two cores (torch autograd of F.scaled_dot_product_attention, cfg5 training step,
 * tools/train_video_controlnet.py:451-488).  dout [rows, C] bf16 -> dqkv [rows, 3C] bf16 (dq | dk | dv, every column
 * written).  spatial: lse from ctrlv_attention_spatial_lse, delta = ctrlv_attention_bwd_scratch_floats() floats of
 * scratch (receives rowsum(dO * O)).  Deterministic (no atomics). */
size_t ctrlv_attention_bwd_scratch_floats(int n_img, int S, int C);
int ctrlv_attention_spatial_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                float* delta, int n_img, int S, int C, ctrlv_stream_t stream);
int ctrlv_attention_temporal_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, int B, int F, int S, int C,
                                 ctrlv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Element-wise / layout kernels.
 * ------------------------------------------------------------------------------------------------------------------ */
/* NCHW (any of fp32/fp16/bf16: src_dtype 0/1/2) -> channels-last bf16 rows [n_img*HW, ldc], writing channels
 * [c_off, c_off+C) and leaving others untouched (used to build the 8|4|pad conv_in input). */
int ctrlv_nchw_to_rows(const void* src, int src_dtype, int n_img, int C, int HW, void* dst, int ldc, int c_off,
                       ctrlv_stream_t stream);
/* channels-last bf16 rows [n_img*HW, ldc] (first C columns) -> NCHW of dst_dtype (0 fp32, 1 fp16, 2 bf16). */
int ctrlv_rows_to_nchw(const void* src, int ldc, int n_img, int C, int HW, void* dst, int dst_dtype,
                       ctrlv_stream_t stream);
/* The (3, 1, 1) time convolution that ends AutoencoderKLTemporalDecoder.decode (`time_conv_out`, C -> C channels, C <= 4,
 * zero padding over the frames of ONE clip; pipeline_video_control.py:346 calls decode per chunk) fused with the
 * rows -> NCHW conversion: src = conv_out rows [n_frames*HW, ldc] bf16, weight fp32 [C][C][3] (out, in, tap), bias
 * fp32 [C]; dst (n_frames, C, H, W) in dst_dtype. */
int ctrlv_time_conv_rows_to_nchw(const void* src, int ldc, int n_frames, int C, int HW, const float* weight,
                                 const float* bias, void* dst, int dst_dtype, ctrlv_stream_t stream);
/* im2col for the tiny-channel 3x3 input convs (conv_in, control_conv_in): rows [n_img*H*W, Cp] -> [.., 9*Cp (+pad to Kp)] */
int ctrlv_im2col3x3(const void* x, int n_img, int H, int W, int Cp, void* col, int Kp, ctrlv_stream_t stream);
/* y = a*x + b*r  (bf16 rows, n elements) -- the ControlNet residual add of
 * unet_spatio_temporal_condition.py:119-127,136-137. */
int ctrlv_axpby(const void* x, const void* r, float a, float b, void* y, size_t n, ctrlv_stream_t stream);
/* (y, y_lo) = split(a * (x + x_lo) + b * r): the same add on a SPLIT skip tensor (x_lo may be NULL; y_lo receives the lo
 * plane of the result; in place allowed). */
int ctrlv_axpby_split(const void* x, const void* x_lo, const void* r, float a, float b, void* y, void* y_lo, size_t n,
                      ctrlv_stream_t stream);
/* Sinusoidal `Timesteps` (flip_sin_to_cos=True, shift 0, max_period 1e4): t[n] -> out[n, dim] = [cos | sin], bf16. */
int ctrlv_timestep_embedding(const float* t, int n, int dim, void* out, ctrlv_stream_t stream);
/* y = silu(x) on bf16 (the SiLU in front of every time_emb_proj). */
int ctrlv_silu(const void* x, void* y, size_t n, ctrlv_stream_t stream);
/* Fused CFG combine + Euler (v-prediction) update of pipeline_video_control.py:327-332:
 *   v = uncond + g[f]*(cond - uncond);  x0 = v*(-sigma/sqrt(sigma^2+1)) + x/(sigma^2+1);  x += (x - x0)/sigma*(sigma_next-sigma)
 * latents fp32 [B, F, CHW]; noise_pred bf16/fp32 [(2)B, F, CHW] (uncond first); also writes the next scaled model
 * input (x_next / sqrt(sigma_next^2+1)) as bf16 into `scaled_next` if not NULL. */
int ctrlv_cfg_euler_step(float* latents, const void* noise_pred, int pred_dtype, int cfg, const float* guidance,
                         int B, int F, int CHW, float sigma, float sigma_next, void* scaled_next,
                         ctrlv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward kernels: first slice of the training step (tools/train_video_controlnet.py:451-488 -- ControlNet dgrad +
 * wgrad, UNet up-path dgrad).  DGRAD of the gather-GEMM family is ctrlv_gemm itself on
 * role-swapped weights (ctrlv_amd/autograd.py); these are the reductions the forward kernels cannot express.
 * ------------------------------------------------------------------------------------------------------------------ */
/* dW[n][tap*Cin + c] += sum_m dY[m][n] * A[src(m, tap)][c]  for the forward GEMM described by `fwd` (A, A2/c_split, M, N,
 * Cin, taps, mode and its geometry are read; W / out / epilogue fields are ignored).  dY: bf16 [M][ldy]; dW: fp32
 * [N][taps*Cin] in the packed (tap-major) K order -- or, with torch_layout = 1, [N][Cin][taps], the layout of the
 * nn.Conv2d / nn.Conv3d / nn.Linear parameter itself -- accumulated with atomics: zero it first.  dbias (fp32 [N], or
 * NULL) += scale * column sums of dY, computed by the workgroups that stream dY anyway; dW is scaled by `scale` too
 * (the forward's s_acc).
 * scratch (ctrlv_gemm_wgrad_scratch_bytes(fwd) bytes, or NULL): with it the launch is DETERMINISTIC -- the row slabs write
 * fp32 partial matrices with plain stores and a second kernel adds them to dW / dbias in slab order (a single writer per
 * element): the same gradient bits in every run.  Without it the slabs add with fp32 atomics (arrival order).
 * torch_layout bit 1 (value 2; with scratch only): ASSIGN -- dW / dbias = scale * sum instead of +=, their previous contents
 * are not read (no zero fill in front of the launch). */
size_t ctrlv_gemm_wgrad_scratch_bytes(const ctrlv_gemm_desc* fwd);
int ctrlv_gemm_wgrad(const ctrlv_gemm_desc* fwd, const void* dY, int ldy, float* dW, float* dbias, float scale,
                     int torch_layout, void* scratch, size_t scratch_bytes, ctrlv_stream_t stream);
/* One-kernel packing of a PyTorch-layout parameter (fp32 / fp16 / bf16: src_dtype 0 / 1 / 2; [N][C][taps] contiguous:
 * nn.Linear taps 1, Conv2d 3x3 taps 9, Conv3d (3,1,1) taps 3) into the bf16 GEMM layouts -- what a training step does
 * to every trainable weight every step.  form 0: the forward layout dst[n][tap*C + c] (geglu: rows in the packed
 * (value, gate) block order); form 1: the role-swapped dgrad layout dst[c][(taps-1-tap)*Np + n], Np = ld_dst / taps
 * (columns n >= N zero).  Rows of dst beyond those written (row padding to 32) must be zeroed by the caller. */
int ctrlv_pack_weight(const void* src, int src_dtype, int N, int C, int taps, int form, int geglu, void* dst, int ld_dst,
                      ctrlv_stream_t stream);
/* out[idx(m)][n] += scale * x[m][n] summed over rows; idx = 0 (vmode 0: bias gradient) or (m / vdiv) % vmod (vmode 1: the
 * gradient of a per-clip row-vector operand V).  x bf16 [M][ldx], out fp32 [*][ldo], accumulated -- with atomics, or,
 * given `scratch` (ctrlv_colsum_scratch_floats floats; 0 = this row grouping has no deterministic form), per-block sums
 * added in block order: bit-reproducible. */
size_t ctrlv_colsum_scratch_floats(int M, int N, int vmode, int vdiv);
int ctrlv_colsum(const void* x, int M, int N, int ldx, int vmode, int vdiv, int vmod, float scale, float* out, int ldo,
                 float* scratch, ctrlv_stream_t stream);
/* out[0] += scale * sum_i dy[i] * (p[i] - q[i]) over n bf16 elements: the gradient of a folded AlphaBlender's mixing
 * weight (out = xs + (1 - a) * h  =>  dL/da = -sum dy * (out - xs) / (1 - a)).  scratch: 1024 floats (deterministic: block
 * sums added in order) or NULL (one atomic per block). */
int ctrlv_dot_diff(const void* dy, const void* p, const void* q, size_t n, float scale, float* out, float* scratch,
                   ctrlv_stream_t stream);
/* GroupNorm(32)(+SiLU) backward on channels-last rows.  `fwd_partials` is the buffer ctrlv_groupnorm_stats filled for
 * the same x (its (mean, rstd) table is reused); dx bf16; dgamma / dbeta fp32 [C], ACCUMULATED (ordered two-level sums, no
 * atomics: bit-reproducible); scratch fp32 of ctrlv_groupnorm_bwd_scratch_floats() elements. */
int ctrlv_groupnorm_bwd_scratch_floats(int n_img, int S, int C, int imgs_per_stat);
int ctrlv_groupnorm_bwd(const void* x, const void* dy, int n_img, int S, int C, int imgs_per_stat,
                        const float* fwd_partials, const float* gamma, const float* beta, int silu, void* dx,
                        float* dgamma, float* dbeta, float* scratch, ctrlv_stream_t stream);
/* ... + add (element rows like dy, or NULL): dx = (norm backward) + add in the same pass.  x feeds the norm that opens a
 * residual branch AND the skip connection around it; `add` is the gradient that arrives through the skip (what autograd would
 * otherwise sum in a separate pass over both tensors: 181 such sums per cfg5 step).  ABI 19. */
int ctrlv_groupnorm_bwd_add(const void* x, const void* dy, const void* add, int n_img, int S, int C, int imgs_per_stat,
                            const float* fwd_partials, const float* gamma, const float* beta, int silu, void* dx,
                            float* dgamma, float* dbeta, float* scratch, ctrlv_stream_t stream);

/* LayerNorm backward over [M, C] rows (optionally of x + V[(m / vdiv) % vmod], as the forward): dx bf16; dgamma / dbeta
 * fp32 [C], ACCUMULATED; scratch = ctrlv_layernorm_bwd_scratch_floats(M, C) floats (per-wave column partials, folded by a
 * second kernel).  The gradient of V's rows is ctrlv_colsum(dx, vmode 1). */
size_t ctrlv_layernorm_bwd_scratch_floats(int M, int C);
int ctrlv_layernorm_bwd(const void* x, const void* dy, int M, int C, const float* gamma, float eps, const float* V, int vdiv,
                        int vmod, int ldv, void* dx, float* dgamma, float* dbeta, float* scratch, ctrlv_stream_t stream);
/* ... + add: as ctrlv_groupnorm_bwd_add.  ABI 19. */
int ctrlv_layernorm_bwd_add(const void* x, const void* dy, const void* add, int M, int C, const float* gamma, float eps,
                            const float* V, int vdiv, int vmod, int ldv, void* dx, float* dgamma, float* dbeta, float* scratch,
                            ctrlv_stream_t stream);
/* GEGLU backward.  raw: the projection output [M, 2I] bf16 in the packed (16 value, 16 gate) column-block order, i.e.
 * ctrlv_gemm on the GEGLU-packed weight with geglu = 0; du: [M, I] bf16; draw: [M, 2I] bf16, same layout as raw. */
int ctrlv_geglu_bwd(const void* raw, const void* du, size_t M, int I, void* draw, ctrlv_stream_t stream);

/* ==================================================================================================================
 * Plan-level entry points: one call = one model forward.  (SURVEY.md 8b "what a C-ABI replacement must export".)
 *
 * A plan is the execution plan of ONE model instance -- the reference's `UNetSpatioTemporalConditionModel`
 * (src/ctrlv/models/unet_spatio_temporal_condition.py:13-171) or `ControlNetModel` (src/ctrlv/models/controlnet.py:20-351):
 * its module graph (built from the diffusers-style config), its weights packed once into the kernels' layouts, and the
 * walk over the layer list that issues the kernels above.  A host in any language runs a forward with
 *     ctrlv_plan_create -> ctrlv_plan_load_weights -> ctrlv_plan_workspace_bytes -> ctrlv_{unet,controlnet}_forward.
 * Ownership: packed weights are library-owned device memory (allocated by load_weights, freed by destroy); every
 * input / output / workspace buffer is the caller's; a forward allocates nothing, never synchronises the host and
 * enqueues everything on the caller's stream (HIP-graph capturable).  A plan is re-entrant across plans, not
 * thread-safe on one plan.
 * ================================================================================================================== */
#define CTRLV_MAX_BLOCKS 8
typedef struct ctrlv_model_config {
  int32_t kind;                 /* 0 = UNetSpatioTemporalConditionModel, 1 = ControlNetModel */
  int32_t in_channels;          /* 8 (4 noisy + 4 image latents); ControlNet: control_cond has in_channels / 2 */
  int32_t out_channels;         /* UNet: 4 */
  int32_t n_blocks;             /* len(down_block_types) == len(up_block_types) */
  int32_t block_out_channels[CTRLV_MAX_BLOCKS];
  int32_t down_cross_attn[CTRLV_MAX_BLOCKS]; /* 1: CrossAttnDownBlockSpatioTemporal, 0: DownBlockSpatioTemporal */
  int32_t up_cross_attn[CTRLV_MAX_BLOCKS];   /* 1: CrossAttnUpBlockSpatioTemporal, 0: UpBlockSpatioTemporal (UNet) */
  int32_t layers_per_block[CTRLV_MAX_BLOCKS];
  int32_t num_attention_heads[CTRLV_MAX_BLOCKS];   /* head_dim = channels / heads must be 64 */
  int32_t cross_attention_dim;                     /* 1024 */
  int32_t addition_time_embed_dim;                 /* 256 */
  int32_t projection_class_embeddings_input_dim;   /* 768 */
  int32_t num_frames;                              /* frames the frame-embedding table is prepared for (others work) */
  int32_t time_context_order;   /* 0: "sb" = diffusers 0.27.2 temporal-context ordering quirk, 1: "bs" */
} ctrlv_model_config;

typedef struct ctrlv_tensor_desc {
  const char* name;             /* diffusers state-dict key, e.g. "down_blocks.0.resnets.0.spatial_res_block.conv1.weight" */
  const void* data;             /* contiguous, row-major, in the PyTorch parameter layout */
  int32_t dtype;                /* 0 fp32, 1 fp16, 2 bf16 */
  int32_t on_device;            /* 1: device pointer (of the plan's device), 0: host pointer */
  int64_t numel;
} ctrlv_tensor_desc;

typedef struct ctrlv_plan ctrlv_plan;

int ctrlv_plan_create(const ctrlv_model_config* cfg, int device, ctrlv_plan** out);
/* All parameters of the model by their diffusers key names (extra names are ignored, a missing one is an error).
 * Packs into the kernel layouts: K-contiguous bf16 weights [N32][K64], q|k|v fused, GEGLU rows interleaved in 16-row
 * (value, gate) blocks, every time_emb_proj and every cross-attention to_v concatenated into one GEMM each.
 * May be called again (e.g. after an optimizer step); synchronises the device. */
int ctrlv_plan_load_weights(ctrlv_plan* plan, const ctrlv_tensor_desc* tensors, size_t n);
/* Switch the temporal cross-attention context order of an existing plan (0 "sb" / 1 "bs", see the config). */
int ctrlv_plan_set_time_context_order(ctrlv_plan* plan, int order);
/* Storage of the RESIDUAL TRUNK of the plan's forwards (conv_in output, every block / AlphaBlender output, the skip
 * tensors, the tensors the ControlNet residuals are added into): 0 = one element per value like every other activation
 * (default), 1 = SPLIT into a hi element plane + a one-byte lo plane (see ctrlv_gemm_desc.out_lo: 3 bytes per element, ~15
 * significant bits under fp16 branches; north_star's 1e-3 model-level tolerance).  Mode 1 needs the fp16 element library; the workspace grows
 * (ctrlv_plan_workspace_bytes answers for the current mode).  Model inputs / outputs / residual tensors across the ABI are
 * unchanged. */
int ctrlv_plan_set_trunk_mode(ctrlv_plan* plan, int mode);
/* Bytes of workspace one forward of (B clips, F frames, H x W latent) needs; 0 on error (see ctrlv_last_error). */
size_t ctrlv_plan_workspace_bytes(ctrlv_plan* plan, int B, int F, int H, int W);
/* Number of residual tensors the ControlNet produces / the UNet consumes on the down path (12 for SVD), and the
 * rows / channels of residual i at (B, F, H, W): i in [0, n) = down residuals, i == n = the mid residual. */
int ctrlv_plan_num_down_residuals(ctrlv_plan* plan);
int ctrlv_plan_residual_shape(ctrlv_plan* plan, int i, int B, int F, int H, int W, int64_t* rows, int32_t* channels);

/* UNetSpatioTemporalConditionModel.forward (unet_spatio_temporal_condition.py:31-171).
 *   sample   (B, F, in_channels, H, W) contiguous, dtype 0/1/2       timestep  device fp32 [1] or [B] (n_timestep)
 *   ehs      (B, 1, cross_attention_dim) contiguous, dtype as sample  added_time_ids device fp32 (B, n_ids)
 *   down_res n pointers (or NULL) to channels-last bf16 rows [B*F*h_i*w_i, C_i] -- the layout
 *            ctrlv_controlnet_forward writes; mid_res likewise (both or neither, :61); they are ADDED to the skip
 *            tensors / the mid block output (:119-127,136-137) and not modified.
 *   residual_event: optional hipEvent_t the stream waits on right before it reads the residuals (so a ControlNet
 *            running on another stream overlaps the UNet's down / mid blocks); NULL = none.
 *   out      (B, F, out_channels, H, W) contiguous, dtype as sample. */
int ctrlv_unet_forward(ctrlv_plan* plan, const void* sample, int dtype, const float* timestep, int n_timestep,
                       const void* ehs, const float* added_time_ids, int n_ids, const void* const* down_res,
                       const void* mid_res, void* residual_event, void* out, int B, int F, int H, int W,
                       void* workspace, size_t workspace_bytes, ctrlv_stream_t stream);
/* The down + mid half of the same forward (unet_spatio_temporal_condition.py:64-117,130-135), for the training step of
 * tools/train_video_controlnet.py:451-466: the frozen UNet's encoder has no gradient path (the ControlNet residuals are
 * added to its OUTPUTS, :119-137), so it runs here and hands the n skip tensors and the mid block output back as
 * channels-last bf16 rows (shapes: ctrlv_plan_residual_shape; out_taps[i] / out_mid are caller-owned). */
int ctrlv_unet_encoder_forward(ctrlv_plan* plan, const void* sample, int dtype, const float* timestep, int n_timestep,
                               const void* ehs, const float* added_time_ids, int n_ids, void* const* out_taps,
                               void* out_mid, int B, int F, int H, int W, void* workspace, size_t workspace_bytes,
                               ctrlv_stream_t stream);
/* ControlNetModel.forward (controlnet.py:226-351): control_cond (B, F, in_channels/2, H, W) dtype as sample;
 * writes out_down[i] / out_mid as channels-last bf16 rows (shapes: ctrlv_plan_residual_shape), already multiplied by
 * conditioning_scale (:343-344, folded into the zero-conv epilogue). */
int ctrlv_controlnet_forward(ctrlv_plan* plan, const void* sample, const void* control_cond, int dtype,
                             const float* timestep, int n_timestep, const void* ehs, const float* added_time_ids,
                             int n_ids, float conditioning_scale, void* const* out_down, void* out_mid, int B, int F,
                             int H, int W, void* workspace, size_t workspace_bytes, ctrlv_stream_t stream);
int ctrlv_plan_destroy(ctrlv_plan* plan);

/* Per-launch profile of the plan's OWN launches (measurement aid: bench.py's roofline leg, tools/shape_table.py).  While
 * enabled, every kernel a forward issues is bracketed by HIP events on the launch stream and recorded with its algorithmic
 * FLOPs / bytes (SURVEY.md Appendix B accounting).  Enable OUTSIDE HIP-graph capture and for eager forwards only.
 * ctrlv_plan_profile_read synchronises on the recorded events, writes up to max_records records in launch order, clears
 * the list and returns the number written (max_records == 0: the number pending, nothing cleared); <0 on error. */
enum {
  CTRLV_FAM_GEMM_LINEAR = 0, CTRLV_FAM_GEMM_CONV3X3 = 1, CTRLV_FAM_GEMM_CONV_TEMPORAL = 2, CTRLV_FAM_ATTENTION_SPATIAL = 3,
  CTRLV_FAM_ATTENTION_TEMPORAL = 4, CTRLV_FAM_GROUPNORM = 5, CTRLV_FAM_LAYERNORM = 6, CTRLV_FAM_RESIDUAL_ADD = 7,
  CTRLV_FAM_GEMM_TEMPORAL_BLOCK = 8,    /* ctrlv_temporal_fused: q|k|v projection + attention over the frames + output projection */
};
typedef struct ctrlv_profile_record {
  int32_t family;             /* CTRLV_FAM_* */
  int32_t M, N, K;            /* GEMM: rows, weight rows, taps * Cin; attention: images, tokens, channels; norms: rows, channels */
  int32_t flags;              /* GEMM: bit 0 GEGLU, bits 1-2 residual operands, bits 3-4 vmode, bit 8 fused feed-forward */
  float ms;                   /* elapsed between the launch's two events */
  double flops, bytes;        /* algorithmic work of the launch */
} ctrlv_profile_record;
int ctrlv_plan_profile(ctrlv_plan* plan, int enable);
int ctrlv_plan_profile_read(ctrlv_plan* plan, ctrlv_profile_record* out, int max_records);

#ifdef __cplusplus
}
#endif
#endif /* CTRLV_HIP_H */
