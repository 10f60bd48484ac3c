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
 * [N*H*W, C] of bf16 here (row = (n, y, x), n = b*F + f).  fp32 is used for all accumulation and statistics.
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
};

/* Library ABI version (bumped on any signature change). */
int ctrlv_abi_version(void);
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
  int32_t F, S;                       /* mode 2: frames per clip, pixels per frame */
  int32_t ldo, n_store;               /* output leading dimension; columns >= n_store are not written */
  int32_t ldr1, ldr2;
  float s_acc, s1, s2;
  int32_t vmode;                      /* 0 none; 1: vidx = (m / vdiv) % vmod; 2: vidx = ((m / vdiv) * vS + m % vS) % vmod */
  int32_t vdiv, vmod, vS, ldv;
  int32_t act;                        /* 0 none, 1 SiLU */
  int32_t geglu;
  int32_t out_f32;
  int32_t tile;                       /* 0 auto, else forces a tile configuration (testing) */
} ctrlv_gemm_desc;

int ctrlv_gemm(const ctrlv_gemm_desc* d, ctrlv_stream_t stream);

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

/* LayerNorm over the channel axis of [M, C] bf16 rows (BasicTransformerBlock.norm1/3,
 * TemporalBasicTransformerBlock.norm_in/1/3).  If V != NULL, normalises x[m,:] + V[(m / vdiv) % vmod, :]
 * (the frame positional embedding add of TransformerSpatioTemporalModel). */
int ctrlv_layernorm(const void* x, int M, int C, const float* gamma, const float* beta, float eps,
                    const float* V, int vdiv, int vmod, int ldv, void* y, ctrlv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Self-attention cores (diffusers AttnProcessor2_0 = F.scaled_dot_product_attention, head_dim 64, no mask).
 * qkv: [rows, 3*C] bf16 with q | k | v column blocks (fused to_q/to_k/to_v output), heads = C/64; out: [rows, C].
 *   spatial : rows = n_img*S, attention over the S tokens of each image (BasicTransformerBlock.attn1)
 *   temporal: rows = B*F*S ordered (b, f, s); attention over the F frames of each (b, s) -- the
 *             (b f) s c <-> (b s) f c permutes of TemporalBasicTransformerBlock are index math, F <= 32.
 * ------------------------------------------------------------------------------------------------------------------ */
int ctrlv_attention_spatial(const void* qkv, void* out, int n_img, int S, int C, ctrlv_stream_t stream);
int ctrlv_attention_temporal(const void* qkv, void* out, int B, int F, int S, int C, ctrlv_stream_t stream);

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
/* im2col for the tiny-channel 3x3 input convs (conv_in, control_conv_in): rows [n_img*H*W, Cp] -> [.., 9*Cp (+pad to Kp)] */
int ctrlv_im2col3x3(const void* x, int n_img, int H, int W, int Cp, void* col, int Kp, ctrlv_stream_t stream);
/* y = a*x + b*r  (bf16 rows, n elements) -- the ControlNet residual add of
 * unet_spatio_temporal_condition.py:119-127,136-137. */
int ctrlv_axpby(const void* x, const void* r, float a, float b, void* y, size_t n, ctrlv_stream_t stream);
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

#ifdef __cplusplus
}
#endif
#endif /* CTRLV_HIP_H */
