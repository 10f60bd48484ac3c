// Plan-level C ABI: the execution plan of one UNetSpatioTemporalConditionModel / ControlNetModel instance.
//
//   ctrlv_plan_create        builds the module graph from the diffusers-style config (the wiring of
//                            src/ctrlv/models/controlnet.py:100-195 and of the diffusers parent UNet class)
//   ctrlv_plan_load_weights  takes every parameter by its diffusers state-dict key and packs it ON THE DEVICE into the
//                            layouts the kernels consume (include/ctrlv_hip.h: ctrlv_gemm_desc)
//   ctrlv_{unet,controlnet}_forward   walk the layer list and issue the kernels of this library on the caller's stream,
//                            with every activation in a stack-discipline arena over the caller's workspace
//
// The walk is the C++ twin of ctrlv_amd/models/{blocks,encoder,unet_spatio_temporal_condition,controlnet}.py: same
// kernels, same descriptors, same order -- the two executors produce bit-identical outputs (tests/test_plan_gpu.py).
#include <math.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

#define TRY(expr)                   \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != CTRLV_OK) return rc__; \
  } while (0)

namespace {

// ------------------------------------------------------------------------------------------------ packing kernels
__device__ __forceinline__ float ld_any(const void* p, int dtype, long i) {
  if (dtype == 0) return ((const float*)p)[i];
  if (dtype == 1) return (float)((const _Float16*)p)[i];
  return bf16_to_f32(((const bf16_t*)p)[i]);
}
// GEGLU row interleave (packing.geglu_interleave): rows [a_0..a_{I-1} | g_0..g_{I-1}] -> alternating 16-row blocks
__device__ __forceinline__ int row_map(int n, int N, int geglu) {
  if (!geglu) return n;
  const int inner = N >> 1, isg = n >= inner, r = isg ? n - inner : n;
  return ((r >> 4) * 2 + isg) * 16 + (r & 15);
}
// dst[(n_off + rowmap(n)) * ld + tap * cp + c_off + c] = src[(n * C + c) * taps + tap]   (dst pre-zeroed)
// covers nn.Linear / 1x1 conv (taps 1), Conv2d 3x3 (taps 9, tap = ky*3+kx), Conv3d (3,1,1) (taps 3), the shared
// conv_in | control_conv_in im2col slots (cp = slot width, c_off = slot offset) and row concatenation (n_off).
__global__ void pack_weight_kernel(const void* __restrict__ src, int dtype, int N, int C, int taps, el_t* __restrict__ dst,
                                   int ld, int cp, int c_off, int n_off, int geglu) {
  const long total = (long)N * C * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long nc = i / taps;
    const int c = (int)(nc % C), n = (int)(nc / C);
    dst[(long)(n_off + row_map(n, N, geglu)) * ld + tap * cp + c_off + c] = f32_to_el(ld_any(src, dtype, i));
  }
}
// Role-swapped (dgrad) form of a weight: dst[c][(taps - 1 - tap) * Np + n] = src[(n * C + c) * taps + tap]  -- taps
// reversed, output and input channels transposed (autograd.py::gemm_grads); columns n in [N, Np) are zeros.
// A transpose: 32 output rows n x 64 consecutive (c, tap) source elements per workgroup go through LDS, so that reads are
// contiguous along a source row and writes are 32 consecutive n (64 B) per (c, tap).   grid (ceil(C*taps/64), ceil(Np/32))
__global__ __launch_bounds__(256) void pack_weight_swapped_kernel(const void* __restrict__ src, int dtype, int N, int C,
                                                                  int taps, int Np, el_t* __restrict__ dst, int ld) {
  __shared__ float tile[32][65];
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 32, ktot = C * taps;
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int r = i >> 6, kk = i & 63, n = n0 + r, k = k0 + kk;
    tile[r][kk] = (n < N && k < ktot) ? ld_any(src, dtype, (long)n * ktot + k) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int kk = i >> 5, r = i & 31, k = k0 + kk, n = n0 + r;
    if (k >= ktot || n >= Np) continue;
    const int c = k / taps, tap = k - c * taps;
    dst[(long)c * ld + (long)(taps - 1 - tap) * Np + n] = f32_to_el(tile[r][kk]);
  }
}
// Forward form with coalesced writes: dst[rowmap(n)][tap * C + c] = src[(n * C + c) * taps + tap]   grid (ceil(taps*C/256), N)
__global__ void pack_weight_rows_kernel(const void* __restrict__ src, int dtype, int N, int C, int taps,
                                        el_t* __restrict__ dst, int ld, int geglu) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= taps * C) return;
  const int tap = k / C, c = k - tap * C;
  dst[(long)row_map(n, N, geglu) * ld + k] = f32_to_el(ld_any(src, dtype, ((long)n * C + c) * taps + tap));
}
__global__ void pack_vector_kernel(const void* __restrict__ src, int dtype, int N, float* __restrict__ dst, int n_off,
                                   int geglu, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float v = ld_any(src, dtype, n);
  float* d = dst + n_off + row_map(n, N, geglu);
  *d = accumulate ? *d + v : v;
}
__global__ void arange_kernel(float* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)i;
}
__global__ void expand_f32_kernel(const float* __restrict__ src, int n_src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[n_src == 1 ? 0 : i];
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------ plan data
struct Linear {      // packed GEMM weight: bf16 [n][k] (n % 32 == 0, k % 64 == 0 for taps == 1) + fp32 bias [n] (or null)
  el_t* w = nullptr;
  float* b = nullptr;
  int n = 0, k = 0;  // padded rows, total K (taps * cin)
};
struct Norm { float* g = nullptr; float* b = nullptr; };

struct ResBlock {
  int cin = 0, cout = 0;
  float eps = 1e-6f;
  double alpha = 0.5;     // sigmoid(mix_factor), kept in double: 1 - alpha is formed in double like the Python executor does
  Norm n1, n2, tn1, tn2;
  Linear c1, c2, tc1, tc2, sc;
  bool has_sc = false;
  int temb_off[2] = {0, 0};
  std::string name;
};
struct FeedFwd {
  Linear proj, out;
  el_t* w1f = nullptr;      // C = 320 only: the fragment-major forms of ctrlv_ff_fused (ff_fused.hip), else null
  el_t* w2f = nullptr;
};
struct Transformer {
  int C = 0;
  double alpha = 0.5;
  Norm gn, s_ln1, s_ln3, t_lnin, t_ln1, t_ln3;
  Linear pin, pout, s_qkv, s_o, t_qkv, t_o, tpe1, tpe2;
  FeedFwd s_ff, t_ffin, t_ff;
  el_t* t_wf = nullptr;           // C = 320 only: fragment-major q|k|v + to_out of the temporal attention (temporal_fused.hip)
  int xattn_off[2] = {0, 0};
  float* frame_emb = nullptr;     // [cfg.num_frames][C] fp32, prepared at load time
  std::string name;
};
struct Resample { int C = 0; Linear conv; bool present = false; };
struct DownBlock { std::vector<ResBlock> res; std::vector<Transformer> attn; Resample down; };
struct UpBlock { std::vector<ResBlock> res; std::vector<Transformer> attn; Resample up; };
struct CrossOut { int off, c; Linear to_out; };

}  // namespace

struct ctrlv_plan {
  ctrlv_model_config cfg;
  int device = 0;
  bool loaded = false;
  int trunk_mode = 0;         // 0: plain residual trunk, 1: SPLIT hi + lo planes (ctrlv_plan_set_trunk_mode)
  std::vector<DownBlock> down;
  ResBlock mid_r0, mid_r1;
  Transformer mid_attn;
  std::vector<UpBlock> up;
  Linear cin, te1, te2, ae1, ae2, temb, xv, cout;
  int cin_cp = 0, cin_kp = 0, temb_n = 0, xattn_n = 0;
  std::vector<CrossOut> xouts;
  Norm gno;
  std::vector<Linear> zc;     // controlnet_down_blocks
  Linear zc_mid;
  std::vector<void*> owned;   // every device allocation of load_weights
  struct WsEntry { int B, F, H, W; size_t bytes; };
  std::vector<WsEntry> ws_cache;   // ctrlv_plan_workspace_bytes per input shape (check_workspace)
  // per-launch profile of the plan's OWN launches (ctrlv_plan_profile): HIP events on the launch stream around every
  // kernel the walk issues, with the launch's algorithmic FLOPs / bytes -- bench.py's roofline leg and tools/shape_table.py
  struct ProfRec { ctrlv_profile_record r; hipEvent_t e0, e1; };
  bool profiling = false;
  std::vector<ProfRec> prof;
};

namespace {

// ------------------------------------------------------------------------------------------------ graph construction
void make_res(ResBlock& r, int cin, int cout, float eps, const std::string& name) {
  r.cin = cin; r.cout = cout; r.eps = eps; r.has_sc = cin != cout; r.name = name;
}
void make_tr(Transformer& t, int C, const std::string& name) { t.C = C; t.name = name; }

int build_graph(ctrlv_plan* p) {
  const ctrlv_model_config& c = p->cfg;
  const int n = c.n_blocks;
  CTRLV_CHECK_ARG(n >= 1 && n <= CTRLV_MAX_BLOCKS, "plan: n_blocks=%d out of range", n);
  CTRLV_CHECK_ARG(c.kind == 0 || c.kind == 1, "plan: kind must be 0 (UNet) or 1 (ControlNet)");
  for (int i = 0; i < n; ++i) {
    const int ch = c.block_out_channels[i];
    CTRLV_CHECK_SHAPE(ch > 0 && ch % 32 == 0, "plan: block_out_channels[%d]=%d must be a multiple of 32", i, ch);
    const bool attn = c.down_cross_attn[i] || (c.kind == 0 && c.up_cross_attn[n - 1 - i]) || i == n - 1;   // (mid block)
    if (attn)
      CTRLV_CHECK_SHAPE(c.num_attention_heads[i] > 0 && ch == c.num_attention_heads[i] * 64,
                        "plan: attention kernels are specialised for head_dim 64 (block %d: %d channels / %d heads)", i,
                        ch, c.num_attention_heads[i]);
    CTRLV_CHECK_ARG(c.layers_per_block[i] >= 1, "plan: layers_per_block[%d] must be >= 1", i);
  }
  p->down.resize(n);
  int out_ch = c.block_out_channels[0];
  for (int i = 0; i < n; ++i) {
    const int in_ch = out_ch;
    out_ch = c.block_out_channels[i];
    DownBlock& b = p->down[i];
    const std::string base = "down_blocks." + std::to_string(i);
    // get_down_block: CrossAttnDownBlockSpatioTemporal resnets eps 1e-6, DownBlockSpatioTemporal 1e-5
    const float eps = c.down_cross_attn[i] ? 1e-6f : 1e-5f;
    b.res.resize(c.layers_per_block[i]);
    for (int j = 0; j < c.layers_per_block[i]; ++j)
      make_res(b.res[j], j == 0 ? in_ch : out_ch, out_ch, eps, base + ".resnets." + std::to_string(j));
    if (c.down_cross_attn[i]) {
      b.attn.resize(c.layers_per_block[i]);
      for (int j = 0; j < c.layers_per_block[i]; ++j) make_tr(b.attn[j], out_ch, base + ".attentions." + std::to_string(j));
    }
    b.down.present = i != n - 1;
    b.down.C = out_ch;
  }
  const int cm = c.block_out_channels[n - 1];
  make_res(p->mid_r0, cm, cm, 1e-5f, "mid_block.resnets.0");
  make_res(p->mid_r1, cm, cm, 1e-5f, "mid_block.resnets.1");
  make_tr(p->mid_attn, cm, "mid_block.attentions.0");
  if (c.kind == 0) {
    p->up.resize(n);
    int prev = c.block_out_channels[n - 1];
    for (int i = 0; i < n; ++i) {
      const int oc = c.block_out_channels[n - 1 - i];
      const int ic = c.block_out_channels[n - 1 - (i + 1 < n ? i + 1 : n - 1)];
      const int layers = c.layers_per_block[n - 1 - i] + 1;
      UpBlock& b = p->up[i];
      const std::string base = "up_blocks." + std::to_string(i);
      b.res.resize(layers);
      for (int j = 0; j < layers; ++j) {
        const int skip = j == layers - 1 ? ic : oc;
        const int rin = j == 0 ? prev : oc;
        make_res(b.res[j], rin + skip, oc, 1e-6f, base + ".resnets." + std::to_string(j));
      }
      if (c.up_cross_attn[i]) {
        b.attn.resize(layers);
        for (int j = 0; j < layers; ++j) make_tr(b.attn[j], oc, base + ".attentions." + std::to_string(j));
      }
      b.up.present = i != n - 1;
      b.up.C = oc;
      prev = oc;
    }
  }
  return CTRLV_OK;
}

template <class Fn>
void for_each_res(ctrlv_plan* p, Fn fn) {
  for (auto& b : p->down) for (auto& r : b.res) fn(r);
  fn(p->mid_r0); fn(p->mid_r1);
  for (auto& b : p->up) for (auto& r : b.res) fn(r);
}
template <class Fn>
void for_each_tr(ctrlv_plan* p, Fn fn) {
  for (auto& b : p->down) for (auto& t : b.attn) fn(t);
  fn(p->mid_attn);
  for (auto& b : p->up) for (auto& t : b.attn) fn(t);
}

// ------------------------------------------------------------------------------------------------ weight loading
struct Loader {
  ctrlv_plan* p;
  std::unordered_map<std::string, const ctrlv_tensor_desc*> map;
  std::vector<void*> staged;      // temporary device copies of host tensors
  hipStream_t st = nullptr;

  int find(const std::string& name, const ctrlv_tensor_desc** out, long expect_numel) {
    auto it = map.find(name);
    if (it == map.end()) { ctrlv_set_error("plan_load_weights: missing tensor '%s'", name.c_str()); return CTRLV_E_BAD_ARG; }
    const ctrlv_tensor_desc* t = it->second;
    if (expect_numel >= 0 && t->numel != expect_numel) {
      ctrlv_set_error("plan_load_weights: '%s' has %ld elements, expected %ld", name.c_str(), (long)t->numel, expect_numel);
      return CTRLV_E_BAD_SHAPE;
    }
    *out = t;
    return CTRLV_OK;
  }
  int dev_ptr(const ctrlv_tensor_desc* t, const void** out) {
    if (t->on_device) { *out = t->data; return CTRLV_OK; }
    const size_t es = t->dtype == 0 ? 4 : 2;
    void* d = nullptr;
    CTRLV_HIP_TRY(hipMalloc(&d, t->numel * es));
    staged.push_back(d);
    CTRLV_HIP_TRY(hipMemcpy(d, t->data, t->numel * es, hipMemcpyHostToDevice));
    *out = d;
    return CTRLV_OK;
  }
  int alloc(size_t bytes, void** out, bool zero) {
    void* d = nullptr;
    CTRLV_HIP_TRY(hipMalloc(&d, bytes ? bytes : 256));
    p->owned.push_back(d);
    if (zero) CTRLV_HIP_TRY(hipMemsetAsync(d, 0, bytes ? bytes : 256, st));
    *out = d;
    return CTRLV_OK;
  }
  // one weight tensor [N][C][taps] -> rows [n_off, n_off+N) of `dst`
  int pack_w(const std::string& name, int N, int C, int taps, Linear& dst, int cp, int c_off, int n_off, int geglu) {
    const ctrlv_tensor_desc* t;
    TRY(find(name, &t, (long)N * C * taps));
    const void* src;
    TRY(dev_ptr(t, &src));
    const long total = (long)N * C * taps;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, src, t->dtype, N, C, taps, dst.w, dst.k, cp,
                       c_off, n_off, geglu);
    CTRLV_LAUNCH_CHECK();
    return CTRLV_OK;
  }
  int pack_v(const std::string& name, int N, float* dst, int n_off, int geglu, int accumulate) {
    const ctrlv_tensor_desc* t;
    TRY(find(name, &t, N));
    const void* src;
    TRY(dev_ptr(t, &src));
    hipLaunchKernelGGL(pack_vector_kernel, dim3((N + 255) / 256), dim3(256), 0, st, src, t->dtype, N, dst, n_off, geglu,
                       accumulate);
    CTRLV_LAUNCH_CHECK();
    return CTRLV_OK;
  }
  int new_linear(Linear& l, int n_rows, int k_total, bool bias) {
    l.n = pad_to(n_rows, 32);
    l.k = k_total;
    TRY(alloc((size_t)l.n * l.k * 2, (void**)&l.w, true));
    l.b = nullptr;
    if (bias) TRY(alloc((size_t)l.n * 4, (void**)&l.b, true));
    return CTRLV_OK;
  }
  // nn.Linear / 1x1 conv [N][K] -> [N32][K64]
  int linear(const std::string& mod, int N, int K, Linear& l, bool bias = true, int geglu = 0) {
    TRY(new_linear(l, N, pad_to(K, 64), bias));
    TRY(pack_w(mod + ".weight", N, K, 1, l, 0, 0, 0, geglu));
    if (bias) TRY(pack_v(mod + ".bias", N, l.b, 0, geglu, 0));
    return CTRLV_OK;
  }
  int conv3x3(const std::string& mod, int N, int C, Linear& l) {      // k = (ky*3+kx)*C + c
    TRY(new_linear(l, N, 9 * C, true));
    TRY(pack_w(mod + ".weight", N, C, 9, l, C, 0, 0, 0));
    return pack_v(mod + ".bias", N, l.b, 0, 0, 0);
  }
  int conv_t(const std::string& mod, int N, int C, Linear& l) {       // k = t*C + c
    TRY(new_linear(l, N, 3 * C, true));
    TRY(pack_w(mod + ".weight", N, C, 3, l, C, 0, 0, 0));
    return pack_v(mod + ".bias", N, l.b, 0, 0, 0);
  }
  int norm(const std::string& mod, int C, Norm& nm) {
    TRY(alloc((size_t)C * 4, (void**)&nm.g, false));
    TRY(alloc((size_t)C * 4, (void**)&nm.b, false));
    TRY(pack_v(mod + ".weight", C, nm.g, 0, 0, 0));
    return pack_v(mod + ".bias", C, nm.b, 0, 0, 0);
  }
  int mix_alpha(const std::string& name, double* alpha) {             // sigmoid(mix_factor): AlphaBlender, scalar
    const ctrlv_tensor_desc* t;
    TRY(find(name, &t, 1));
    unsigned char raw[4] = {0, 0, 0, 0};
    const size_t es = t->dtype == 0 ? 4 : 2;
    if (t->on_device) CTRLV_HIP_TRY(hipMemcpy(raw, t->data, es, hipMemcpyDeviceToHost));
    else memcpy(raw, t->data, es);
    float v;
    if (t->dtype == 0) memcpy(&v, raw, 4);
    else if (t->dtype == 1) { _Float16 h; memcpy(&h, raw, 2); v = (float)h; }
    else { uint32_t u = ((uint32_t)raw[1] << 24) | ((uint32_t)raw[0] << 16); memcpy(&v, &u, 4); }
    *alpha = 1.0 / (1.0 + exp(-(double)v));
    return CTRLV_OK;
  }
  int ff(const std::string& mod, int C, int C_out, FeedFwd& f) {     // FeedForward: GEGLU(C -> 8C) then Linear(4C -> C_out)
    TRY(linear(mod + ".net.0.proj", 8 * C, C, f.proj, true, 1));
    TRY(linear(mod + ".net.2", C_out, 4 * C, f.out));
    if (C == 320 && C_out == 320 && f.proj.n == 2560 && f.proj.k == 320 && f.out.n == 320 && f.out.k == 1280) {
      TRY(alloc((size_t)ctrlv_ff_fused_w1f_bytes(), (void**)&f.w1f, false));
      TRY(alloc((size_t)320 * 1280 * 2, (void**)&f.w2f, false));
      TRY(ctrlv_ff_fused_pack(f.proj.w, f.proj.b, f.out.w, f.w1f, f.w2f, st));
    }
    return CTRLV_OK;
  }
  int qkv(const std::string& mod, int C, Linear& l) {
    TRY(new_linear(l, 3 * C, pad_to(C, 64), false));
    TRY(pack_w(mod + ".to_q.weight", C, C, 1, l, 0, 0, 0, 0));
    TRY(pack_w(mod + ".to_k.weight", C, C, 1, l, 0, 0, C, 0));
    return pack_w(mod + ".to_v.weight", C, C, 1, l, 0, 0, 2 * C, 0);
  }
};

int load_res(Loader& L, ResBlock& r, int ted) {
  const std::string s = r.name + ".spatial_res_block", t = r.name + ".temporal_res_block";
  TRY(L.norm(s + ".norm1", r.cin, r.n1));
  TRY(L.conv3x3(s + ".conv1", r.cout, r.cin, r.c1));
  TRY(L.norm(s + ".norm2", r.cout, r.n2));
  TRY(L.conv3x3(s + ".conv2", r.cout, r.cout, r.c2));
  if (r.has_sc) TRY(L.linear(s + ".conv_shortcut", r.cout, r.cin, r.sc));
  TRY(L.norm(t + ".norm1", r.cout, r.tn1));
  TRY(L.conv_t(t + ".conv1", r.cout, r.cout, r.tc1));
  TRY(L.norm(t + ".norm2", r.cout, r.tn2));
  TRY(L.conv_t(t + ".conv2", r.cout, r.cout, r.tc2));
  TRY(L.mix_alpha(r.name + ".time_mixer.mix_factor", &r.alpha));
  // the two time_emb_proj rows go into the model-wide [sum Cout, ted] GEMM
  ctrlv_plan* p = L.p;
  const std::string tp[2] = {s + ".time_emb_proj", t + ".time_emb_proj"};
  for (int k = 0; k < 2; ++k) {
    TRY(L.pack_w(tp[k] + ".weight", r.cout, ted, 1, p->temb, 0, 0, r.temb_off[k], 0));
    TRY(L.pack_v(tp[k] + ".bias", r.cout, p->temb.b, r.temb_off[k], 0, 0));
  }
  return CTRLV_OK;
}

int load_tr(Loader& L, Transformer& t, int cross_dim) {
  const int C = t.C;
  const std::string sb = t.name + ".transformer_blocks.0", tb = t.name + ".temporal_transformer_blocks.0";
  TRY(L.norm(t.name + ".norm", C, t.gn));
  TRY(L.linear(t.name + ".proj_in", C, C, t.pin));
  TRY(L.linear(t.name + ".proj_out", C, C, t.pout));
  TRY(L.norm(sb + ".norm1", C, t.s_ln1));
  TRY(L.norm(sb + ".norm3", C, t.s_ln3));
  TRY(L.qkv(sb + ".attn1", C, t.s_qkv));
  TRY(L.linear(sb + ".attn1.to_out.0", C, C, t.s_o));
  TRY(L.ff(sb + ".ff", C, C, t.s_ff));
  TRY(L.norm(tb + ".norm_in", C, t.t_lnin));
  TRY(L.norm(tb + ".norm1", C, t.t_ln1));
  TRY(L.norm(tb + ".norm3", C, t.t_ln3));
  TRY(L.ff(tb + ".ff_in", C, C, t.t_ffin));
  TRY(L.qkv(tb + ".attn1", C, t.t_qkv));
  TRY(L.linear(tb + ".attn1.to_out.0", C, C, t.t_o));
  TRY(L.ff(tb + ".ff", C, C, t.t_ff));
  if (C == 320 && t.t_qkv.n == 960 && t.t_qkv.k == 320 && t.t_o.n == 320 && t.t_o.k == 320) {
    TRY(L.alloc(ctrlv_temporal_fused_weight_bytes(), (void**)&t.t_wf, false));
    TRY(ctrlv_temporal_fused_pack(t.t_qkv.w, t.t_qkv.k, t.t_o.w, t.t_o.k, t.t_wf, L.st));
  }
  TRY(L.linear(t.name + ".time_pos_embed.linear_1", 4 * C, C, t.tpe1));
  TRY(L.linear(t.name + ".time_pos_embed.linear_2", C, 4 * C, t.tpe2));
  TRY(L.mix_alpha(t.name + ".time_mixer.mix_factor", &t.alpha));
  // cross-attentions (1 key: attn2(x) = to_out(to_v(e)), SURVEY finding 8): to_v rows into the model-wide GEMM
  ctrlv_plan* p = L.p;
  const std::string a2[2] = {sb + ".attn2", tb + ".attn2"};
  for (int k = 0; k < 2; ++k) {
    TRY(L.pack_w(a2[k] + ".to_v.weight", C, cross_dim, 1, p->xv, 0, 0, t.xattn_off[k], 0));
    CrossOut xo;
    xo.off = t.xattn_off[k];
    xo.c = C;
    TRY(L.linear(a2[k] + ".to_out.0", C, C, xo.to_out));
    p->xouts.push_back(xo);
  }
  return CTRLV_OK;
}

// ------------------------------------------------------------------------------------------------ execution context
// A tensor of the RESIDUAL TRUNK (block inputs / outputs, the tensors every branch result is added back into).  In trunk
// mode 1 it carries a second plane: value = hi + lo (include/ctrlv_hip.h, ctrlv_gemm_desc.out_lo).  GEMM A operands and
// the ABI's tensors read `hi`; residual operands (R1 / R2) and norm inputs read both.
struct Trk {
  el_t* hi = nullptr;
  lo_t* lo = nullptr;        // one byte per element (e5m2: common.h lo_t)
};
struct Ctx {
  ctrlv_plan* p;
  hipStream_t st;
  bool dry;                  // measuring pass for ctrlv_plan_workspace_bytes: no launches, no dereferences
  char* base;
  size_t cap, off = 0, peak = 0;
  bool overflow = false;
  int B, F;
  float* temb = nullptr; int ldtemb = 0;
  float* xattn = nullptr; int ldx = 0;
  bool quirk = false;
  // GroupNorm chunk partials that cross a block boundary: written by the last GEMM of a res block (temporal conv2) for
  // the GroupNorm that opens the transformer behind it (one buffer per forward, used in stream order)
  float* gn_cross = nullptr; size_t gn_cross_floats = 0; bool gn_cross_valid = false;

  void* alloc(size_t bytes) {
    const size_t a = (bytes + 255) & ~(size_t)255;
    const size_t o = off;
    off += a;
    if (off > peak) peak = off;
    if (!dry && off > cap) overflow = true;
    return base + o;
  }
  el_t* rows(long m, int c) { return (el_t*)alloc((size_t)m * c * 2); }
  Trk trunk(long m, int c) {          // a trunk tensor: one plane, or hi + lo in trunk mode 1
    Trk t;
    t.hi = rows(m, c);
    if (p->trunk_mode == 1) t.lo = (lo_t*)alloc((size_t)m * c);
    return t;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

// One launch (or launch group) of the walk under the profiler: events on the launch stream before / after it.
struct ProfScope {
  ctrlv_plan* p = nullptr;
  hipStream_t st;
  size_t idx = 0;
  ProfScope(Ctx& c, int family, double flops, double bytes, int M = 0, int N = 0, int K = 0, int flags = 0) : st(c.st) {
    if (c.dry || !c.p->profiling) return;
    ctrlv_plan::ProfRec pr;
    memset(&pr.r, 0, sizeof(pr.r));
    pr.r.family = family; pr.r.M = M; pr.r.N = N; pr.r.K = K; pr.r.flags = flags; pr.r.flops = flops; pr.r.bytes = bytes;
    if (hipEventCreate(&pr.e0) != hipSuccess || hipEventCreate(&pr.e1) != hipSuccess) return;
    (void)hipEventRecord(pr.e0, st);
    p = c.p;
    idx = p->prof.size();
    p->prof.push_back(pr);
  }
  ~ProfScope() {
    if (p) (void)hipEventRecord(p->prof[idx].e1, st);
  }
};
// algorithmic work of a gather-GEMM launch (the accounting of ops.py: 2 MAC per output element and K step; operands once)
void gemm_work(const ctrlv_gemm_desc& d, int* fam, double* flops, double* bytes) {
  *fam = d.mode == 1 ? CTRLV_FAM_GEMM_CONV3X3 : (d.mode == 2 ? CTRLV_FAM_GEMM_CONV_TEMPORAL : CTRLV_FAM_GEMM_LINEAR);
  const double n_alg = d.geglu ? d.N : (d.N < d.n_store ? d.N : d.n_store);
  const double n_out = d.geglu ? (d.n_store < d.N / 2 ? d.n_store : d.N / 2) : d.n_store;
  const double K = (double)d.taps * d.Cin;
  *flops = 2.0 * d.M * n_alg * K;
  *bytes = (double)d.M * d.Cin * 2 + (double)d.M * n_out * ((d.out_f32 & 1) ? 4 : 2) + (double)d.N * K * 2 +
           (d.R1 ? (double)d.M * n_out * 2 : 0) + (d.R2 ? (double)d.M * n_out * 2 : 0) +
           ((d.R1_lo ? 1 : 0) + (d.R2_lo ? 1 : 0) + (d.out_lo ? 1 : 0)) * (double)d.M * n_out;       // (lo planes: 1 byte)
}
// residual operand / output of a GEMM from a trunk tensor
inline void set_r1(ctrlv_gemm_desc& d, const Trk& t, int ld) { d.R1 = t.hi; d.R1_lo = t.lo; d.ldr1 = ld; }
inline void set_r2(ctrlv_gemm_desc& d, const Trk& t, int ld) { d.R2 = t.hi; d.R2_lo = t.lo; d.ldr2 = ld; }
inline void set_out(ctrlv_gemm_desc& d, const Trk& t) { d.out = t.hi; d.out_lo = t.lo; }

ctrlv_gemm_desc gd(const void* A, int lda, const Linear& w, void* out, int ldo, int M, int N, int cin, int n_store) {
  ctrlv_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.W = w.w; d.out = out; d.bias = w.b;
  d.M = M; d.N = N; d.Cin = cin; d.taps = 1;
  d.lda = lda; d.ldo = ldo; d.n_store = n_store;
  d.s_acc = d.s1 = d.s2 = 1.0f;
  d.vdiv = 1; d.vmod = 1 << 30; d.vS = 1;
  return d;
}
int gemm(Ctx& c, const ctrlv_gemm_desc& d0) {
  // split contraction of the small-image long-K convs (gemm.hip splitk_plan): its scratch comes from the arena for the
  // duration of the launch (the plan depends on the layer's shape only, so the measuring walk sees the same sizes)
  ctrlv_gemm_desc d = d0;
  const size_t mk = c.mark();
  const size_t ws = ctrlv_gemm_splitk_ws_bytes(&d);
  if (ws) d.splitk_ws = c.alloc(ws);
  int rc = CTRLV_OK;
  if (!c.dry) {
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    int fam; double fl, by;
    gemm_work(d, &fam, &fl, &by);
    ProfScope ps(c, fam, fl, by, d.M, d.N, d.taps * d.Cin, (d.geglu ? 1 : 0) | ((d.R1 ? 1 : 0) + (d.R2 ? 1 : 0)) << 1 | d.vmode << 3);
    rc = ctrlv_gemm(&d, c.st);
  }
  c.release(mk);
  return rc;
}
int groupnorm(Ctx& c, const Trk& x, const Trk& x2, int c_split, int n_img, int S, int C, int ips, const Norm& nm,
              float eps, int silu, el_t* y) {
  const int chunks = ctrlv_groupnorm_chunks(n_img, S, C, ips);
  if (chunks < 0) return chunks;
  const size_t m = c.mark();
  float* part = (float*)c.alloc(((size_t)n_img * chunks + n_img / ips) * 64 * 4);
  int rc = CTRLV_OK;
  if (!c.dry) {
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    ProfScope ps(c, CTRLV_FAM_GROUPNORM, 0.0, 2.0 * 2 * n_img * (double)S * C, n_img * S, C);   // algorithmic: 1 read + 1 write
    rc = ctrlv_groupnorm_stats_split(x.hi, x.lo, x2.hi, x2.lo, c_split, n_img, S, C, ips, eps, part, c.st);
    if (rc == CTRLV_OK)
      rc = ctrlv_groupnorm_apply_split(x.hi, x.lo, x2.hi, x2.lo, c_split, n_img, S, C, ips, part, nm.g, nm.b, silu, y, c.st);
  }
  c.release(m);          // stream order keeps the scratch alive until the apply pass has read it
  return rc;
}
// A GEMM whose output goes straight into a GroupNorm (conv1 -> norm2, conv2 -> temporal norm1, temporal conv1 -> temporal
// norm2 of a res block): where the launch serves it, its epilogue writes the norm's chunk partials and the norm is
// finalize + apply -- one read and one write of the tensor instead of two reads.  The scratch is sized from the shape alone
// (the measuring walk has no operand pointers to ask the predicate with).
int gemm_groupnorm(Ctx& c, ctrlv_gemm_desc d, int n_img, int S, int C, int ips, const Norm& nm, float eps, int silu, el_t* y) {
  const Trk dout{(el_t*)d.out, (lo_t*)d.out_lo};
  if (S % 64 != 0) {
    TRY(gemm(c, d));
    return groupnorm(c, dout, Trk{}, 0, n_img, S, C, ips, nm, eps, silu, y);
  }
  const size_t m = c.mark();
  float* part = (float*)c.alloc(((size_t)n_img * (S / 64) + n_img / ips) * 64 * 4);
  int rc = CTRLV_OK;
  if (!c.dry && ctrlv_gemm_gn_partials_serves(&d)) {
    d.gn_partials = part;
    rc = gemm(c, d);
    if (rc == CTRLV_OK) {
      ProfScope ps(c, CTRLV_FAM_GROUPNORM, 0.0, 2.0 * 2 * n_img * (double)S * C, n_img * S, C, 1);   // flags 1: fused statistics
      rc = ctrlv_groupnorm_from_partials_split(d.out, d.out_lo, n_img, S, C, ips, eps, part, nm.g, nm.b, silu, y, c.st);
    }
    c.release(m);
    return rc;
  }
  c.release(m);
  TRY(gemm(c, d));
  return groupnorm(c, dout, Trk{}, 0, n_img, S, C, ips, nm, eps, silu, y);
}
int layernorm(Ctx& c, const Trk& x, int M, int C, const Norm& nm, el_t* y, const float* V = nullptr, int vdiv = 1,
              int vmod = 1 << 30, int ldv = 0) {
  if (c.dry) return CTRLV_OK;
  ProfScope ps(c, CTRLV_FAM_LAYERNORM, 0.0, 2.0 * 2 * (double)M * C, M, C);
  return ctrlv_layernorm_split(x.hi, x.lo, M, C, nm.g, nm.b, 1e-5f, V, vdiv, vmod, ldv, y, c.st);
}

// ---- SpatioTemporalResBlock (blocks.py::SpatioTemporalResBlock.run)
int run_res(Ctx& c, const ResBlock& r, const Trk& x, const Trk& x2, int c1, int H, int W, Trk* out_,
            bool feeds_norm = false) {
  const int N = c.B * c.F, S = H * W, F = c.F;
  const long M = (long)N * S;
  const int cin = r.cin, cout = r.cout;
  c.gn_cross_valid = false;
  if (feeds_norm && S % 64 == 0) {      // (sized from the shape alone, outside this block's scratch region)
    const size_t need = ((size_t)N * (S / 64) + N) * 64;
    if (c.gn_cross_floats < need) { c.gn_cross = (float*)c.alloc(need * 4); c.gn_cross_floats = need; }
  }
  const Trk out = c.trunk(M, cout);
  const size_t mk = c.mark();
  const int lda_x = x2.hi ? c1 : cin;
  el_t* xn = c.rows(M, cin);
  TRY(groupnorm(c, x, x2, x2.hi ? c1 : 0, N, S, cin, 1, r.n1, r.eps, 1, xn));
  el_t* h = c.rows(M, cout);
  el_t* hn = nullptr;
  {
    ctrlv_gemm_desc d = gd(xn, cin, r.c1, h, cout, (int)M, cout, cin, cout);
    d.taps = 9; d.mode = 1; d.H = H; d.Wd = W; d.Ho = H; d.Wo = W; d.stride = 1; d.up = 0;
    d.V = c.temb + r.temb_off[0]; d.ldv = c.ldtemb; d.vmode = 1; d.vdiv = F * S;
    hn = c.rows(M, cout);
    TRY(gemm_groupnorm(c, d, N, S, cout, 1, r.n2, r.eps, 1, hn));
  }
  Trk res = x;
  int ldres = cin;
  if (r.has_sc) {
    const Trk rs = c.trunk(M, cout);
    ctrlv_gemm_desc d = gd(x.hi, lda_x, r.sc, rs.hi, cout, (int)M, cout, cin, cout);
    set_out(d, rs);
    if (x2.hi) { d.A2 = x2.hi; d.lda2 = cin - c1; d.c_split = c1; }
    TRY(gemm(c, d));
    res = rs;
    ldres = cout;
  }
  const Trk xs = c.trunk(M, cout);
  {
    ctrlv_gemm_desc d = gd(hn, cout, r.c2, xs.hi, cout, (int)M, cout, cout, cout);
    d.taps = 9; d.mode = 1; d.H = H; d.Wd = W; d.Ho = H; d.Wo = W; d.stride = 1;
    set_out(d, xs);
    set_r1(d, res, ldres);
    TRY(gemm_groupnorm(c, d, N, S, cout, F, r.tn1, r.eps, 1, hn));
  }
  {
    ctrlv_gemm_desc d = gd(hn, cout, r.tc1, h, cout, (int)M, cout, cout, cout);
    d.taps = 3; d.mode = 2; d.F = F; d.S = S;
    d.V = c.temb + r.temb_off[1]; d.ldv = c.ldtemb; d.vmode = 1; d.vdiv = F * S;
    TRY(gemm_groupnorm(c, d, N, S, cout, F, r.tn2, r.eps, 1, hn));
  }
  {   // AlphaBlender: a*xs + (1-a)*(xs + conv2) = xs + (1-a)*conv2
    ctrlv_gemm_desc d = gd(hn, cout, r.tc2, out.hi, cout, (int)M, cout, cout, cout);
    d.taps = 3; d.mode = 2; d.F = F; d.S = S;
    d.s_acc = (float)(1.0 - r.alpha);
    set_out(d, out);
    set_r1(d, xs, cout);
    if (feeds_norm && S % 64 == 0 && !c.dry && ctrlv_gemm_gn_partials_serves(&d)) {
      d.gn_partials = c.gn_cross;       // the transformer behind this block opens with a GroupNorm of `out`
      c.gn_cross_valid = true;
    }
    TRY(gemm(c, d));
  }
  c.release(mk);
  *out_ = out;
  return CTRLV_OK;
}

// frame positional embedding table [F][C] fp32 = time_pos_embed(time_proj(arange(F)))  (depends on the weights and F only)
int frame_embedding(Ctx& c, const Transformer& t, int F, float* e) {
  const int C = t.C, kp = t.tpe1.k;
  const size_t mk = c.mark();
  float* ar = (float*)c.alloc((size_t)F * 4);
  el_t* te = c.rows(F, kp);
  el_t* hh = c.rows(F, 4 * C);
  el_t* tmp = kp != C ? c.rows(F, C) : nullptr;
  if (!c.dry) {
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small"); return CTRLV_E_BAD_ARG; }
    hipLaunchKernelGGL(arange_kernel, dim3((F + 63) / 64), dim3(64), 0, c.st, ar, F);
    CTRLV_LAUNCH_CHECK();
    if (kp != C) {       // K zero padding (tiny configs only): sinusoid into a compact buffer, then strided copy
      CTRLV_HIP_TRY(hipMemsetAsync(te, 0, (size_t)F * kp * 2, c.st));
      TRY(ctrlv_timestep_embedding(ar, F, C, tmp, c.st));
      CTRLV_HIP_TRY(hipMemcpy2DAsync(te, (size_t)kp * 2, tmp, (size_t)C * 2, (size_t)C * 2, F, hipMemcpyDeviceToDevice, c.st));
    } else {
      TRY(ctrlv_timestep_embedding(ar, F, C, te, c.st));
    }
  }
  {
    ctrlv_gemm_desc d = gd(te, kp, t.tpe1, hh, 4 * C, F, t.tpe1.n, kp, 4 * C);
    d.act = 1;
    TRY(gemm(c, d));
  }
  {
    ctrlv_gemm_desc d = gd(hh, 4 * C, t.tpe2, e, C, F, t.tpe2.n, 4 * C, C);
    d.out_f32 = 1;
    TRY(gemm(c, d));
  }
  c.release(mk);
  return CTRLV_OK;
}

// `u` = the 4C-wide intermediate of the two-launch path, allocated HERE on first need (a transformer whose three
// feed-forwards all run fused never reserves it: 1.2 GB per lane at the 72 x 128 level); proj.out / outd.A are set from it.
int ff_pair(Ctx& c, const FeedFwd& f, ctrlv_gemm_desc proj, ctrlv_gemm_desc outd, int C, el_t** u) {
  // C = 320: one fused launch, the 4C-wide intermediate stays on chip (ff_fused.hip): 234.5 -> 231.3 ms per step (three
  // alternations on one device).  CTRLV_FF_FUSED=0: the two launches.
  if (ctrlv_debug().ff_fused && f.w1f && ctrlv_ff_fused_serves(&outd, proj.lda)) {
    if (c.dry) return CTRLV_OK;
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    ProfScope ps(c, CTRLV_FAM_GEMM_LINEAR, 2.0 * outd.M * 320.0 * (2560 + 1280),
                 (double)outd.M * 320 * (2 * (2 + (outd.R1 ? 1 : 0) + (outd.R2 ? 1 : 0)) + (outd.R1_lo ? 1 : 0) + (outd.R2_lo ? 1 : 0) +
                                         (outd.out_lo ? 1 : 0)), outd.M, 320, 320, 0x100);
    return ctrlv_ff_fused(proj.A, proj.lda, f.w1f, f.w2f, &outd, c.st);
  }
  if (*u == nullptr) *u = c.rows(proj.M, 4 * C);
  proj.out = *u;
  outd.A = *u;
  TRY(gemm(c, proj));
  TRY(gemm(c, outd));
  return CTRLV_OK;
}

// LayerNorm -> feed-forward (blocks.py::_ln_ff).  OPT-IN (CTRLV_FF_LN=1): the norm folded into the fused kernel's tile
// prologue (ctrlv_ff_fused_ln: one launch and one write + read of the activation less).  Measured equal in the model
// (224.2 vs 224.2 and 230.3 vs 230.4 ms per step, three alternations each: the LayerNorm family drops 7.9 -> 5.3 ms, the
// fused kernel's per-tile prologue takes it back), so the default keeps ctrlv_layernorm in front of the fused kernel.
int ln_ff(Ctx& c, const FeedFwd& f, const Norm& nm, const Trk& xraw, const float* lnV, int lnvdiv, int lnvmod, int lnldv,
          el_t* tt, const ctrlv_gemm_desc& proj, const ctrlv_gemm_desc& outd, int C, el_t** u) {
  if (ctrlv_debug().ff_fused && ctrlv_debug().ff_ln && f.w1f && !xraw.lo && ctrlv_ff_fused_serves(&outd, C)) {
    if (c.dry) return CTRLV_OK;
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    ProfScope ps(c, CTRLV_FAM_GEMM_LINEAR, 2.0 * outd.M * 320.0 * (2560 + 1280),
                 (double)outd.M * 320 * 2 * (2 + (outd.R1 ? 1 : 0) + (outd.R2 ? 1 : 0)), outd.M, 320, 320, 0x300);
    return ctrlv_ff_fused_ln(xraw.hi, C, nm.g, nm.b, 1e-5f, lnV, lnvdiv, lnvmod, lnldv, f.w1f, f.w2f, &outd, c.st);
  }
  TRY(layernorm(c, xraw, proj.M, C, nm, tt, lnV, lnvdiv, lnvmod, lnldv));
  return ff_pair(c, f, proj, outd, C, u);
}

// ---- TransformerSpatioTemporalModel (blocks.py::TransformerSpatioTemporalModel.run)
int run_tr(Ctx& c, const Transformer& t, const Trk& x, int H, int W, Trk* out_) {
  const int B = c.B, F = c.F, C = t.C, N = B * F, S = H * W;
  const long M = (long)N * S;
  const Trk out = c.trunk(M, C);
  const size_t mk = c.mark();
  const float* emb = t.frame_emb;
  if (F != c.p->cfg.num_frames || emb == nullptr) {
    float* e = (float*)c.alloc((size_t)F * C * 4);
    TRY(frame_embedding(c, t, F, e));
    emb = e;
  }
  el_t* tt = c.rows(M, C);
  if (c.gn_cross_valid && !c.dry) {     // statistics from the res block's last GEMM (run_res, feeds_norm)
    c.gn_cross_valid = false;
    ProfScope ps(c, CTRLV_FAM_GROUPNORM, 0.0, 2.0 * 2 * N * (double)S * C, N * S, C, 1);
    TRY(ctrlv_groupnorm_from_partials_split(x.hi, x.lo, N, S, C, 1, 1e-6f, c.gn_cross, t.gn.g, t.gn.b, 0, tt, c.st));
  } else {
    TRY(groupnorm(c, x, Trk{}, 0, N, S, C, 1, t.gn, 1e-6f, 0, tt));
  }
  const Trk h0 = c.trunk(M, C);
  {
    ctrlv_gemm_desc d = gd(tt, C, t.pin, h0.hi, C, (int)M, C, C, C);
    set_out(d, h0);
    TRY(gemm(c, d));
  }
  // ---- spatial BasicTransformerBlock
  TRY(layernorm(c, h0, (int)M, C, t.s_ln1, tt));
  el_t* qkv = c.rows(M, 3 * C);
  {
    ctrlv_gemm_desc d = gd(tt, C, t.s_qkv, qkv, 3 * C, (int)M, 3 * C, C, 3 * C);
    d.n_scale2 = C; d.s_acc2 = 0.125f * 1.44269504088896340736f;      // q block pre-scaled: (1/sqrt(64)) log2(e)
    TRY(gemm(c, d));
  }
  el_t* a = c.rows(M, C);
  if (!c.dry) {
    ProfScope ps(c, CTRLV_FAM_ATTENTION_SPATIAL, 4.0 * N * (C / 64) * (double)S * S * 64, 2.0 * 4 * N * (double)S * C, N, S, C);
    TRY(ctrlv_attention_spatial_prescaled(qkv, a, N, S, C, c.st));
  }
  const Trk h1 = c.trunk(M, C);
  {   // attn2 with one key == to_out(to_v(ehs[b])) for every query: a per-clip row vector
    ctrlv_gemm_desc d = gd(a, C, t.s_o, h1.hi, C, (int)M, C, C, C);
    set_out(d, h1);
    set_r1(d, h0, C);
    d.V = c.xattn + t.xattn_off[0]; d.ldv = c.ldx; d.vmode = 1; d.vdiv = F * S;
    TRY(gemm(c, d));
  }
  el_t* u = nullptr;  // 4C-wide GEGLU output of the two-launch path: allocated by ff_pair on first need
  const Trk h2 = h0;  // h0 is dead from here on
  {
    ctrlv_gemm_desc dp = gd(tt, C, t.s_ff.proj, u, 4 * C, (int)M, 8 * C, C, 4 * C);
    dp.geglu = 1;
    ctrlv_gemm_desc d = gd(u, 4 * C, t.s_ff.out, h2.hi, C, (int)M, C, 4 * C, C);
    set_out(d, h2);
    set_r1(d, h1, C);
    d.S = S;                              // (S: rows per image, the split plan's shape key for a mode-0 launch)
    TRY(ln_ff(c, t.s_ff, t.s_ln3, h1, nullptr, 1, 1 << 30, 0, tt, dp, d, C, &u));
  }
  // ---- temporal block on tokens (b, s) x frames; rows stay ordered (b, f, s)
  const Trk g0 = h1;  // h1 is dead
  {
    ctrlv_gemm_desc dp = gd(tt, C, t.t_ffin.proj, u, 4 * C, (int)M, 8 * C, C, 4 * C);
    dp.geglu = 1;
    ctrlv_gemm_desc d = gd(u, 4 * C, t.t_ffin.out, g0.hi, C, (int)M, C, 4 * C, C);
    set_out(d, g0);
    set_r1(d, h2, C);
    d.S = S;
    d.V = emb; d.ldv = C; d.vmode = 1; d.vdiv = S; d.vmod = F;
    TRY(ln_ff(c, t.t_ffin, t.t_lnin, h2, emb, S, F, C, tt, dp, d, C, &u));
  }
  const Trk g1 = c.trunk(M, C);
  {
    // norm1 + attn1 over the frames + residual + the one-key cross-attention vector (attn2): at C = 320 ONE launch -- the
    // normalised rows, q|k|v and the attention output never reach HBM (temporal_fused.hip; CTRLV_TEMPORAL_FUSED=0: the four
    // launches).  With split trunk planes the in-kernel norm reads the hi plane (the element-rounded branch input: one more
    // rounding on this branch's input, none on the trunk -- the residual operand stays hi + lo; the complete step's rel-L2
    // against the oracle is unchanged within 1e-5, bench.py's parity leg).
    ctrlv_temporal_fused_desc fd;
    memset(&fd, 0, sizeof(fd));
    fd.x = tt; fd.ldx = C; fd.wf = t.t_wf; fd.bias = t.t_o.b;
    const bool fuse = ctrlv_debug().temporal_fused && t.t_wf;
    bool ln_in = fuse;
    if (ln_in) { fd.x = g0.hi; fd.ln_gamma = t.t_ln1.g; fd.ln_beta = t.t_ln1.b; fd.ln_eps = 1e-5f; }
    fd.R1 = g0.hi; fd.R1_lo = g0.lo; fd.ldr1 = C;
    fd.V = c.xattn + t.xattn_off[1]; fd.ldv = c.ldx; fd.vdiv = F * S; fd.vS = 1; fd.vmod = 1 << 30;
    if (c.quirk && B > 1) { fd.vmode = 2; fd.vS = S; fd.vmod = B; }   // diffusers 0.27.2: context rows (s, b), tokens (b, s)
    else fd.vmode = 1;
    fd.out = g1.hi; fd.out_lo = g1.lo; fd.ldo = C;
    fd.B = B; fd.F = F; fd.S = S; fd.C = C;
    const bool fused = fuse && ctrlv_temporal_fused_serves(&fd);
    if (!fused || !ln_in) {
      if (ln_in) { ln_in = false; fd.x = tt; fd.ln_gamma = fd.ln_beta = nullptr; }
      TRY(layernorm(c, g0, (int)M, C, t.t_ln1, tt));
    }
    if (fused) {
      if (!c.dry) {
        if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
        ProfScope ps(c, CTRLV_FAM_GEMM_TEMPORAL_BLOCK, 2.0 * M * C * 4.0 * C + 4.0 * B * S * (C / 64) * (double)F * F * 64,
                     (double)M * C * (2 * (ln_in ? 2 : 3) + (g0.lo ? 1 : 0) + (g1.lo ? 1 : 0)), (int)M, 4 * C, C, ln_in ? 1 : 0);
        TRY(ctrlv_temporal_fused(&fd, c.st));
      }
    } else {
      TRY(gemm(c, gd(tt, C, t.t_qkv, qkv, 3 * C, (int)M, 3 * C, C, 3 * C)));
      if (!c.dry) {
        ProfScope ps(c, CTRLV_FAM_ATTENTION_TEMPORAL, 4.0 * B * S * (C / 64) * (double)F * F * 64, 2.0 * 4 * B * F * (double)S * C, B * S,
                     F, C);
        TRY(ctrlv_attention_temporal(qkv, a, B, F, S, C, c.st));
      }
      ctrlv_gemm_desc d = gd(a, C, t.t_o, g1.hi, C, (int)M, C, C, C);
      set_out(d, g1);
      set_r1(d, g0, C);
      d.V = fd.V; d.ldv = fd.ldv; d.vdiv = fd.vdiv; d.vmode = fd.vmode;
      if (fd.vmode == 2) { d.vS = S; d.vmod = B; }
      TRY(gemm(c, d));
    }
  }
  const Trk h3 = g0;
  {   // AlphaBlender folded: h3 = a*h2 + (1-a)*(g1 + ff)
    ctrlv_gemm_desc dp = gd(tt, C, t.t_ff.proj, u, 4 * C, (int)M, 8 * C, C, 4 * C);
    dp.geglu = 1;
    ctrlv_gemm_desc d = gd(u, 4 * C, t.t_ff.out, h3.hi, C, (int)M, C, 4 * C, C);
    set_out(d, h3);
    d.s_acc = (float)(1.0 - t.alpha); d.s1 = (float)(1.0 - t.alpha); d.s2 = (float)t.alpha; d.S = S;
    set_r1(d, g1, C);
    set_r2(d, h2, C);
    TRY(ln_ff(c, t.t_ff, t.t_ln3, g1, nullptr, 1, 1 << 30, 0, tt, dp, d, C, &u));
  }
  {
    ctrlv_gemm_desc d = gd(h3.hi, C, t.pout, out.hi, C, (int)M, C, C, C);
    set_out(d, out);
    set_r1(d, x, C);
    TRY(gemm(c, d));
  }
  c.release(mk);
  *out_ = out;
  return CTRLV_OK;
}

int run_resample(Ctx& c, const Resample& r, const Trk& x, int H, int W, bool up, Trk* out_, int* Ho_, int* Wo_) {
  const int N = c.B * c.F;
  const int Ho = up ? 2 * H : (H + 2 - 3) / 2 + 1, Wo = up ? 2 * W : (W + 2 - 3) / 2 + 1;
  const long M = (long)N * Ho * Wo;
  const Trk out = c.trunk(M, r.C);
  ctrlv_gemm_desc d = gd(x.hi, r.C, r.conv, out.hi, r.C, (int)M, r.C, r.C, r.C);
  set_out(d, out);
  d.taps = 9; d.mode = 1; d.H = H; d.Wd = W; d.Ho = Ho; d.Wo = Wo; d.stride = up ? 1 : 2; d.up = up ? 1 : 0;
  TRY(gemm(c, d));
  *out_ = out; *Ho_ = Ho; *Wo_ = Wo;
  return CTRLV_OK;
}

struct Tap { Trk x; int H, W, C; };

// ---- embeddings + per-clip row-vector tables (encoder.py::_context)
int run_context(Ctx& c, int dtype, const float* timestep, int n_t, const void* ehs, const float* ids32, int n_ids) {
  ctrlv_plan* p = c.p;
  const ctrlv_model_config& cfg = p->cfg;
  const int B = c.B, boc0 = cfg.block_out_channels[0], ted = 4 * boc0;
  CTRLV_CHECK_ARG(n_t == 1 || n_t == B, "plan forward: timestep must have 1 or B=%d entries (got %d)", B, n_t);
  const int add_dim = cfg.addition_time_embed_dim;
  CTRLV_CHECK_SHAPE(add_dim * n_ids == cfg.projection_class_embeddings_input_dim,
                    "Model expects an added time embedding vector of length %d, but a vector of %d was created.",
                    cfg.projection_class_embeddings_input_dim, add_dim * n_ids);
  float* t32 = (float*)c.alloc((size_t)B * 4);
  const int kt = p->te1.k, ka = p->ae1.k, kx = p->xv.k;
  el_t* te = c.rows(B, kt);
  el_t* ae = c.rows(B, ka);
  el_t* te_tmp = kt != boc0 ? c.rows(B, boc0) : nullptr;
  el_t* ae_tmp = ka != n_ids * add_dim ? c.rows((long)B * n_ids, add_dim) : nullptr;
  el_t* h = c.rows(B, ted);
  el_t* emb_t = c.rows(B, ted);
  el_t* emb_s = c.rows(B, ted);
  c.ldtemb = p->temb.n;
  c.temb = (float*)c.alloc((size_t)B * c.ldtemb * 4);
  if (!c.dry) {
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    hipLaunchKernelGGL(expand_f32_kernel, dim3((B + 63) / 64), dim3(64), 0, c.st, timestep, n_t, t32, B);
    CTRLV_LAUNCH_CHECK();
    if (kt != boc0) {
      CTRLV_HIP_TRY(hipMemsetAsync(te, 0, (size_t)B * kt * 2, c.st));
      TRY(ctrlv_timestep_embedding(t32, B, boc0, te_tmp, c.st));
      CTRLV_HIP_TRY(hipMemcpy2DAsync(te, (size_t)kt * 2, te_tmp, (size_t)boc0 * 2, (size_t)boc0 * 2, B, hipMemcpyDeviceToDevice, c.st));
    } else {
      TRY(ctrlv_timestep_embedding(t32, B, boc0, te, c.st));
    }
    // added ids: sinusoid of every id -> [B*n_ids, add_dim] == [B, n_ids*add_dim]
    if (ka != n_ids * add_dim) {
      CTRLV_HIP_TRY(hipMemsetAsync(ae, 0, (size_t)B * ka * 2, c.st));
      TRY(ctrlv_timestep_embedding(ids32, B * n_ids, add_dim, ae_tmp, c.st));
      CTRLV_HIP_TRY(hipMemcpy2DAsync(ae, (size_t)ka * 2, ae_tmp, (size_t)n_ids * add_dim * 2, (size_t)n_ids * add_dim * 2, B,
                                     hipMemcpyDeviceToDevice, c.st));
    } else {
      TRY(ctrlv_timestep_embedding(ids32, B * n_ids, add_dim, ae, c.st));
    }
  }
  // (tile 11: per-clip rows -- one kernel for these launches whatever B is, so a clip's conditioning vectors have the same
  //  bits alone and in any batch: csrc/gemm.hip gemv_small_kernel)
  constexpr int kClipRows = 11;
  { ctrlv_gemm_desc d = gd(te, kt, p->te1, h, ted, B, ted, kt, ted); d.act = 1; d.tile = kClipRows; TRY(gemm(c, d)); }
  { ctrlv_gemm_desc d = gd(h, ted, p->te2, emb_t, ted, B, ted, ted, ted); d.tile = kClipRows; TRY(gemm(c, d)); }
  { ctrlv_gemm_desc d = gd(ae, ka, p->ae1, h, ted, B, ted, ka, ted); d.act = 1; d.tile = kClipRows; TRY(gemm(c, d)); }
  {   // silu(emb + aug_emb): the SiLU in front of every time_emb_proj
    ctrlv_gemm_desc d = gd(h, ted, p->ae2, emb_s, ted, B, ted, ted, ted);
    d.R1 = emb_t; d.ldr1 = ted; d.act = 1; d.tile = kClipRows;
    TRY(gemm(c, d));
  }
  { ctrlv_gemm_desc d = gd(emb_s, ted, p->temb, c.temb, c.ldtemb, B, p->temb.n, ted, c.ldtemb); d.out_f32 = 1; d.tile = kClipRows; TRY(gemm(c, d)); }
  if (p->xattn_n) {
    const int dc = cfg.cross_attention_dim, nx = p->xv.n;
    el_t* e = c.rows(B, kx);
    el_t* v_all = c.rows(B, nx);
    c.ldx = nx;
    c.xattn = (float*)c.alloc((size_t)B * nx * 4);
    if (!c.dry) {
      if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
      if (kx != dc) CTRLV_HIP_TRY(hipMemsetAsync(e, 0, (size_t)B * kx * 2, c.st));
      TRY(ctrlv_nchw_to_rows(ehs, dtype, B, dc, 1, e, kx, 0, c.st));      // (B, 1, dc) any dtype -> bf16 rows [B, kx]
    }
    { ctrlv_gemm_desc d = gd(e, kx, p->xv, v_all, nx, B, nx, kx, nx); d.tile = kClipRows; TRY(gemm(c, d)); }
    for (const CrossOut& xo : p->xouts) {
      ctrlv_gemm_desc d = gd(v_all + xo.off, nx, xo.to_out, c.xattn + xo.off, nx, B, xo.c, xo.c, xo.c);
      d.out_f32 = 1; d.tile = kClipRows;
      TRY(gemm(c, d));
    }
  }
  c.quirk = cfg.time_context_order == 0;
  return CTRLV_OK;
}

// conv_in (+ control_conv_in) as ONE im2col GEMM over the [conv_in channels | control channels | pad] slots
int run_input(Ctx& c, int dtype, const void* sample, const void* control, int h, int w, Trk* out_) {
  ctrlv_plan* p = c.p;
  const int N = c.B * c.F, cin = p->cfg.in_channels, c0 = p->cfg.block_out_channels[0];
  const long M = (long)N * h * w;
  el_t* x16 = c.rows(M, p->cin_cp);
  el_t* col = c.rows(M, p->cin_kp);
  const Trk x = c.trunk(M, c0);
  if (!c.dry) {
    if (c.overflow) { ctrlv_set_error("plan forward: workspace too small (need >= %zu bytes)", c.peak); return CTRLV_E_BAD_ARG; }
    CTRLV_HIP_TRY(hipMemsetAsync(x16, 0, (size_t)M * p->cin_cp * 2, c.st));
    TRY(ctrlv_nchw_to_rows(sample, dtype, N, cin, h * w, x16, p->cin_cp, 0, c.st));
    if (control) TRY(ctrlv_nchw_to_rows(control, dtype, N, cin / 2, h * w, x16, p->cin_cp, cin, c.st));
    TRY(ctrlv_im2col3x3(x16, N, h, w, p->cin_cp, col, p->cin_kp, c.st));
  }
  {
    ctrlv_gemm_desc d = gd(col, p->cin_kp, p->cin, x.hi, c0, (int)M, p->cin.n, p->cin_kp, c0);
    set_out(d, x);
    TRY(gemm(c, d));
  }
  *out_ = x;
  return CTRLV_OK;
}

int run_down_mid(Ctx& c, Trk x, int h, int w, std::vector<Tap>& taps, Trk* mid_, int* H_, int* W_) {
  ctrlv_plan* p = c.p;
  int H = h, W = w;
  taps.push_back({x, H, W, p->cfg.block_out_channels[0]});
  for (auto& b : p->down) {
    for (size_t j = 0; j < b.res.size(); ++j) {
      Trk y;
      TRY(run_res(c, b.res[j], x, Trk{}, 0, H, W, &y, !b.attn.empty()));
      x = y;
      if (!b.attn.empty()) { TRY(run_tr(c, b.attn[j], x, H, W, &y)); x = y; }
      taps.push_back({x, H, W, b.res[j].cout});
    }
    if (b.down.present) {
      Trk y; int Ho, Wo;
      TRY(run_resample(c, b.down, x, H, W, false, &y, &Ho, &Wo));
      x = y; H = Ho; W = Wo;
      taps.push_back({x, H, W, b.down.C});
    }
  }
  Trk y;
  TRY(run_res(c, p->mid_r0, x, Trk{}, 0, H, W, &y, true)); x = y;
  TRY(run_tr(c, p->mid_attn, x, H, W, &y)); x = y;
  TRY(run_res(c, p->mid_r1, x, Trk{}, 0, H, W, &y)); x = y;
  *mid_ = x; *H_ = H; *W_ = W;
  return CTRLV_OK;
}

int check_common(ctrlv_plan* p, int B, int F, int H, int W) {
  CTRLV_CHECK_ARG(p != nullptr, "plan: null plan");
  CTRLV_CHECK_ARG(p->loaded, "plan: weights not loaded (ctrlv_plan_load_weights)");
  CTRLV_CHECK_SHAPE(B > 0 && F > 0 && H > 0 && W > 0, "plan forward: B, F, H, W must be positive");
  const int m = 1 << (p->cfg.n_blocks - 1);
  CTRLV_CHECK_SHAPE(H % m == 0 && W % m == 0, "latent height and width have to be divisible by %d but are %d and %d.", m, H, W);
  return CTRLV_OK;
}

int unet_forward(ctrlv_plan* p, Ctx& c, const void* sample, int dtype, const float* timestep, int n_t, const void* ehs,
                 const float* ids, int n_ids, const void* const* down_res, const void* mid_res, void* res_event, void* out,
                 int h, int w) {
  const int N = c.B * c.F;
  TRY(run_context(c, dtype, timestep, n_t, ehs, ids, n_ids));
  Trk x;
  TRY(run_input(c, dtype, sample, nullptr, h, w, &x));
  std::vector<Tap> taps;
  int H, W;
  TRY(run_down_mid(c, x, h, w, taps, &x, &H, &W));
  // the ControlNet residual add on a trunk tensor: in place; in trunk mode 1 the sum is split again
  auto add_res = [&](const Trk& t, const void* r, size_t n) -> int {
    if (t.lo) return ctrlv_axpby_split(t.hi, t.lo, r, 1.0f, 1.0f, t.hi, t.lo, n, c.st);
    return ctrlv_axpby(t.hi, r, 1.0f, 1.0f, t.hi, n, c.st);
  };
  if (down_res && mid_res) {        // unet_spatio_temporal_condition.py:61,119-127,136-137
    if (!c.dry) {
      if (res_event) CTRLV_HIP_TRY(hipStreamWaitEvent(c.st, (hipEvent_t)res_event, 0));
      for (size_t i = 0; i < taps.size(); ++i) {
        CTRLV_CHECK_ARG(down_res[i] != nullptr, "unet_forward: down_res[%zu] is null", i);
        const size_t n = (size_t)N * taps[i].H * taps[i].W * taps[i].C;
        ProfScope ps(c, CTRLV_FAM_RESIDUAL_ADD, 0.0, 2.0 * 3 * (double)n, (int)(n / taps[i].C), taps[i].C);
        TRY(add_res(taps[i].x, down_res[i], n));
      }
      const size_t nm_ = (size_t)N * H * W * p->cfg.block_out_channels[p->cfg.n_blocks - 1];
      ProfScope ps(c, CTRLV_FAM_RESIDUAL_ADD, 0.0, 2.0 * 3 * (double)nm_, N * H * W, p->cfg.block_out_channels[p->cfg.n_blocks - 1]);
      TRY(add_res(x, mid_res, nm_));
    }
  }
  for (auto& b : p->up) {           // :140-158 -- torch.cat([hidden, skip], dim=1) is read in place (x | x2)
    for (size_t j = 0; j < b.res.size(); ++j) {
      const Tap skip = taps.back();
      taps.pop_back();
      Trk y;
      TRY(run_res(c, b.res[j], x, skip.x, b.res[j].cin - skip.C, H, W, &y, !b.attn.empty()));
      x = y;
      if (!b.attn.empty()) { TRY(run_tr(c, b.attn[j], x, H, W, &y)); x = y; }
    }
    if (b.up.present) {
      Trk y; int Ho, Wo;
      TRY(run_resample(c, b.up, x, H, W, true, &y, &Ho, &Wo));
      x = y; H = Ho; W = Wo;
    }
  }
  const int c0 = p->cfg.block_out_channels[0], co = p->cfg.out_channels, co_p = pad_to(co, 4);
  const long M = (long)N * H * W;
  el_t* xn = c.rows(M, c0);
  TRY(groupnorm(c, x, Trk{}, 0, N, H * W, c0, 1, p->gno, 1e-5f, 1, xn));         // :161-163
  el_t* y = c.rows(M, co_p);
  {
    ctrlv_gemm_desc d = gd(xn, c0, p->cout, y, co_p, (int)M, p->cout.n, c0, co_p);
    d.taps = 9; d.mode = 1; d.H = H; d.Wd = W; d.Ho = H; d.Wo = W; d.stride = 1;
    TRY(gemm(c, d));
  }
  if (!c.dry) TRY(ctrlv_rows_to_nchw(y, co_p, N, co, H * W, out, dtype, c.st));  // :166
  return CTRLV_OK;
}

int controlnet_forward(ctrlv_plan* p, Ctx& c, const void* sample, const void* control, int dtype, const float* timestep,
                       int n_t, const void* ehs, const float* ids, int n_ids, float scale, void* const* out_down,
                       void* out_mid, int h, int w) {
  const int N = c.B * c.F;
  TRY(run_context(c, dtype, timestep, n_t, ehs, ids, n_ids));
  Trk x;
  TRY(run_input(c, dtype, sample, control, h, w, &x));
  std::vector<Tap> taps;
  int H, W;
  TRY(run_down_mid(c, x, h, w, taps, &x, &H, &W));
  CTRLV_CHECK_ARG(taps.size() == p->zc.size(), "controlnet_forward: %zu taps but %zu zero-convs", taps.size(), p->zc.size());
  for (size_t i = 0; i < taps.size(); ++i) {       // controlnet.py:331-344: zero-conv * conditioning_scale, one GEMM
    const long M = (long)N * taps[i].H * taps[i].W;
    const int C = taps[i].C;
    if (!c.dry) CTRLV_CHECK_ARG(out_down && out_down[i], "controlnet_forward: out_down[%zu] is null", i);
    ctrlv_gemm_desc d = gd(taps[i].x.hi, C, p->zc[i], c.dry ? nullptr : out_down[i], C, (int)M, p->zc[i].n, C, C);
    d.s_acc = scale;
    TRY(gemm(c, d));
  }
  {
    const int C = p->cfg.block_out_channels[p->cfg.n_blocks - 1];
    ctrlv_gemm_desc d = gd(x.hi, C, p->zc_mid, out_mid, C, (int)((long)N * H * W), p->zc_mid.n, C, C);
    d.s_acc = scale;
    TRY(gemm(c, d));
  }
  return CTRLV_OK;
}

void free_owned(ctrlv_plan* p) {
  for (void* d : p->owned) (void)hipFree(d);
  p->owned.clear();
  p->xouts.clear();
  p->zc.clear();
  p->loaded = false;
}

}  // namespace

// ================================================================================================== C entry points
extern "C" int ctrlv_pack_weight(const void* src, int src_dtype, int N, int C, int taps, int form, int geglu, void* dst,
                                 int ld_dst, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(src && dst, "pack_weight: null pointer");
  CTRLV_CHECK_ARG(src_dtype >= 0 && src_dtype <= 2 && (form == 0 || form == 1), "pack_weight: bad dtype / form");
  CTRLV_CHECK_SHAPE(N > 0 && C > 0 && taps > 0 && N <= 65535 && C <= 65535, "pack_weight: bad shape");
  if (form == 0) {
    CTRLV_CHECK_SHAPE(ld_dst >= taps * C && (!geglu || N % 32 == 0), "pack_weight: ld_dst < taps * C (or odd GEGLU rows)");
    hipLaunchKernelGGL(pack_weight_rows_kernel, dim3((taps * C + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, src,
                       src_dtype, N, C, taps, (el_t*)dst, ld_dst, geglu);
  } else {
    CTRLV_CHECK_SHAPE(ld_dst % taps == 0 && ld_dst / taps >= N && !geglu, "pack_weight: ld_dst must be taps * Np, Np >= N");
    const int Np = ld_dst / taps;
    hipLaunchKernelGGL(pack_weight_swapped_kernel, dim3((taps * C + 63) / 64, (Np + 31) / 32), dim3(256), 0,
                       (hipStream_t)stream, src, src_dtype, N, C, taps, Np, (el_t*)dst, ld_dst);
  }
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_create(const ctrlv_model_config* cfg, int device, ctrlv_plan** out) {
  CTRLV_CHECK_ARG(cfg && out, "plan_create: null argument");
  ctrlv_plan* p = new ctrlv_plan();
  p->cfg = *cfg;
  p->device = device;
  const int rc = build_graph(p);
  if (rc != CTRLV_OK) { delete p; return rc; }
  *out = p;
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_destroy(ctrlv_plan* p) {
  if (!p) return CTRLV_OK;
  int dev = 0;
  const bool sw = hipGetDevice(&dev) == hipSuccess && dev != p->device;
  if (sw) (void)hipSetDevice(p->device);
  free_owned(p);
  for (auto& r : p->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  if (sw) (void)hipSetDevice(dev);
  delete p;
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_load_weights(ctrlv_plan* p, const ctrlv_tensor_desc* tensors, size_t n) {
  CTRLV_CHECK_ARG(p && tensors, "plan_load_weights: null argument");
  CTRLV_HIP_TRY(hipSetDevice(p->device));
  free_owned(p);
  Loader L;
  L.p = p;
  for (size_t i = 0; i < n; ++i) {
    CTRLV_CHECK_ARG(tensors[i].name && tensors[i].data && tensors[i].dtype >= 0 && tensors[i].dtype <= 2,
                    "plan_load_weights: bad tensor descriptor %zu", i);
    L.map[tensors[i].name] = &tensors[i];
  }
  const ctrlv_model_config& c = p->cfg;
  const int nb = c.n_blocks, c0 = c.block_out_channels[0], ted = 4 * c0;
  auto body = [&]() -> int {
    // input convs: slots [conv_in channels | control_conv_in channels | pad], biases summed (controlnet.py:297-299)
    const int cin_tot = c.in_channels + (c.kind == 1 ? c.in_channels / 2 : 0);
    p->cin_cp = pad_to(cin_tot, 8);
    p->cin_kp = pad_to(9 * p->cin_cp, 64);
    TRY(L.new_linear(p->cin, c0, p->cin_kp, true));
    TRY(L.pack_w("conv_in.weight", c0, c.in_channels, 9, p->cin, p->cin_cp, 0, 0, 0));
    TRY(L.pack_v("conv_in.bias", c0, p->cin.b, 0, 0, 0));
    if (c.kind == 1) {
      TRY(L.pack_w("control_conv_in.weight", c0, c.in_channels / 2, 9, p->cin, p->cin_cp, c.in_channels, 0, 0));
      TRY(L.pack_v("control_conv_in.bias", c0, p->cin.b, 0, 0, 1));
    }
    TRY(L.linear("time_embedding.linear_1", ted, c0, p->te1));
    TRY(L.linear("time_embedding.linear_2", ted, ted, p->te2));
    TRY(L.linear("add_embedding.linear_1", ted, c.projection_class_embeddings_input_dim, p->ae1));
    TRY(L.linear("add_embedding.linear_2", ted, ted, p->ae2));
    // model-wide row-vector GEMMs: offsets first, then storage, then the blocks fill their rows
    int off = 0;
    for_each_res(p, [&](ResBlock& r) { r.temb_off[0] = off; off += r.cout; r.temb_off[1] = off; off += r.cout; });
    p->temb_n = off;
    TRY(L.new_linear(p->temb, off, pad_to(ted, 64), true));
    off = 0;
    for_each_tr(p, [&](Transformer& t) { t.xattn_off[0] = off; off += t.C; t.xattn_off[1] = off; off += t.C; });
    p->xattn_n = off;
    if (off) TRY(L.new_linear(p->xv, off, pad_to(c.cross_attention_dim, 64), false));
    int rc = CTRLV_OK;
    for_each_res(p, [&](ResBlock& r) { if (rc == CTRLV_OK) rc = load_res(L, r, ted); });
    TRY(rc);
    for_each_tr(p, [&](Transformer& t) { if (rc == CTRLV_OK) rc = load_tr(L, t, c.cross_attention_dim); });
    TRY(rc);
    for (int i = 0; i < nb; ++i) {
      if (p->down[i].down.present)
        TRY(L.conv3x3("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", p->down[i].down.C, p->down[i].down.C,
                      p->down[i].down.conv));
      if (c.kind == 0 && p->up[i].up.present)
        TRY(L.conv3x3("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", p->up[i].up.C, p->up[i].up.C, p->up[i].up.conv));
    }
    if (c.kind == 0) {
      TRY(L.norm("conv_norm_out", c0, p->gno));
      TRY(L.conv3x3("conv_out", c.out_channels, c0, p->cout));
    } else {
      // controlnet.py:148-185: one zero-conv per encoder tap (conv_in, every layer, every downsampler) + the mid block
      std::vector<int> ch;
      ch.push_back(c0);
      for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < c.layers_per_block[i]; ++j) ch.push_back(c.block_out_channels[i]);
        if (i != nb - 1) ch.push_back(c.block_out_channels[i]);
      }
      p->zc.resize(ch.size());
      for (size_t i = 0; i < ch.size(); ++i) TRY(L.linear("controlnet_down_blocks." + std::to_string(i), ch[i], ch[i], p->zc[i]));
      TRY(L.linear("controlnet_mid_block", c.block_out_channels[nb - 1], c.block_out_channels[nb - 1], p->zc_mid));
    }
    // frame positional embedding tables for cfg.num_frames (weights-only constants)
    if (c.num_frames > 0) {
      size_t need = 0;
      for_each_tr(p, [&](Transformer& t) {
        const size_t a = ((size_t)c.num_frames * 4 + 255) / 256 * 256 + ((size_t)c.num_frames * t.tpe1.k * 2 + 255) / 256 * 256 +
                         ((size_t)c.num_frames * 4 * t.C * 2 + 255) / 256 * 256 + ((size_t)c.num_frames * t.C * 2 + 255) / 256 * 256;
        if (a > need) need = a;
      });
      void* scratch = nullptr;
      CTRLV_HIP_TRY(hipMalloc(&scratch, need + 1024));
      L.staged.push_back(scratch);
      for_each_tr(p, [&](Transformer& t) {
        if (rc != CTRLV_OK) return;
        rc = L.alloc((size_t)c.num_frames * t.C * 4, (void**)&t.frame_emb, false);
        if (rc != CTRLV_OK) return;
        Ctx cx{p, L.st, false, (char*)scratch, need + 1024};
        cx.B = 1; cx.F = c.num_frames;
        rc = frame_embedding(cx, t, c.num_frames, t.frame_emb);
      });
      TRY(rc);
    }
    return CTRLV_OK;
  };
  const int rc = body();
  const hipError_t e = hipDeviceSynchronize();
  for (void* d : L.staged) (void)hipFree(d);
  if (rc != CTRLV_OK) { free_owned(p); return rc; }
  if (e != hipSuccess) {
    ctrlv_set_error("plan_load_weights: %s", hipGetErrorString(e));
    free_owned(p);
    return CTRLV_E_HIP;
  }
  p->loaded = true;
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_set_time_context_order(ctrlv_plan* p, int order) {
  CTRLV_CHECK_ARG(p != nullptr && (order == 0 || order == 1), "plan_set_time_context_order: order must be 0 (sb) or 1 (bs)");
  p->cfg.time_context_order = order;
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_set_trunk_mode(ctrlv_plan* p, int mode) {
  CTRLV_CHECK_ARG(p != nullptr && (mode == 0 || mode == 1), "plan_set_trunk_mode: mode must be 0 (plain) or 1 (split hi + lo planes)");
  CTRLV_CHECK_ARG(mode == 0 || CTRLV_ELEM_DTYPE == 1, "plan_set_trunk_mode: the split trunk needs the fp16 element library "
                                                      "(libctrlv_hip_f16.so)");
  if (p->trunk_mode != mode) p->ws_cache.clear();       // the workspace grows with the lo planes
  p->trunk_mode = mode;
  return CTRLV_OK;
}

extern "C" int ctrlv_plan_num_down_residuals(ctrlv_plan* p) {
  CTRLV_CHECK_ARG(p != nullptr, "plan: null plan");
  int n = 1;
  for (int i = 0; i < p->cfg.n_blocks; ++i) n += p->cfg.layers_per_block[i] + (i != p->cfg.n_blocks - 1 ? 1 : 0);
  return n;
}

extern "C" int ctrlv_plan_residual_shape(ctrlv_plan* p, int idx, int B, int F, int H, int W, int64_t* rows, int32_t* channels) {
  CTRLV_CHECK_ARG(p && rows && channels, "plan_residual_shape: null argument");
  const int n = ctrlv_plan_num_down_residuals(p);
  CTRLV_CHECK_ARG(idx >= 0 && idx <= n, "plan_residual_shape: index %d out of range [0, %d]", idx, n);
  int k = 0, h = H, w = W, ch = p->cfg.block_out_channels[0];
  auto hit = [&]() { if (k == idx) { *rows = (int64_t)B * F * h * w; *channels = ch; } ++k; };
  hit();
  for (int i = 0; i < p->cfg.n_blocks; ++i) {
    ch = p->cfg.block_out_channels[i];
    for (int j = 0; j < p->cfg.layers_per_block[i]; ++j) hit();
    if (i != p->cfg.n_blocks - 1) { h = (h + 2 - 3) / 2 + 1; w = (w + 2 - 3) / 2 + 1; hit(); }
  }
  hit();      // idx == n: the mid residual (same shape as the last tap)
  return CTRLV_OK;
}

extern "C" size_t ctrlv_plan_workspace_bytes(ctrlv_plan* p, int B, int F, int H, int W) {
  if (check_common(p, B, F, H, W) != CTRLV_OK) return 0;
  Ctx c{p, nullptr, true, nullptr, 0};
  c.B = B; c.F = F;
  const int n_ids = p->cfg.projection_class_embeddings_input_dim / (p->cfg.addition_time_embed_dim > 0 ? p->cfg.addition_time_embed_dim : 1);
  int rc;
  if (p->cfg.kind == 0) {
    // residual adds need no workspace; the dry walk is the same with or without them
    rc = unet_forward(p, c, nullptr, 2, nullptr, 1, nullptr, nullptr, n_ids, nullptr, nullptr, nullptr, nullptr, H, W);
  } else {
    rc = controlnet_forward(p, c, nullptr, nullptr, 2, nullptr, 1, nullptr, nullptr, n_ids, 1.0f, nullptr, nullptr, H, W);
  }
  return rc == CTRLV_OK ? c.peak + 256 : 0;
}

// A too-small workspace is refused BEFORE anything is launched (the walk's own overflow flags are a second line of
// defence: not every launch helper tests them).  The dry walk's result is cached per (B, F, H, W).
static int check_workspace(ctrlv_plan* p, const char* who, int B, int F, int H, int W, size_t have) {
  size_t need = 0;
  for (const auto& e : p->ws_cache)
    if (e.B == B && e.F == F && e.H == H && e.W == W) need = e.bytes;
  if (need == 0) {
    need = ctrlv_plan_workspace_bytes(p, B, F, H, W);
    if (need == 0) return CTRLV_E_BAD_SHAPE;           // (message set by the dry walk)
    if (p->ws_cache.size() >= 16) p->ws_cache.clear();
    p->ws_cache.push_back({B, F, H, W, need});
  }
  if (have < need) {
    ctrlv_set_error("%s: workspace too small (%zu bytes, need >= %zu: ctrlv_plan_workspace_bytes)", who, have, need);
    return CTRLV_E_WORKSPACE;
  }
  return CTRLV_OK;
}

extern "C" int ctrlv_unet_forward(ctrlv_plan* p, const void* sample, int dtype, const float* timestep, int n_timestep,
                                  const void* ehs, const float* added_time_ids, int n_ids, const void* const* down_res,
                                  const void* mid_res, void* residual_event, void* out, int B, int F, int H, int W,
                                  void* workspace, size_t workspace_bytes, ctrlv_stream_t stream) {
  TRY(check_common(p, B, F, H, W));
  CTRLV_CHECK_ARG(p->cfg.kind == 0, "unet_forward: the plan is a ControlNet");
  CTRLV_CHECK_ARG(sample && timestep && ehs && added_time_ids && out && workspace, "unet_forward: null pointer");
  if (dtype < 0 || dtype > 2) { ctrlv_set_error("unet_forward: dtype must be 0 (fp32), 1 (fp16) or 2 (bf16)"); return CTRLV_E_BAD_DTYPE; }
  CTRLV_CHECK_ARG((down_res == nullptr) == (mid_res == nullptr), "unet_forward: pass both down_res and mid_res or neither");
  TRY(check_workspace(p, "unet_forward", B, F, H, W, workspace_bytes));
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  Ctx c{p, (hipStream_t)stream, false, base, workspace_bytes - (size_t)(base - (char*)workspace)};
  c.B = B; c.F = F;
  const int rc = unet_forward(p, c, sample, dtype, timestep, n_timestep, ehs, added_time_ids, n_ids, down_res, mid_res,
                              residual_event, out, H, W);
  if (rc == CTRLV_OK && c.overflow) { ctrlv_set_error("unet_forward: workspace too small (need >= %zu bytes)", c.peak + 256); return CTRLV_E_WORKSPACE; }
  return rc;
}

// Down + mid path of the UNet only (the frozen-UNet half of the training step): the skip tensors and the mid output are
// COPIED out of the arena into the caller's buffers (the caller adds the ControlNet residuals with autograd attached).
static int unet_encoder(ctrlv_plan* p, Ctx& c, const void* sample, int dtype, const float* timestep, int n_t, const void* ehs,
                        const float* ids, int n_ids, void* const* out_taps, void* out_mid, int h, int w) {
  const int N = c.B * c.F;
  TRY(run_context(c, dtype, timestep, n_t, ehs, ids, n_ids));
  Trk x;
  TRY(run_input(c, dtype, sample, nullptr, h, w, &x));
  std::vector<Tap> taps;
  int H, W;
  TRY(run_down_mid(c, x, h, w, taps, &x, &H, &W));
  if (!c.dry) {     // (the training step's tensors are plain element rows: the hi planes)
    for (size_t i = 0; i < taps.size(); ++i) {
      CTRLV_CHECK_ARG(out_taps[i] != nullptr, "unet_encoder_forward: out_taps[%zu] is null", i);
      TRY(ctrlv_axpby(taps[i].x.hi, taps[i].x.hi, 1.0f, 0.0f, out_taps[i], (size_t)N * taps[i].H * taps[i].W * taps[i].C, c.st));
    }
    TRY(ctrlv_axpby(x.hi, x.hi, 1.0f, 0.0f, out_mid, (size_t)N * H * W * p->cfg.block_out_channels[p->cfg.n_blocks - 1], c.st));
  }
  return CTRLV_OK;
}

extern "C" int ctrlv_unet_encoder_forward(ctrlv_plan* p, const void* sample, int dtype, const float* timestep, int n_timestep,
                                          const void* ehs, const float* added_time_ids, int n_ids, void* const* out_taps,
                                          void* out_mid, int B, int F, int H, int W, void* workspace,
                                          size_t workspace_bytes, ctrlv_stream_t stream) {
  TRY(check_common(p, B, F, H, W));
  CTRLV_CHECK_ARG(p->cfg.kind == 0, "unet_encoder_forward: the plan is a ControlNet");
  CTRLV_CHECK_ARG(sample && timestep && ehs && added_time_ids && out_taps && out_mid && workspace,
                  "unet_encoder_forward: null pointer");
  if (dtype < 0 || dtype > 2) { ctrlv_set_error("unet_encoder_forward: dtype must be 0 (fp32), 1 (fp16) or 2 (bf16)"); return CTRLV_E_BAD_DTYPE; }
  TRY(check_workspace(p, "unet_encoder_forward", B, F, H, W, workspace_bytes));     // (sized for the whole forward)
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  Ctx c{p, (hipStream_t)stream, false, base, workspace_bytes - (size_t)(base - (char*)workspace)};
  c.B = B; c.F = F;
  const int rc = unet_encoder(p, c, sample, dtype, timestep, n_timestep, ehs, added_time_ids, n_ids, out_taps, out_mid, H, W);
  if (rc == CTRLV_OK && c.overflow) { ctrlv_set_error("unet_encoder_forward: workspace too small"); return CTRLV_E_WORKSPACE; }
  return rc;
}

extern "C" int ctrlv_controlnet_forward(ctrlv_plan* p, const void* sample, const void* control_cond, int dtype,
                                        const float* timestep, int n_timestep, const void* ehs,
                                        const float* added_time_ids, int n_ids, float conditioning_scale,
                                        void* const* out_down, void* out_mid, int B, int F, int H, int W, void* workspace,
                                        size_t workspace_bytes, ctrlv_stream_t stream) {
  TRY(check_common(p, B, F, H, W));
  CTRLV_CHECK_ARG(p->cfg.kind == 1, "controlnet_forward: the plan is a UNet");
  CTRLV_CHECK_ARG(sample && control_cond && timestep && ehs && added_time_ids && out_down && out_mid && workspace,
                  "controlnet_forward: null pointer (control_cond is required, controlnet.py:289)");
  if (dtype < 0 || dtype > 2) { ctrlv_set_error("controlnet_forward: dtype must be 0 (fp32), 1 (fp16) or 2 (bf16)"); return CTRLV_E_BAD_DTYPE; }
  TRY(check_workspace(p, "controlnet_forward", B, F, H, W, workspace_bytes));
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  Ctx c{p, (hipStream_t)stream, false, base, workspace_bytes - (size_t)(base - (char*)workspace)};
  c.B = B; c.F = F;
  const int rc = controlnet_forward(p, c, sample, control_cond, dtype, timestep, n_timestep, ehs, added_time_ids, n_ids,
                                    conditioning_scale, out_down, out_mid, H, W);
  if (rc == CTRLV_OK && c.overflow) { ctrlv_set_error("controlnet_forward: workspace too small (need >= %zu bytes)", c.peak + 256); return CTRLV_E_WORKSPACE; }
  return rc;
}

// ---- per-launch profile of the plan's own launches (include/ctrlv_hip.h)
extern "C" int ctrlv_plan_profile(ctrlv_plan* p, int enable) {
  CTRLV_CHECK_ARG(p != nullptr, "plan_profile: null plan");
  for (auto& r : p->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  p->prof.clear();
  p->profiling = enable != 0;
  return CTRLV_OK;
}
extern "C" int ctrlv_plan_profile_read(ctrlv_plan* p, ctrlv_profile_record* out, int max_records) {
  CTRLV_CHECK_ARG(p != nullptr && (out != nullptr || max_records == 0), "plan_profile_read: null pointer");
  const int n = (int)p->prof.size();
  if (max_records == 0) return n;                            // (count query)
  const int m = n < max_records ? n : max_records;
  for (int i = 0; i < m; ++i) {
    auto& r = p->prof[i];
    CTRLV_HIP_TRY(hipEventSynchronize(r.e1));
    float ms = 0.f;
    CTRLV_HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
    r.r.ms = ms;
    out[i] = r.r;
  }
  for (auto& r : p->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  p->prof.clear();
  return m;
}
