// Weight gradient of the gather-GEMM family on an LDS-DMA ring (cfg5 training step; the
// reference's step is tools/train_video_controlnet.py:451-488: `accelerator.backward(loss)` over the ControlNet).
//
//   dW[n][tap * Cin + c] = sum_m dY[m][n] * A[src(m, tap)][c]
//
// The contraction runs over ROWS, so both MFMA operands are column reads of row-major tiles: the tiles are staged row-major
// in LDS and read with ds_read_b64_tr_b16 (backward.hip's first kernel, wgrad_kernel, has the recipe).  That kernel stages
// through registers (12 global loads + 12 ds_write_b128 per thread and chunk, two barriers per 64 rows) at a 128 x 256
// tile: 87 FLOP per byte staged, and the staging instructions share the issue slots with the MFMAs -- 0.21 of the MFMA
// peak, 50 ms of the 385 ms step (profile of round 6).  This one:
//   * 320 (n) x 256 (k) output tile, 8 waves as 2 (n) x 4 (k), wave tile 160 x 64 = 5 x 2 accumulators of 32 x 32;
//     N = 320 / 640 / 1280 / 2560 / 5120 / 10240 are exact multiples; 7 transposed fragments per 10 MFMAs;
//   * rows advance in chunks of 32 through a ring of 4 LDS slots (36 KiB each: dY as 5, A as 4 panels of [32 rows][64
//     columns], 128-B rows, the tr-read swizzle applied on the per-lane DMA source address); the gather (3x3 / temporal
//     taps, zero padding, the slab's ragged end) is the LDS-DMA's per-lane offset, out of range = zeros;
//   * ONE barrier per chunk (20 MFMAs per wave).  A wave has a chunk's second fragment set and the next chunk's first set
//     in flight while it issues MFMAs; the DMA of chunk c + 4 is issued behind the barrier all waves pass after their last
//     read of chunk c, waits are counted (vmcnt(2 x pieces per wave)): three chunks (~3800 MFMA cycles) of latency cover;
//   * LDS accesses are asm statements: the compiler would put vmcnt(0) in front of LDS reads it knows of while LDS-DMA is
//     in flight (gemm_pp_kernel.h, "Epilogue staging accesses");
//   * the bias gradient rides along: the k tiles of a slab take turns (chunk c belongs to tile c mod ktiles) adding up the dY
//     fragments they hold anyway, with v_dot2c against (1, 1) (with the k-tile-0 workgroups doing all of it they set the
//     launch time: +5.6 % over the 21 shapes of tools/wgrad_bench.py).
// Measured and dropped (round 6, same-device A/B): the two-group ping-pong of gemm_pp_kernel.h (load phase / MFMA phase half a
// chunk apart, two barriers per chunk) is 1.4 % SLOWER than this loop, whose halves already mix 10 MFMAs with 14 reads.
// Grid: (n tile, k tile, row slab), dealt out XCD-aware like wgrad_kernel (the tiles of one slab share its rows in one L2);
// slab partials + the ordered sum of wgrad_reduce_kernel (deterministic), or fp32 atomics without scratch.
// Serves mode 0, stride-1 3x3 without upsampling, the temporal conv; N, Cin multiples of 64; everything else (stride 2,
// the concat operand, tiny M) stays on wgrad_kernel.
#include "common.h"
#include "wgrad_pp.h"

// Diagnostic builds only (tools/ab_build.py NAME -DCTRLV_WP_DIAG=bits; results are garbage): 1 = no MFMAs, 2 = no LDS-DMA in
// the loop, 4 = no fragment reads in the loop -- which resource the loop waits for.  The product library is built with 0.
#ifndef CTRLV_WP_DIAG
#define CTRLV_WP_DIAG 0
#endif


namespace {

constexpr int kBN = 320, kBK = 256, kRows = 32;
constexpr int kPanel = kRows * 128;                 // [32 rows][64 columns] of 16-bit elements
constexpr int kYPanels = kBN / 64, kAPanels = kBK / 64;
constexpr int kAOff = kYPanels * kPanel;
constexpr int kSlot = (kYPanels + kAPanels) * kPanel;
constexpr int kSlots = 4;
constexpr int kSmem = kSlots * kSlot;
constexpr unsigned kOOB = 0xFFFFFFFFu;
constexpr int kFlags = 0x00020000;

struct WpArgs {
  const el_t* A; const el_t* dY; float* dW; float* dbias; float* part; float scale; int torch_layout;
  int M, N, Cin, taps, lda, ldy, mode, H, Wd, F, S, rows_per_slab, ntiles, ktiles, slabs;
  float inv_w, inv_h, inv_s, inv_f;
};

typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct Frag { i32x2_t lo, hi; };

template <int N>
__device__ __forceinline__ void wp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// floor(m / d) for 0 <= m < 2^24 through the reciprocal (the float product is off by at most one)
__device__ __forceinline__ int fdiv(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  const int r = m - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// the 14 transposed reads of one 16-row step: A-operand panel `kh` (2 fragments), dY half-panels (5 fragments)
template <int KS>
__device__ __forceinline__ void read_step(Frag (&f)[7], unsigned a0, unsigned a1, unsigned yA, unsigned yB) {
#define TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  TR(f[5].lo, a0, KS * 2048); TR(f[5].hi, a0, KS * 2048 + 1024);
  TR(f[6].lo, a1, KS * 2048); TR(f[6].hi, a1, KS * 2048 + 1024);
  TR(f[0].lo, yA, KS * 2048); TR(f[0].hi, yA, KS * 2048 + 1024);
  TR(f[1].lo, yB, KS * 2048); TR(f[1].hi, yB, KS * 2048 + 1024);
  TR(f[2].lo, yA, kPanel + KS * 2048); TR(f[2].hi, yA, kPanel + KS * 2048 + 1024);
  TR(f[3].lo, yB, kPanel + KS * 2048); TR(f[3].hi, yB, kPanel + KS * 2048 + 1024);
  TR(f[4].lo, yA, 2 * kPanel + KS * 2048); TR(f[4].hi, yA, 2 * kPanel + KS * 2048 + 1024);
#undef TR
}
// one fragment (two reads) of step KS: I = 5 / 6 the A-operand halves, 0 .. 4 the dY half panels
template <int KS, int I>
__device__ __forceinline__ void read_frag(Frag& f, unsigned a0, unsigned a1, unsigned yA, unsigned yB) {
#define TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  if constexpr (I == 5) { TR(f.lo, a0, KS * 2048); TR(f.hi, a0, KS * 2048 + 1024); }
  else if constexpr (I == 6) { TR(f.lo, a1, KS * 2048); TR(f.hi, a1, KS * 2048 + 1024); }
  else if constexpr ((I & 1) == 0) { TR(f.lo, yA, (I / 2) * kPanel + KS * 2048); TR(f.hi, yA, (I / 2) * kPanel + KS * 2048 + 1024); }
  else { TR(f.lo, yB, (I / 2) * kPanel + KS * 2048); TR(f.hi, yB, (I / 2) * kPanel + KS * 2048 + 1024); }
#undef TR
}
// wait until at most N LDS operations are outstanding; "defines" the fragment registers, so that no consumer (and no
// register copy) can be scheduled above it
template <int N>
__device__ __forceinline__ void wait_frags(Frag (&f)[7]) {
  asm volatile("s_waitcnt lgkmcnt(%14)"
               : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo),
                 "+v"(f[3].hi), "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi), "+v"(f[6].lo), "+v"(f[6].hi)
               : "n"(N));
}
__device__ __forceinline__ elx8 as_elx8(const Frag& f) {
  const i32x4_t v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3);
  return __builtin_bit_cast(elx8, v);
}
// sum of a fragment's eight elements: four v_dot2c against (1, 1) (common.h el_pair_sum), fp32 accumulate
__device__ __forceinline__ float frag_sum(const Frag& f) {
  float s = 0.f;
  const int w[4] = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
#pragma unroll
  for (int i = 0; i < 4; ++i) s = el_pair_sum(__builtin_bit_cast(elx2n, w[i]), s);
  return s;
}

template <int MODE>
__global__ __launch_bounds__(512) void wgrad_pp_kernel(const WpArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  CTRLV_CLOCK_BEGIN();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lin = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bx = lin % a.ntiles, by = (lin / a.ntiles) % a.ktiles, bz = lin / (a.ntiles * a.ktiles);
  const int n0 = bx * kBN, k0 = by * kBK;
  const int ktot = a.taps * a.Cin;
  const int m_lo = bz * a.rows_per_slab;
  const int m_hi = min(a.M, m_lo + a.rows_per_slab);
  const int nchunks = (m_hi - m_lo + kRows - 1) / kRows;
  const int nh = wid & 1, kh = wid >> 1;

  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)((long)a.M * a.ldy * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)((long)a.M * a.lda * 2), kFlags);

  // ---- DMA roles.  Waves 0-3: rows 8w .. 8w+7 of the 5 dY panels; waves 4-7: rows 8(w-4) .. of the 4 A panels.  Lane l
  // writes 16 B at l * 16 of its piece: row l >> 3, PHYSICAL chunk l & 7 = logical chunk (l & 7) ^ swizzle(row).
  // Everything a chunk's issue needs is a register from here on: the loop body is branch-free apart from the role.
  const int prow = (wid & 3) * 8 + (lane >> 3);
  const int lchunk = (lane & 7) ^ (((lane >> 4) & 1) << 2);        // (row >> 1) & 1 of row = 8w + (lane >> 3)
  const bool is_y = wid < 4;
  int pbase[5];                                                    // per-piece byte offset relative to row m; kOOB = never valid
  int pdu[4], pdv[4];                                              // the piece's tap offsets (A waves)
  const int dim_u = MODE == 1 ? a.H : a.F, dim_v = a.Wd, dim_s = a.S;
  const float inv_w = a.inv_w, inv_h = a.inv_h, inv_s = a.inv_s, inv_f = a.inv_f;
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const int col = n0 + 64 * p;
    const int yb = col < a.N ? (col + lchunk * 8) * 2 : (int)kOOB;
    const int kcol = k0 + 64 * (p & 3);
    const bool k_ok = kcol < ktot && p < 4;
    const int tap = k_ok ? kcol / a.Cin : 0;
    const int c = kcol - tap * a.Cin;
    int delta = 0, du = 0, dv = 0;
    if (MODE == 1) { du = tap / 3 - 1; dv = tap % 3 - 1; delta = du * a.Wd + dv; }
    if (MODE == 2) { du = tap - 1; delta = du * a.S; }
    const int ab = k_ok ? (delta * a.lda + c + lchunk * 8) * 2 : (int)kOOB;
    pbase[p] = is_y ? yb : ab;
    if (p < 4) { pdu[p] = du; pdv[p] = dv; }
  }
  const unsigned ld2 = (unsigned)((is_y ? a.ldy : a.lda) * 2);
  // running row state of the NEXT chunk to issue (chunks are issued in order): m, and for the A waves the row's position
  // (MODE 1: image row u, pixel v; MODE 2: frame u, position v inside the frame) -- one division at the start, then
  // 32 rows further per chunk with a conditional wrap (the serves() conditions make one wrap per axis enough)
  int dm = m_lo + prow, du_ = 0, dv_ = 0;
  int step_u = 0, step_v = 0;
  if (MODE == 1) {
    const int q1 = fdiv(dm, dim_v, inv_w);
    dv_ = dm - q1 * dim_v;
    du_ = q1 - fdiv(q1, dim_u, inv_h) * dim_u;
    step_u = kRows / dim_v; step_v = kRows - step_u * dim_v;
  }
  if (MODE == 2) {
    const int q1 = fdiv(dm, dim_s, inv_s);
    dv_ = dm - q1 * dim_s;
    du_ = q1 - fdiv(q1, dim_u, inv_f) * dim_u;
  }
  (void)inv_w; (void)inv_h; (void)inv_s; (void)inv_f;
  auto dma = [&](int c) {                                          // chunk c of this slab into ring slot c & 3
    char* slot = smem + (c & (kSlots - 1)) * kSlot + (wid & 3) * 1024;
    const bool row_ok = dm < m_hi;
    const unsigned rbase = (unsigned)dm * ld2;
    if (is_y) {
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const unsigned voff = (row_ok & (pbase[p] != (int)kOOB)) ? rbase + (unsigned)pbase[p] : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LDS_PTR(slot + p * kPanel), 16, voff, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        bool ok = row_ok & (pbase[p] != (int)kOOB);
        if (MODE != 0) ok = ok & ((unsigned)(du_ + pdu[p]) < (unsigned)dim_u);
        if (MODE == 1) ok = ok & ((unsigned)(dv_ + pdv[p]) < (unsigned)dim_v);
        const unsigned voff = ok ? rbase + (unsigned)pbase[p] : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(slot + kAOff + p * kPanel), 16, voff, 0, 0, 0);
      }
      if (MODE == 1) {
        dv_ += step_v; du_ += step_u;
        const bool wv = dv_ >= dim_v;
        dv_ -= wv ? dim_v : 0; du_ += wv ? 1 : 0;
        du_ -= du_ >= dim_u ? dim_u : 0;
      }
      if (MODE == 2) {
        dv_ += kRows;
        const bool wv = dv_ >= dim_s;
        dv_ -= wv ? dim_s : 0; du_ += wv ? 1 : 0;
        du_ = du_ >= dim_u ? 0 : du_;
      }
    }
    dm += kRows;
  };

  // ---- fragment addresses (backward.hip tr_frag): lane -> row 4 hsel + (i16 >> 2) [+ 8], column 16 ((lane >> 4) & 1) + 4 (i16 & 3)
  // of a 32-column half panel; the second half panel flips bit 2 of the 16-B chunk index
  const int hsel = lane >> 5, i16 = lane & 15;
  const int frow = 4 * hsel + (i16 >> 2), fcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  const int base0 = frow * 128 + (((fcol >> 3) ^ (((frow >> 1) & 1) << 2)) * 16) + (fcol & 7) * 2;
  const int base1 = base0 ^ 64;
  const unsigned smem_u = (unsigned)(unsigned long)LDS_PTR(smem);
  // wave nh = 0: half panels 0 .. 4 of dY, nh = 1: 5 .. 9;  fragment t: even -> yA + (t / 2) panels, odd -> yB + (t / 2) panels
  const unsigned yA0 = smem_u + (nh ? base1 + 2 * kPanel : base0);
  const unsigned yB0 = smem_u + (nh ? base0 + 3 * kPanel : base1);
  const unsigned a00 = smem_u + kAOff + kh * kPanel + base0;
  const unsigned a10 = smem_u + kAOff + kh * kPanel + base1;

  f32x16 acc[5][2];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;
  // bias gradient (column sums of dY): the four kh waves of an n half hold the same dY fragments; wave kh sums fragment
  // kh + 1 (kh = 0: also 0).  The k tiles of a slab all stream the same dY rows: tile `by` takes the chunks c = by (mod
  // ktiles), so that no workgroup carries the whole sum (with the k-tile-0 workgroups doing all of it they finished 10 %
  // behind the rest and set the launch time).  One partial row per (slab, k tile).
  const bool has_bias = a.dbias != nullptr;
  int bias_turn = by;                                       // chunks until this tile's next turn
  bool do_bias = false;
  float bsum0 = 0.f, bsum1 = 0.f;

  dma(0); dma(1); dma(2); dma(3);
  if (is_y) wp_wait_vmcnt<15>(); else wp_wait_vmcnt<12>();
  asm volatile("s_barrier" ::: "memory");
  Frag f0[7], f1[7];
  read_step<0>(f0, a00, a10, yA0, yB0);
  // One half chunk: the 10 MFMAs of the fragment set `cur` with the 14 reads of the NEXT set (`nxt`, step KSN at the given
  // addresses) issued between them, one fragment behind each of the first seven MFMAs -- a wave never sits in a long
  // read-issue phase with the matrix pipe idle (all eight waves leave the barrier together); sched_barrier pins the order.
#define WP_HALF(cur, nxt, KSN, A0, A1, YA, YB)                                                                          \
  do {                                                                                                                  \
    wait_frags<0>(cur);                                                                                                 \
    const elx8 b0 = as_elx8(cur[5]), b1 = as_elx8(cur[6]);                                                              \
    WP_PAIR(cur, nxt, KSN, 0, 5, 6, A0, A1, YA, YB);                                                                    \
    WP_PAIR(cur, nxt, KSN, 1, 0, 1, A0, A1, YA, YB);                                                                    \
    WP_PAIR(cur, nxt, KSN, 2, 2, 3, A0, A1, YA, YB);                                                                    \
    WP_PAIR(cur, nxt, KSN, 3, 4, -1, A0, A1, YA, YB);                                                                   \
    WP_PAIR(cur, nxt, KSN, 4, -1, -1, A0, A1, YA, YB);                                                                  \
    if (do_bias) {                                                                                                      \
      if (kh == 0) { bsum0 += frag_sum(cur[0]); bsum1 += frag_sum(cur[1]); }                                            \
      else if (kh == 1) bsum0 += frag_sum(cur[2]);                                                                      \
      else if (kh == 2) bsum0 += frag_sum(cur[3]);                                                                      \
      else bsum0 += frag_sum(cur[4]);                                                                                   \
    }                                                                                                                   \
  } while (0)
  // the two MFMAs of dY fragment T, the reads of fragments R0 / R1 of the next set behind them
#define WP_PAIR(cur, nxt, KSN, T, R0, R1, A0, A1, YA, YB)                                                               \
  do {                                                                                                                  \
    const elx8 y_ = as_elx8(cur[T]);                                                                                    \
    if (!(CTRLV_WP_DIAG & 1)) acc[T][0] = mfma_32x32x16(y_, b0, acc[T][0]);                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if constexpr (R0 >= 0) { if (!(CTRLV_WP_DIAG & 4)) read_frag<KSN, (R0 >= 0 ? R0 : 0)>(nxt[R0 >= 0 ? R0 : 0], A0, A1, YA, YB); } \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (!(CTRLV_WP_DIAG & 1)) acc[T][1] = mfma_32x32x16(y_, b1, acc[T][1]);                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if constexpr (R1 >= 0) { if (!(CTRLV_WP_DIAG & 4)) read_frag<KSN, (R1 >= 0 ? R1 : 0)>(nxt[R1 >= 0 ? R1 : 0], A0, A1, YA, YB); } \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  } while (0)
  for (int c = 0; c < nchunks; ++c) {
    do_bias = has_bias && bias_turn == 0;
    bias_turn = bias_turn == 0 ? a.ktiles - 1 : bias_turn - 1;
    const unsigned so = (unsigned)((c & (kSlots - 1)) * kSlot);
    WP_HALF(f0, f1, 1, a00 + so, a10 + so, yA0 + so, yB0 + so);
    wait_frags<0>(f1);                                     // every read of chunk c has landed in registers ...
    if (CTRLV_WP_DIAG & 2) wp_wait_vmcnt<0>();
    else if (is_y) wp_wait_vmcnt<10>(); else wp_wait_vmcnt<8>();  // ... and this wave's pieces of chunk c + 1 in LDS
    asm volatile("s_barrier" ::: "memory");
    // slot of chunk c: no wave reads it any more.  The two waves of a SIMD (w, w + 4) have different roles: the A wave issues
    // its pieces (row coordinates, bounds) BEFORE its second MFMA set, the dY wave AFTER -- one wave's address arithmetic
    // runs under the other's MFMAs instead of both leaving the matrix pipe idle behind the barrier
    if (!(CTRLV_WP_DIAG & 2) && !is_y) dma(c + 4);
    const unsigned sn = (unsigned)(((c + 1) & (kSlots - 1)) * kSlot);
    WP_HALF(f1, f0, 0, a00 + sn, a10 + sn, yA0 + sn, yB0 + sn);
    if (!(CTRLV_WP_DIAG & 2) && is_y) dma(c + 4);
  }
#undef WP_HALF
#undef WP_PAIR
  wait_frags<0>(f0);                                       // (the reads issued for the chunk behind the last: discarded)
  wp_wait_vmcnt<0>();

  // ---- results.  D[t][j]: lane holds column k = lane % 32, rows n = (e & 3) + 8 (e >> 2) + 4 (lane / 32).  N and K are
  // multiples of 64 and the 32-wide blocks are aligned: a block is inside or outside as a whole (scalar branches only)
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kb = k0 + 64 * kh + 32 * j, nbk = n0 + 160 * nh + 32 * t;
      if (kb >= ktot || nbk >= a.N) continue;
      const int k = kb + (lane & 31);
      long col = k;                                        // packed (tap-major) K order ...
      if (a.torch_layout) {                                // ... or the parameter's own [N][Cin][taps] layout
        const int tp = k / a.Cin;
        col = (long)(k - tp * a.Cin) * a.taps + tp;
      }
      const int nb = nbk + 4 * (lane >> 5);
      if (a.part) {                                        // deterministic: this slab's partial matrix (packed K order, unscaled)
        float* ps = a.part + ((long)bz * a.N + nb) * ktot + k;
#pragma unroll
        for (int e = 0; e < 16; ++e) ps[(long)((e & 3) + 8 * (e >> 2)) * ktot] = acc[t][j][e];
      } else {
        float* pd = a.dW + (long)nb * ktot + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(pd + (long)((e & 3) + 8 * (e >> 2)) * ktot, acc[t][j][e] * a.scale);
      }
    }
  if (has_bias) {
    const float s0 = bsum0 + __shfl_xor(bsum0, 32), s1 = bsum1 + __shfl_xor(bsum1, 32);   // the two row halves of a step
    const int t0 = kh == 0 ? 0 : kh + 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int t = r == 0 ? t0 : 1;
      const int n = n0 + 160 * nh + 32 * t + (lane & 31);
      if ((r == 0 || kh == 0) && lane < 32 && n < a.N) {
        const float sv = r == 0 ? s0 : s1;
        if (a.part) a.part[(long)a.slabs * a.N * ktot + ((long)bz * a.ktiles + by) * a.N + n] = sv;
        else atomicAdd(a.dbias + n, sv * a.scale);
      }
    }
  }
  CTRLV_CLOCK_END();
#endif
}

}  // namespace

CTRLV_CLOCK_READER(wgrad_pp)

bool ctrlv_wgrad_pp_serves(const ctrlv_gemm_desc& d, const void* dY, int ldy) {
  if (!ctrlv_debug().wgrad_pp) return false;
  if (d.A2 != nullptr || d.A == nullptr) return false;
  if (!(d.mode == 0 || d.mode == 2 || (d.mode == 1 && (d.stride == 0 || d.stride == 1) && d.up == 0 && d.Ho == d.H && d.Wo == d.Wd)))
    return false;
  if (d.N % 64 != 0 || d.Cin % 64 != 0 || ldy % 8 != 0 || d.lda % 8 != 0) return false;
  if (((uintptr_t)d.A | (uintptr_t)dY) & 15) return false;
  if (d.M < 1024 || d.M >= (1 << 24)) return false;
  if ((long)d.M * d.lda * 2 >= (1L << 31) || (long)d.M * ldy * 2 >= (1L << 31)) return false;
  // (the kernel advances a row's (image row, pixel) / (frame, position) by 32 rows with ONE conditional wrap per axis)
  if (d.mode == 1 && (d.H <= 0 || d.Wd <= 0 || d.M % (d.H * d.Wd) != 0 || d.H <= kRows / d.Wd + 1)) return false;
  if (d.mode == 2 && (d.F <= 0 || d.S < kRows || d.M % (d.F * d.S) != 0)) return false;
  return true;
}

// Row slabs: one workgroup per CU at a time; a slab costs its chunks plus writing (and later summing) a 320 x 256 fp32 tile
void ctrlv_wgrad_pp_plan(const ctrlv_gemm_desc& d, ctrlv_wgrad_pp_plan_t* p) {
  const int ktot = d.taps * d.Cin;
  p->ntiles = (d.N + kBN - 1) / kBN;
  p->ktiles = (ktot + kBK - 1) / kBK;
  const int tiles = p->ntiles * p->ktiles;
  const int chunks = (d.M + kRows - 1) / kRows;
  const int num_cu = ctrlv_num_cu(ctrlv_current_device());
  const double t_chunk = 0.9, t_tile = 14.0;               // us: 20 MFMAs per wave at ~0.6 of the pipe rate; 328 KB out + its share of the sum
  int best = 1;
  double best_t = 1e30;
  for (int s = 1; s <= 256 && s <= chunks; ++s) {
    const int cps = (chunks + s - 1) / s;
    const int slabs = (chunks + cps - 1) / cps;
    if (slabs != s) continue;
    const long items = (long)tiles * slabs;
    const long rounds = (items + num_cu - 1) / num_cu;
    const double t = rounds * (cps * t_chunk + t_tile);
    if (t < best_t * 0.999) { best_t = t; best = s; }
  }
  if (ctrlv_debug().wgrad_slabs > 0) best = ctrlv_debug().wgrad_slabs < chunks ? ctrlv_debug().wgrad_slabs : chunks;
  const int cps = (chunks + best - 1) / best;
  p->rows_per_slab = cps * kRows;
  p->slabs = (chunks + cps - 1) / cps;
}

int ctrlv_wgrad_pp_launch(const ctrlv_gemm_desc& d, const void* dY, int ldy, float* dW, float* dbias, float scale,
                          int torch_layout, float* part, const ctrlv_wgrad_pp_plan_t& p, ctrlv_stream_t stream) {
  WpArgs a;
  a.A = (const el_t*)d.A; a.dY = (const el_t*)dY; a.dW = dW; a.dbias = dbias; a.part = part; a.scale = scale;
  a.torch_layout = torch_layout;
  a.M = d.M; a.N = d.N; a.Cin = d.Cin; a.taps = d.taps; a.lda = d.lda; a.ldy = ldy; a.mode = d.mode;
  a.H = d.H; a.Wd = d.Wd; a.F = d.F; a.S = d.S;
  a.rows_per_slab = p.rows_per_slab; a.ntiles = p.ntiles; a.ktiles = p.ktiles; a.slabs = p.slabs;
  a.inv_w = d.mode == 1 ? 1.0f / (float)d.Wd : 0.f;
  a.inv_h = d.mode == 1 ? 1.0f / (float)d.H : 0.f;
  a.inv_s = d.mode == 2 ? 1.0f / (float)d.S : 0.f;
  a.inv_f = d.mode == 2 ? 1.0f / (float)d.F : 0.f;
  const int dev = ctrlv_current_device();
  static bool attr_set[3][CTRLV_MAX_DEVICES] = {};
  const unsigned grid = (unsigned)(p.ntiles * p.ktiles * p.slabs);
#define WP_LAUNCH(MODEV)                                                                                             \
  do {                                                                                                               \
    auto kfn = wgrad_pp_kernel<MODEV>;                                                                               \
    if (!attr_set[MODEV][dev]) {                                                                                     \
      CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));       \
      attr_set[MODEV][dev] = true;                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kSmem, (hipStream_t)stream, a);                                   \
  } while (0)
  if (d.mode == 0) WP_LAUNCH(0); else if (d.mode == 1) WP_LAUNCH(1); else WP_LAUNCH(2);
#undef WP_LAUNCH
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
