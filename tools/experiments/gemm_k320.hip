// nn.Linear at Cin = 320 with many rows (the q|k|v and output projections of the 72 x 128 level: M = 460 800), gfx950.
//
// The ping-pong tile serves these layers at 0.34-0.45 of either bound: K = 320 is ten half-steps, so a 256 x 320 tile spends
// about as long in its store-bound epilogue as in its main loop, and one workgroup per CU cannot overlap the two.  This kernel
// is the fused feed-forward's GEMM 1 (ff_fused.hip) on its own: a wave keeps its 32 x rows -- all 20 k-steps -- in REGISTERS
// (80) and walks the output in chunks of 32 columns; a chunk is one chain of 20 MFMAs whose A operands, the chunk's weight
// fragments, arrive by LDS-DMA.  A chunk's epilogue (32 x 32 values through four KiB of the wave's own LDS, 16-B stores)
// runs while the other wave of the SIMD is in its chain, so the store-bound part hides behind MFMAs instead of following them.
//   * 256-row tile, 8 waves x 32 rows, persistent grid; x rows and the next tile's requested a tile ahead (ff_fused's seam).
//   * weights: the row-major packed weight [N, 320] AS IT IS -- LDS-DMA takes a byte offset per lane, so a 1-KiB piece is
//     gathered straight into MFMA A-fragment order (lane (r, h) <- row chunk * 32 + r, k-step ks, half h: 16 B): no packed
//     copy, no plan change; dispatch is by the layer's shape inside ctrlv_gemm.  3-deep ring of 20-KiB chunks, issued two
//     chunks ahead by all eight waves, waited for with a COUNTED vmcnt (the younger stores and residual loads stay in flight).
//   * bias as the C operand of the chain's first MFMA, K in ascending order, s_acc / s_acc2 then ONE fma per residual: the
//     ping-pong tile's operation sequence -- the two kernels give the same bits, so a layer may be served by either.
// Serves: mode 0, Cin = 320, bias [+ R1], element-type output, N % 32 == 0 (ctrlv_gemm_k320_serves).
#include "common.h"
#include "gemm_pp_kernel.h"

namespace {

constexpr int kK = 320, kSteps = kK / 16;                       // 20 k-steps of 16
constexpr int kSlot = kSteps * 1024;                             // a chunk: 32 weight rows x 320 k = 20 KiB of fragments
constexpr int kStg = 0;                                          // staging: 8 waves x 4 KiB
constexpr int kBias = 8 * 4096;                                  // bias strip: N floats (N <= 1024)
constexpr int kRing = kBias + 4096;                              // 3 chunks
constexpr int kSmemK = kRing + 3 * kSlot;
constexpr int kNQk = 6;                                          // fragment reads in flight ahead of the chain

template <int EPI>
__global__ __launch_bounds__(512) void gemm_k320_kernel(const ctrlv_gemm_desc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, l4 = lane & 3;
  const int M = d.M, NC = d.N >> 5;
  const int tiles = (M + 255) / 256, G = gridDim.x;
  constexpr int kFlags = 0x00020000;
  constexpr unsigned kOOB = 0xFFFFFFFFu;

  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.W, 0, d.N * kK * 2, kFlags);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)((long)M * d.lda * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)((long)M * d.ldo * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsR1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((EPI & 2) ? d.R1 : d.W), 0, (EPI & 2) ? (int)((long)M * d.ldr1 * 2) : 0, kFlags);

  // bias strip (zeros without a bias): written once, read as the C operand of every chain
  for (int i = threadIdx.x; i < d.N; i += 512) *(float*)(smem + kBias + i * 4) = d.bias ? d.bias[i] : 0.f;

  // LDS-DMA of one weight chunk: piece ks <- rows chunk*32 + r32, columns ks*16 + 8*hsel .. +8 (the A fragment of k-step ks);
  // wave w takes pieces w, w + 8, w + 16.  `g` = the workgroup's running chunk count (ring phase).
  const unsigned wlane = (unsigned)(r32 * (kK * 2) + hsel * 16);
  auto dma = [&](int chunk, int g) {
    char* s = smem + kRing + (g % 3) * kSlot;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ks = k * 8 + wid;
      if (ks < kSteps)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(s + ks * 1024), 16, wlane, chunk * (32 * kK * 2) + ks * 32, 0, 0);
    }
  };
  dma(0, 0);
  dma(1 % NC, 1);
  wait_vmcnt<0>();
  __syncthreads();

  // x rows: the NEXT tile's are requested at a tile's start into a second register set (160 of the 256 registers hold x) and
  // have its whole length to arrive; the chunk loop's counted waits retire everything older than two rounds, so they ARE
  // there at the seam -- the registers pass through an empty asm so that the compiler does not wait for younger stores.
  elx8 xr[kSteps], xn[kSteps];
  auto tile_loads = [&](int tile) {                              // (rows >= M lie behind the descriptor's end: zeros)
    const unsigned xoff = (unsigned)(tile * 256 + wid * 32 + r32) * (unsigned)(d.lda * 2) + 16 * hsel;
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks)
      xn[ks] = __builtin_bit_cast(elx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff, ks * 32, 0));
  };
  tile_loads(blockIdx.x);
  wait_vmcnt<0>();

  // epilogue geometry of a 32 x 32 sub-tile (gemm_epilogue_lds): row r of the image lives in piece r >> 3 at (r & 7) * 128 B,
  // 16-B chunk c of a row at c ^ (r & 7); a lane writes its row r32 and reads rows lane >> 2 / 16 + (lane >> 2)
  char* const stg = smem + kStg + wid * 4096;
  char* const wrow = stg + (r32 >> 3) * 1024 + (r32 & 7) * 128;
  const int row_a = lane >> 2;
  const char* const rp_a = stg + (row_a >> 3) * 1024 + (row_a & 7) * 128;
  const char* const rp_b = rp_a + 2048;
  const int rx0 = ((l4 * 2) ^ (row_a & 7)) * 16, rx1 = ((l4 * 2 + 1) ^ (row_a & 7)) * 16;
  const unsigned bias_lds = (unsigned)(unsigned long)LDS_PTR(smem + kBias + 16 * hsel);

  int g = 0;                                                     // chunks started so far by this workgroup
  for (int tile = blockIdx.x; tile < tiles; tile += G) {
    const int m0 = tile * 256 + wid * 32 + row_a;                // + pass * 16
    const unsigned o_base = (unsigned)m0 * (unsigned)(d.ldo * 2) + (unsigned)(l4 * 16);
    const unsigned r1_base = (unsigned)m0 * (unsigned)(d.ldr1 * 2) + (unsigned)(l4 * 16);
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks) { asm volatile("" : "+v"(xn[ks])); xr[ks] = xn[ks]; }
    u32x4_t rq[2][2];                                            // residual rows of the current / the next chunk, requested a chunk ahead
    auto res_load = [&](int c, u32x4_t (&q)[2]) {                // (c == NC: two loads behind the descriptor's end -- every
      if constexpr (EPI & 2) {                                   //  round issues the same number of operations: the counted wait)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
          q[pass] = __builtin_amdgcn_raw_buffer_load_b128(rsR1, (c < NC && m0 + pass * 16 < M) ? r1_base : kOOB,
                                                          (pass * 16 * d.ldr1 + c * 32) * 2, 0);
      }
    };
    auto round = [&](int c, u32x4_t (&qc)[2], u32x4_t (&qn)[2]) {
      // The chunk read in this round was issued two rounds ago; everything younger of this wave may stay in flight (vmcnt
      // counts in issue order): per round its pieces, two residual loads (EPI & 2), two stores -- and, in a tile's first two
      // rounds, the 22 loads of the seam (residual rows of chunk 0, the next tile's x rows).
      constexpr int kRound = (EPI & 2) ? 8 : 4;
      if (c < 2) { if (wid < 4) wait_vmcnt<3 + kRound + 22>(); else wait_vmcnt<2 + kRound + 22>(); }
      else { if (wid < 4) wait_vmcnt<3 + kRound>(); else wait_vmcnt<2 + kRound>(); }
      lds_done_barrier();
      dma((c + 2) % NC, g + 2);                                  // (ring slot last read in round c - 1; wraps into the next tile's chunks)
      res_load(c + 1, qn);
      // ---- chain: C = bias of the lane's 16 columns (8 q + 4 hsel + r), then K in ascending order
      const char* s1 = smem + kRing + (g % 3) * kSlot + lane * 16;
      elx8 wq[kNQk];
#pragma unroll
      for (int i = 0; i < kNQk; ++i) wq[i] = *(const elx8*)(s1 + i * 1024);
      f32x4 bq0, bq1, bq2, bq3;
      {
        const unsigned ba = bias_lds + (unsigned)(c * 128);
        asm volatile("ds_read_b128 %0, %1" : "=v"(bq0) : "v"(ba) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(bq1) : "v"(ba) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bq2) : "v"(ba) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(bq3) : "v"(ba) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3));
      }
      f32x16 a1 = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w, bq2.x, bq2.y, bq2.z, bq2.w, bq3.x, bq3.y, bq3.z, bq3.w};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < kSteps; ++ks) {
        a1 = mfma_32x32x16(wq[ks % kNQk], xr[ks], a1);
        if (ks + kNQk < kSteps) wq[ks % kNQk] = *(const elx8*)(s1 + (ks + kNQk) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue of the chunk: LDS transpose, scale, residual, pack, two 16-B stores per lane.  (The staging stores are
      // asm statements: the 12 wait states between an 8-pass MFMA and a read of its result, which the compiler inserts for
      // instructions of its own, are written out -- tools/hazard_scan.py scan_mfma_to_lds_store checks every unit for it.)
      asm volatile("s_nop 11" : "+v"(a1));
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int cc = (2 * qd + hsel) ^ (r32 & 7);
        stg_write16(wrow + cc * 16, a1[4 * qd], a1[4 * qd + 1], a1[4 * qd + 2], a1[4 * qd + 3]);
      }
      __builtin_amdgcn_wave_barrier();
      float4 img[2][2];
      stg_read4x16(rp_a + rx0, rp_a + rx1, rp_b + rx0, rp_b + rx1, img);
      __builtin_amdgcn_wave_barrier();
      const float sc = (c * 32 < d.n_scale2) ? d.s_acc2 : d.s_acc;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float4 v0 = img[pass][0], v1 = img[pass][1];
        float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = o[e] * sc;
        }
        if constexpr (EPI & 2) {
          float f[8];
          const u32x4_t r = qc[pass];
          unpack_elx8(make_uint4(r.x, r.y, r.z, r.w), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s1, f[e], o[e]);
        }
        const uint4 pk = pack_elx8(o);
        const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
        pp_store_out(pv, rsO, m0 + pass * 16 < M ? o_base : kOOB, (pass * 16 * d.ldo + c * 32) * 2);
      }
      ++g;
    };
    res_load(0, rq[0]);
    tile_loads(tile + G);
    for (int c = 0; c < NC; c += 2) {                            // (N % 64 == 0: the residual registers alternate statically)
      round(c, rq[0], rq[1]);
      round(c + 1, rq[1], rq[0]);
    }
  }
  wait_vmcnt<0>();
#endif
}

template <int EPI>
int launch_k320(const ctrlv_gemm_desc& d, hipStream_t stream) {
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = gemm_k320_kernel<EPI>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemK));
    attr_set[dev] = true;
  }
  const int num_cu = ctrlv_num_cu(dev);
  const int tiles = (d.M + 255) / 256;
  int grid = tiles;
  if (tiles > num_cu) {                       // persistent, every workgroup the same number of tiles
    const int rounds = (tiles + num_cu - 1) / num_cu;
    grid = (tiles + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kSmemK, stream, d);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

}  // namespace

// The launches this kernel serves -- a function of the layer (and of M only through a size floor below which the tile has
// nothing to amortise; its results are the ping-pong tile's bit for bit, so the floor does not reach a clip's values).
bool ctrlv_gemm_k320_serves(const ctrlv_gemm_desc& d) {
  const long lim = 0xFFFFFFF0L;
  return d.mode == 0 && d.taps == 1 && d.Cin == kK && (d.w_cin == 0 || d.w_cin == kK) && d.ksplit == 0 && !d.geglu && !d.A2 &&
         !d.act && !d.out_f32 && !d.raw_out && !d.gn_partials && !d.vmode && !d.R2 && !d.R1_lo && !d.R2_lo && !d.out_lo &&
         d.N % 64 == 0 && d.N >= 128 && d.N <= 1024 && d.n_store == d.N && d.M >= 16384 && d.lda >= kK && d.lda % 8 == 0 &&
         d.ldo % 8 == 0 && (!d.R1 || d.ldr1 % 8 == 0) && (long)d.M * d.lda * 2 <= lim && (long)d.M * d.ldo * 2 <= lim &&
         (!d.R1 || (long)d.M * d.ldr1 * 2 <= lim);
}

int ctrlv_gemm_launch_k320(const ctrlv_gemm_desc& d, hipStream_t stream) {
  return d.R1 ? launch_k320<2>(d, stream) : launch_k320<0>(d, stream);
}
