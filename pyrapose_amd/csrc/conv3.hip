// Implicit-GEMM convolution with float32-class accuracy at TWO matrix-core units per product ("f16c8"; rounds 1-2 used three
// bf16 MFMAs per product, "bf16x3": the entry points keep that name):
//   x = x_hi + x_lo:  x_hi = f16(x) (11-bit significand, |x| saturates at 65504),  x_lo8 = e5m2((x - x_hi) * 2^12);  same for w
//   x*w ~= x_hi*w_hi                                 v_mfma_f32_32x32x16_f16
//        + (x_hi8*w_lo8 + x_lo8*w_hi8) * 2^-12       v_mfma_scale_f32_32x32x64_f8f6f4 on e5m2 operands (x_hi8 = e5m2(x_hi)), E8M0
//                                                    scale 2^-12: twice the cycles of the f16 form for four times the K
// so a 32-deep k-step of a 32x32 block is 2 f16 MFMAs + 1 scaled MFMA = 4 issue units instead of the 6 of bf16x3, with the
// loads, LDS images and staging of the bf16x3 kernels unchanged.  Measured per launch against bf16x3 on the same box
// (profiles/r03_p16_*): forward / bwd-data 1.16-1.19x, weight gradient 1.15x; error against float64 2.1e-5 (bf16x3: 4.5e-6;
// per-kernel bar 1e-4, tests/test_gpu_conv.py).  Same gather, row spaces, fusions and LDS-staged epilogue as conv.hip.
//
// Plane format ("P16", same packed geometry as before): hi plane = IEEE halves; lo plane = per element TWO e5m2 bytes,
// [e5m2(x_hi) | e5m2((x - x_hi) * 2^12) << 8] for gathered operands (activations, gradients) and the two bytes swapped for
// weights -- so that the two lo-plane fragments a lane reads per 32-deep step (2 x 16 bytes) ARE the 32-byte operands of the
// scaled MFMA: slot pairs are (x_hi8, w_lo8) and (x_lo8, w_hi8), no byte shuffling in the loop.  value = hi + lo8 * 2^-12
// (within 2^-15 of x).  Gradients travel scaled by a power of two (pp_grad_scale: halves stop at 6e-8) and the weight
// gradient divides it out.
//
// Fragment maps (cdna_hip_programming.md section 3): lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7 -> one 16-byte LDS read per operand and plane.  LDS image per operand plane:
// [4 k-octets][rows][8 x 16 bit], rows of octet o rotated by 2*o so that both the 16-byte stores of the staging
// threads (4 threads per row) and the 512-byte fragment reads are bank-conflict free.
#include "conv_common.h"
#include "conv4.h"

// Everything below is compiled once per plane format (PP_FMT = 0: bf16 pairs / bf16x3, PP_FMT = 1: P16 / f16c8; planes_fmt.h) with
// internal linkage; the entry points carry the format in their name and conv3_dispatch.hip forwards pp_* to the build the
// context asks for (pp_ctx_set_planes_format).
#define PP_CAT_(a, b) a##b
#define PP_CAT(a, b) PP_CAT_(a, b)
#if PP_FMT == 1
#define PP_API(name) PP_CAT(name, _fmt1)
#else
#define PP_API(name) PP_CAT(name, _fmt0)
#endif
namespace {
#include "planes_fmt.h"

#include "conv3_shared.h"

#ifndef PP_EPI_GF
#define PP_EPI_GF 4  // output rows in flight per thread in igemm3f's epilogue (8 measured in round 4: see DESIGN.md)
#endif

template <int TM, int TN, bool AP, bool OP>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 2 : ((TM * TN == 4) ? 3 : 4)) void igemm3_kernel(
    const IgemmParams p, const float* __restrict__ g_src, const uint4* __restrict__ g_ahi, const uint4* __restrict__ g_alo,
    const uint4* __restrict__ g_whi, const uint4* __restrict__ g_wlo,
    const float* __restrict__ g_bias, const float* __restrict__ g_addend, const float* __restrict__ g_mask,
    float* __restrict__ g_out, uint2* __restrict__ g_ohi, uint2* __restrict__ g_olo, int w_rows, int w_ld8) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, NO = BK / 8;
  constexpr int SMEM_U4 = 2 * NO * (BM + BN);
  __shared__ __attribute__((aligned(16))) uint4 smem[SMEM_U4];
  uint4* Ahi = smem;
  uint4* Alo = Ahi + NO * BM;
  uint4* Bhi = Alo + NO * BM;
  uint4* Blo = Bhi + NO * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // staging threads: 4 per row (one k-octet = 8 channels each), rows r0 + 64*i
  const int oct = tid & 3, r0 = tid >> 2;
  RowPos rows[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rows[i] = decode_row(p, m0 + r0 + 64 * i);
  long long a_off[TM];
  bool a_ok[TM];

  const int n_taps = p.kh * p.kw;
  const int n_steps = n_taps * (p.Cred / BK);

  float4 ra[TM][2];
  uint4 rah[TM], ral[TM];
  uint4 rbh[TN], rbl[TN];
  int tap = 0, ty = 0, tx = 0, red0 = 0;

  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) a_ok[i] = tap_offset(p, rows[i], ty, tx, &a_off[i]);
  };
  auto load_step = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (AP) {
        if (a_ok[i]) {
          const long long o8 = 2 * (((a_off[i] + red0) >> 3) + oct);  // packed planes: 32-byte groups
          rah[i] = g_ahi[o8];
          ral[i] = g_alo[o8];
        } else {
          rah[i] = make_uint4(0u, 0u, 0u, 0u);
          ral[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      } else if (a_ok[i]) {
        const float4* s4 = reinterpret_cast<const float4*>(g_src + a_off[i] + red0 + 8 * oct);
        ra[i][0] = s4[0];
        ra[i][1] = s4[1];
      } else {
        ra[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        ra[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + r0 + 64 * i;
      if (n < w_rows) {
        const long long off = ((long long)tap * w_rows + n) * w_ld8 + (red0 >> 3) + oct;
        rbh[i] = g_whi[off];
        rbl[i] = g_wlo[off];
      } else {
        rbh[i] = make_uint4(0u, 0u, 0u, 0u);
        rbl[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto advance = [&]() {
    red0 += BK;
    if (red0 >= p.Cred) {
      red0 = 0;
      ++tap;
      ++tx;
      if (tx == p.kw) { tx = 0; ++ty; }
      set_tap();
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      uint4 hi, lo;
      if (AP) {
        hi = rah[i];
        lo = ral[i];
      } else {
        split8(ra[i][0], ra[i][1], &hi, &lo);
      }
      const int slot = oct * BM + ((r0 + 64 * i + 2 * oct) & (BM - 1));
      Ahi[slot] = hi;
      Alo[slot] = lo;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int slot = oct * BN + ((r0 + 64 * i + 2 * oct) & (BN - 1));
      Bhi[slot] = rbh[i];
      Blo[slot] = rbl[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int il = lane & 31, h = lane >> 5;

  set_tap();
  load_step();
  store_step();
  __syncthreads();

  // Measured alternatives that were NOT kept: double-buffered LDS with one barrier per step (64 KB per workgroup,
  // occupancy 3 -> 2: 15-20 % slower); weights streamed by LDS-DMA (global_load_lds_dwordx4, swizzled source,
  // double-buffered B stage): correct but 2-5 % slower -- the DMA issue cost replaces the ds_write cost.
  for (int step = 0; step < n_steps; ++step) {
    const bool more = step + 1 < n_steps;
    if (more) {
      advance();
      load_step();  // in flight under the MFMAs below
    }
    mma_step<TM, TN, BM, BN, 0>(acc, Ahi, Alo, Bhi, Blo, wm * 32 * TM + il, wn * 32 * TN + il, h);
    __syncthreads();  // every wave has read this step's tiles
    if (more) {
      store_step();
      __syncthreads();
    }
  }

  epilogue3<TM, TN, OP>(p, acc, smem, m0, n0, tid, wm, wn, il, h, g_bias, g_addend, g_mask, g_out, g_ohi, g_olo);
}

// ---- igemm3f: the same tile and LDS image with a branch-free, 32-bit-addressed main loop ----
// For gathers that are linear in the tap (div == 1: forward at any stride, bwd-data at stride 1).  Per staged row the
// kernel keeps a byte offset of tap (0,0), the row pitch of its level and a validity bit per tap; every step's
// operand addresses are then 4 VALU per row (mad + add + bit test + select) and the loads are raw buffer loads,
// whose out-of-range offsets return zeros -- padding taps and rows past M need no branch, so the whole k-loop body
// is ONE basic block that the scheduler can interleave under the MFMAs.  Buffers must stay below 2 GiB
// (host-checked; larger tensors take igemm3_kernel).
// Measured and NOT kept (tools/conv_bench.py --shape r3 = exactly three rounds of workgroups, 128x128 tile): a 64-deep k-step
// (twice the MFMAs per barrier pair, 2 workgroups/CU) and register double-buffering of the staged tiles (loads issued two
// steps ahead, 190 VGPRs, 2 workgroups/CU) both land on the same 340-355 TFLOP/s as this loop; dependent MFMAs on one
// accumulator issue back to back at full rate (tools/ubench/mfma_dep.hip).  Ablations of this loop: without global loads
// +30 %, without the f32->bf16 conversion +17 %, with the gathered operand loaded for one tap in nine +12 %, MFMA +
// fragment reads alone 530 TFLOP/s: the staging of the SAME activation rows for each of the nine taps is what is left.
// Measured and NOT kept: walking the grid column-tile-major per XCD (one column tile's 2.4 MB of weight planes resident in
// each 4 MiB L2, activations re-read by four XCDs): 2-5 % slower on the head shapes than the row-tile-major order -- the
// 3x3 gather re-reads its activation rows nine times, so keeping THEM in L2 matters more than the weights.
// Measured and NOT kept: the same loop on v_mfma_f32_16x16x32_bf16 (48 instead of 24 MFMAs per step, lane = (row, octet)
// staging map, un-rotated image): bit-identical results, 20 % slower on every head shape (255-295 vs 325-355 TFLOP/s).
// RL (row list): the launch computes only the 32-row output blocks of a list (g_rl[0] = count, g_rl[1 ..] ascending block
// indices; pp_ctx_set_row_block_skip: the blocks of the data gradient that a non-zero of dy can reach), four (two) blocks
// to a tile; the grid is sized for the dense case and workgroups past the end of the list leave at once.  Rows keep their
// own gather offsets (this loop never assumed that the rows of a tile are consecutive); rl_fill_kernel writes the rest.
template <int TM, int TN, bool AP, bool OP, bool SC = false, bool RL = false>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 2 : ((TM * TN == 4) ? 3 : 4)) void igemm3f_kernel(
    const IgemmParams p, const void* __restrict__ g_a0, const void* __restrict__ g_a1, unsigned a_bytes,
    const void* __restrict__ g_whi, const void* __restrict__ g_wlo, unsigned w_bytes,
    const float* __restrict__ g_bias, const float* __restrict__ g_addend, const float* __restrict__ g_mask,
    float* __restrict__ g_out, uint2* __restrict__ g_ohi, uint2* __restrict__ g_olo, int w_rows, int w_ld8, int splits,
    float* __restrict__ g_ws, const int* __restrict__ g_rl = nullptr, const unsigned char* __restrict__ g_srcflags = nullptr) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, NO = BK / 8;
  constexpr int SMEM_U4 = 2 * NO * (BM + BN);
  constexpr int ES = 4;  // bytes per gathered element: f32, or packed planes (hi at the group's offset, lo 16 bytes behind = rs_a1)
  __shared__ __attribute__((aligned(16))) uint4 smem[SMEM_U4];
  uint4* Ahi = smem;
  uint4* Alo = Ahi + NO * BM;
  uint4* Bhi = Alo + NO * BM;
  uint4* Blo = Bhi + NO * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int n_wg = (int)gridDim.x;
  if (RL) {  // the grid is sized for the dense case: only the first ceil(count / blocks per tile) x n_tiles_n workgroups work,
             // and the XCD remap runs over THEM (over the whole grid the listed tiles would all land on the first XCDs)
    n_wg = ((g_rl[0] + BM / 32 - 1) / (BM / 32)) * p.n_tiles_n * splits;
    if ((int)blockIdx.x >= n_wg) return;  // workgroup-uniform, before any barrier
  }
  const int lbs = xcd_remap((int)blockIdx.x, n_wg);
  const int split = lbs % splits, lb = lbs / splits;  // split-K: `splits` workgroups share one output tile
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int rl_blk[BM / 32];  // RL: first row of each listed block of this tile
  if (RL) {
    const int count = g_rl[0];
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) {
      const int e = tile_m * (BM / 32) + j;
      rl_blk[j] = e < count ? g_rl[1 + e] * 32 : ((p.M + 31) & ~31);
    }
  }
  const int oct = tid & 3, r0 = tid >> 2;
  const int n_taps = p.kh * p.kw;
  const int steps_per_tap = p.Cred / BK;
  const int all_steps = n_taps * steps_per_tap;
  const int s_begin = (int)((long long)all_steps * split / splits), s_end = (int)((long long)all_steps * (split + 1) / splits);

  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_a0), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AP ? g_a1 : g_a0), 0, AP ? a_bytes - 16 : a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_whi), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_wlo), 0, w_bytes, 0x00020000);

  // per staged row: byte offset at tap (0,0), pitch of one tap row, validity bit per tap
  int a_base[TM], a_pitch[TM];
  unsigned a_valid[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t_row = r0 + 64 * i;
    const RowPos r = decode_row(p, RL ? rl_blk[t_row >> 5] + (t_row & 31) : m0 + t_row);
    a_base[i] = ((r.rowbase + r.ybase * r.SW + r.xbase) * p.ld_src + 8 * oct) * ES;
    a_pitch[i] = p.tsign * r.SW * p.ld_src * ES;
    unsigned v = 0;
    int t = 0;
    for (int ty = 0; ty < p.kh; ++ty)
      for (int tx = 0; tx < p.kw; ++tx, ++t) {
        const int sy = r.ybase + ty * p.tsign, sx = r.xbase + tx * p.tsign;
        bool ok = r.ok && (unsigned)sy < (unsigned)r.SH && (unsigned)sx < (unsigned)r.SW;
        // RL with the flags of the gathered tensor (sparse bwd-data): a source row outside the flagged 32-row blocks IS zero by
        // the meaning of the flags -- it is not fetched, so a producer may leave such rows unwritten (pp_ctx_set_row_block_lazy)
        if (RL && g_srcflags && ok) ok = g_srcflags[(r.rowbase + sy * r.SW + sx) >> 5] != 0;
        if (ok) v |= 1u << t;
      }
    a_valid[i] = v;
  }
  const int x_pitch = p.tsign * p.ld_src * ES;
  int b_base[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + r0 + 64 * i;
    b_base[i] = n < w_rows ? (n * w_ld8 + oct) * 16 : PP_BUF_OOB;
  }
  const int b_tap = w_rows * w_ld8 * 16;

  float4 ra[TM][2];
  uint4 rah[TM], ral[TM];
  uint4 rbh[TN], rbl[TN];
  int tap = s_begin / steps_per_tap, red0 = (s_begin - tap * steps_per_tap) * BK;
  int ty = tap / p.kw, tx = tap - ty * p.kw;

  auto load_step = [&]() {
    const int a_uni = tx * x_pitch + red0 * ES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int vo = a_base[i] + __mul24(ty, a_pitch[i]) + a_uni;
      vo = ((a_valid[i] >> tap) & 1u) ? vo : PP_BUF_OOB;
      if (AP) {
        rah[i] = buf_load16(rs_a0, vo, 0);
        ral[i] = buf_load16(rs_a1, vo, 0);
      } else {
        const uint4 q0 = buf_load16(rs_a0, vo, 0), q1 = buf_load16(rs_a0, vo + 16, 0);
        ra[i][0] = *reinterpret_cast<const float4*>(&q0);
        ra[i][1] = *reinterpret_cast<const float4*>(&q1);
      }
    }
    const int b_uni = ((p.w_ty0 + p.w_tstep * ty) * p.w_kw + p.w_tx0 + p.w_tstep * tx) * b_tap + (red0 >> 3) * 16;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      rbh[i] = buf_load16(rs_wh, b_base[i], b_uni);
      rbl[i] = buf_load16(rs_wl, b_base[i], b_uni);
    }
  };
  auto advance = [&](bool more) {  // uniform, branch-free; past the last step it rewinds to step 0 (a harmless re-load)
    red0 += BK;
    const bool wrap = red0 >= p.Cred;
    red0 = wrap ? 0 : red0;
    tap += wrap ? 1 : 0;
    tx += wrap ? 1 : 0;
    const bool wx = tx == p.kw;
    tx = wx ? 0 : tx;
    ty += wx ? 1 : 0;
    red0 = more ? red0 : 0;
    tap = more ? tap : 0;
    tx = more ? tx : 0;
    ty = more ? ty : 0;
  };
  auto split_step = [&]() {
    if (!AP) {
#pragma unroll
      for (int i = 0; i < TM; ++i) split8(ra[i][0], ra[i][1], &rah[i], &ral[i]);
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int slot = oct * BM + ((r0 + 64 * i + 2 * oct) & (BM - 1));
      Ahi[slot] = rah[i];
      Alo[slot] = ral[i];
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int slot = oct * BN + ((r0 + 64 * i + 2 * oct) & (BN - 1));
      Bhi[slot] = rbh[i];
      Blo[slot] = rbl[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int il = lane & 31, h = lane >> 5;

  load_step();
  split_step();
  store_step();
  __syncthreads();

  for (int step = s_begin; step < s_end; ++step) {
    advance(step + 1 < s_end);
    load_step();  // in flight under the MFMAs below
    // (left alone the scheduler sinks the loads below the MFMAs, or pulls the conversion of the loaded tile -- and
    // with it the wait for the loads -- up to the first MFMA: pin loads | first half of the MFMAs | rest + conversion)
    __builtin_amdgcn_sched_barrier(0);
    mma_step<TM, TN, BM, BN, 2>(acc, Ahi, Alo, Bhi, Blo, wm * 32 * TM + il, wn * 32 * TN + il, h);
    split_step();     // conversion VALU issues while this wave's MFMAs drain
    __syncthreads();  // every wave has read this step's tiles
    store_step();
    __syncthreads();
  }
  if (splits > 1) {  // partial sums only (slice `split` of the scratch): splitk_finish_kernel sums the slices in a fixed
                     // order and applies the epilogue -> deterministic, no atomics, nothing to zero
    // (RL: slice rows are COMPACT -- list position * 32 + row in the block, i.e. m0 + tile row -- and every row of a listed
    // block is written, in range or not: rl_splitk_finish_kernel reads them by list position)
    float* slice = g_ws + (long long)split * (RL ? (((long long)p.M + 31) & ~31ll) : (long long)p.M) * p.ld_out;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + trow;
        const bool row_ok = RL ? (rl_blk[trow >> 5] < p.M) : (m < p.M);
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int co = n0 + wn * 32 * TN + b * 32 + il;
          if (row_ok && co < ((p.Nout + 3) & ~3)) slice[(long long)m * p.ld_out + co] = acc[a][b][r];
        }
      }
    return;
  }
  epilogue3<TM, TN, OP, PP_EPI_GF, SC, false, RL>(p, acc, smem, m0, n0, tid, wm, wn, il, h, g_bias, g_addend, g_mask, g_out, g_ohi, g_olo, rl_blk);
}

// ---- igemm3x: 3-wide stride-1 "same" convolutions with the gathered tile shared by the three taps of a kernel row ----
// For a fixed kernel row ty the source pixel of output row m at tap tx is (m + dy * W(m)) + dx with dx = off_x + tx * tsign
// in {-1, 0, 1}: the tile of tap dx is the tile of dx = 0 shifted by one row.  The workgroup stages its BM rows once per
// (ty, 32-channel chunk) and the three taps read their fragments at row offsets -1, 0, +1; so that every output row finds
// both neighbours inside the tile, consecutive tiles overlap by two rows: a tile covers output rows m0 .. m0 + BM - 1 but
// only writes the BM - 2 inner ones (1.6 % of the MFMAs are spent on the two edge rows).  A row shifted across an
// image-row / image / pyramid-level boundary lands on the wrong pixel exactly where the tap is padding: the per-lane
// tap-validity bit zeroes that fragment.  Global loads, f32->bf16 conversions and LDS writes of the gathered operand drop
// to a third.  Loop order: ty, channel chunk, tx (unrolled: the body of each tx is one basic block).
// Conditions (host-checked): kw == 3, stride 1, source space == enumerated space (SH == OH, SW == OW, same row offsets),
// f32 source, no planes / scatter.
// Where the time went before the iglp_opt hint below (profiles/r01_pmc_stalls.txt, 128x128; 65 % MFMA-busy with it): MFMA pipe 58 % busy, VALU 19 % (8 % under an MFMA), LDS unit
// 34 %, no bank conflicts; for 31 % of the cycles all three resident waves of a SIMD wait (barriers, LDS / global
// latency).  Tried against that: a double-buffered weight tile (one barrier per tap instead of two, 48 KB of LDS): the
// kernel alone gains 1 %, the training step loses 1 % (less room for the other lane's workgroups on the CU) -- not kept;
// s_setprio 2 / 0 around the MFMA bursts (the hipBLASLt habit): 1 % slower alone and in the step.
// CAP: the bf16 (hi, lo) split of the gathered f32 operand is also written out ([rows][ld_src] planes, the geometry of
// pp_split_planes_bf16x3): for the centre kernel row (dy = 0: the staged rows are the tile's own rows; the BM - 2 inner
// rows of all tiles cover every row once) the workgroups store the registers they have just converted, the output-channel
// tiles of one row tile taking turns over the channel chunks.  The weight-gradient launch of the same layer then reads both operands pre-split (pp_ctx_set_split_capture).
// AP: the gathered operand is stored as bf16 (hi, lo) planes (g_a = hi, g_a1 = lo; ld_src % 8 == 0): the staging threads move
// 16 B of each plane per row and chunk -- the same bytes as the two f32 quads -- and nothing is converted in the loop.
// OP: the output tile is written as planes (g_ohi / g_olo; also as f32 when g_out != NULL).
template <int TM, int TN, bool CAP, bool AP = false, bool OP = false>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 2 : ((TM * TN == 4) ? 3 : 4)) void igemm3x_kernel(
    const IgemmParams p, const void* __restrict__ g_a, const void* __restrict__ g_a1, unsigned a_bytes, const void* __restrict__ g_whi,
    const void* __restrict__ g_wlo, unsigned w_bytes, const float* __restrict__ g_bias, const float* __restrict__ g_addend,
    const float* __restrict__ g_mask, float* __restrict__ g_out, uint2* __restrict__ g_ohi, uint2* __restrict__ g_olo, int w_rows, int w_ld8,
    int splits, float* __restrict__ g_ws, void* __restrict__ g_chi, void* __restrict__ g_clo, const unsigned char* __restrict__ g_flags,
    int skip_halo) {
  static_assert(!(CAP && AP), "split capture is for f32 operands");
  constexpr int ES = 4;  // bytes per gathered element: f32, or packed planes (hi at the group's offset, lo 16 bytes behind = rs_a1)
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, NO = BK / 8;
  constexpr int AP1 = NO * BM + 1;  // plane size: the tile + one all-zero slot that padded taps read instead of their row
  constexpr int SMEM_U4 = 2 * AP1 + 2 * NO * BN;
  __shared__ __attribute__((aligned(16))) uint4 smem[SMEM_U4];
  uint4* Ahi = smem;
  uint4* Alo = Ahi + AP1;
  uint4* Bhi = Alo + AP1;
  uint4* Blo = Bhi + NO * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lbs = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = lbs % splits, lb = lbs / splits;
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = p.m_off + tile_m * (BM - 2) - 1, n0 = tile_n * BN;  // tile row j <-> output row m0 + j; rows 0 and BM - 1 are halo only
  const int oct = tid & 3, r0 = tid >> 2;
  const int il = lane & 31, h = lane >> 5;
  const int n_chunks = p.Cred / BK;
  const int all_groups = p.kh * n_chunks;  // one group = the three taps of (ty, chunk)
  const int g_begin = (int)((long long)all_groups * split / splits);
  int g_end = (int)((long long)all_groups * (split + 1) / splits);
  if (g_flags) {
    // row-block skip (pp_ctx_set_row_block_skip): when none of the 32-row blocks of the gathered tensor that this tile can
    // reach (its rows +- one image row +- one pixel; skip_halo = widest level + 1) holds a non-zero, the sum is exactly
    // zero: no k-loop, the epilogue still writes bias / addend / mask.  Workgroup-uniform.
    int lo = m0 - skip_halo, hi = m0 + BM - 1 + skip_halo;
    lo = lo < 0 ? 0 : lo;
    hi = hi > p.M - 1 ? p.M - 1 : hi;
    int any = 0;
    for (int b = lo >> 5; b <= (hi >> 5); ++b) any |= g_flags[b];
    g_end = any ? g_end : g_begin;
  }

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_a), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AP ? g_a1 : g_a), 0, AP ? a_bytes - 16 : a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_whi), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_wlo), 0, w_bytes, 0x00020000);

  // staged rows r0 + 64 * i: byte offset of the dx = 0 source pixel at ty = 0, row pitch (low 4 bits: source row y + dy exists, per ty)
  int s_base[TM], s_pitch[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int q = m0 + r0 + 64 * i;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    const bool ok = q >= 0 && r.ok;
    const int x = r.xbase - p.off_x;  // dx = 0 <=> the output cell's own column
    s_base[i] = ((r.rowbase + r.ybase * r.SW + x) * p.ld_src + 8 * oct) * ES;
    int v = 0;
    for (int ty = 0; ty < p.kh; ++ty)
      if (ok && (unsigned)(r.ybase + ty * p.tsign) < (unsigned)r.SH) v |= 1 << ty;
    s_pitch[i] = (p.tsign * r.SW * p.ld_src * ES) | v;  // ld_src % 4 == 0 -> the pitch is a multiple of 16
  }
  // fragment rows of this lane: wm * 32 * TM + a * 32 + il -> validity bit per tap (9 bits each, two rows per register)
  unsigned f_valid[(TM + 1) / 2];
#pragma unroll
  for (int a = 0; a < (TM + 1) / 2; ++a) f_valid[a] = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int q = m0 + wm * 32 * TM + a * 32 + il;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    unsigned v = 0;
    int t = 0;
    for (int ty = 0; ty < p.kh; ++ty)
      for (int tx = 0; tx < 3; ++tx, ++t) {
        const int sy = r.ybase + ty * p.tsign, sx = r.xbase + tx * p.tsign;
        if (q >= 0 && r.ok && (unsigned)sy < (unsigned)r.SH && (unsigned)sx < (unsigned)r.SW) v |= 1u << t;
      }
    f_valid[a >> 1] |= v << (16 * (a & 1));
  }
  int b_base[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + r0 + 64 * i;
    b_base[i] = n < w_rows ? (n * w_ld8 + oct) * 16 : PP_BUF_OOB;
  }
  const int b_tap = w_rows * w_ld8 * 16;

  float4 ra[TM][2];
  uint4 rah[TM], ral[TM];
  uint4 rbh[TN], rbl[TN];
  // group g <-> (chunk, ty).  Default (p.x_ty_inner, PP_CONV3_X_ORDER=1): kernel row innermost -- the three row-shifted tiles of
  // one channel chunk are fetched back to back (they overlap by all but one image row, so the second and third come from
  // L1 / L2 while they are hot): same time, FETCH_SIZE -10 % on the 512-wide head conv and -40 % on the 256-wide one
  // (profiles/r02_traffic.json).  PP_CONV3_X_ORDER=0: channel chunk innermost (round 1).
  const bool ty_inner = p.x_ty_inner != 0;
  int ty = ty_inner ? g_begin % p.kh : g_begin / n_chunks;
  int chunk = ty_inner ? g_begin / p.kh : g_begin - ty * n_chunks;  // group being LOADED

  auto load_a = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int vo = s_base[i] + __mul24(ty, s_pitch[i] & ~15) + chunk * (BK * ES);
      vo = ((s_pitch[i] >> ty) & 1) ? vo : PP_BUF_OOB;
      if (AP) {
        rah[i] = buf_load16(rs_a, vo, 0);
        ral[i] = buf_load16(rs_a1, vo, 0);
      } else {
        const uint4 q0 = buf_load16(rs_a, vo, 0), q1 = buf_load16(rs_a, vo + 16, 0);
        ra[i][0] = *reinterpret_cast<const float4*>(&q0);
        ra[i][1] = *reinterpret_cast<const float4*>(&q1);
      }
    }
  };
  auto load_b = [&](int tx) {
    const int b_uni = ((p.w_ty0 + ty) * p.w_kw + tx) * b_tap + chunk * (BK / 8 * 16);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      rbh[i] = buf_load16(rs_wh, b_base[i], b_uni);
      rbl[i] = buf_load16(rs_wl, b_base[i], b_uni);
    }
  };
  auto split_a = [&]() {
    if (!AP) {
#pragma unroll
      for (int i = 0; i < TM; ++i) split8(ra[i][0], ra[i][1], &rah[i], &ral[i]);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int slot = oct * BM + ((r0 + 64 * i + 2 * oct) & (BM - 1));
      Ahi[slot] = rah[i];
      Alo[slot] = ral[i];
    }
  };
  // CAP: (ty, chunk) is the group whose converted rows the registers hold; plane byte offset = f32 byte offset / 2
  const int ty_c = -p.off_y * p.tsign;  // kernel row with dy = 0
  auto capture = [&]() {
    if constexpr (CAP) {
      const __amdgpu_buffer_rsrc_t rs_ch = __builtin_amdgcn_make_buffer_rsrc(g_chi, 0, a_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_cl = __builtin_amdgcn_make_buffer_rsrc(g_clo, 0, a_bytes - 16, 0x00020000);
      if (ty == ty_c && chunk % p.n_tiles_n == tile_n) {  // uniform: the output-channel tiles of a row tile take turns
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = r0 + 64 * i;
          const int vo = s_base[i] + __mul24(ty, s_pitch[i] & ~15) + chunk * (BK * 4);
          const bool ok = ((s_pitch[i] >> ty) & 1) && row >= 1 && row <= BM - 2;
          const int co = ok ? vo : PP_BUF_OOB;  // packed planes: the f32 byte offset of the group
          buf_store16(rs_ch, rah[i], co);
          buf_store16(rs_cl, ral[i], co);
        }
      }
    }
  };
  auto store_b = [&]() {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int slot = oct * BN + ((r0 + 64 * i + 2 * oct) & (BN - 1));
      Bhi[slot] = rbh[i];
      Blo[slot] = rbl[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // multiply the tile in LDS as tap (c_ty, TX): fragments of the gathered operand at row offset dx, zeroed where the tap pads
  auto mma_tile = [&](int c_ty, auto tx_c) {
    constexpr int TX = decltype(tx_c)::value;
    const int dx = p.off_x + TX * p.tsign;
    const unsigned tap_bits = (1u << (c_ty * 3 + TX)) * 0x10001u;  // the tap's bit in both 16-bit halves
    unsigned okm[(TM + 1) / 2];
#pragma unroll
    for (int a = 0; a < (TM + 1) / 2; ++a) okm[a] = f_valid[a] & tap_bits;
    unsigned a_ok = 0;
#pragma unroll
    for (int a = 0; a < TM; ++a) a_ok |= (((okm[a >> 1] >> (16 * (a & 1))) & 0xffffu) != 0 ? 1u : 0u) << a;
    // padded taps read the all-zero slot: one address select per fragment
    mma_step<TM, TN, BM, BN, 1>(acc, Ahi, Alo, Bhi, Blo, wm * 32 * TM + il + dx, wn * 32 * TN + il, h, a_ok, NO * BM);
  };

  std::integral_constant<int, 0> t0;
  std::integral_constant<int, 1> t1;
  std::integral_constant<int, 2> t2;
  if (threadIdx.x == 0) {
    Ahi[NO * BM] = make_uint4(0u, 0u, 0u, 0u);
    Alo[NO * BM] = make_uint4(0u, 0u, 0u, 0u);
  }
  load_a();
  load_b(0);
  split_a();
  store_a();
  store_b();
  __syncthreads();
  for (int g = g_begin; g < g_end; ++g) {
    const int c_ty = ty;
    // (CAP) the converted rows of this group are still in registers: their stores go out ahead of the tap's loads and
    // complete under its MFMAs (one vmcnt for loads and stores: issued later they would stall the next weight tile)
    capture();
    // tx = 0: the weight tile of tx = 1 is fetched under it
    load_b(1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(c_ty, t0);
    __syncthreads();
    store_b();
    __syncthreads();
    // tx = 1
    load_b(2);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(c_ty, t1);
    __syncthreads();
    store_b();
    __syncthreads();
    // tx = 2: the next group's gathered rows and its first weight tile are fetched, converted and written
    {
      const bool more = g + 1 < g_end;  // past the end: rewind to group 0 (a harmless re-load)
      if (ty_inner) {
        ty += 1;
        const bool wt = ty == p.kh;
        ty = wt ? 0 : ty;
        chunk += wt ? 1 : 0;
      } else {
        chunk += 1;
        const bool wc = chunk == n_chunks;
        chunk = wc ? 0 : chunk;
        ty += wc ? 1 : 0;
      }
      chunk = more ? chunk : 0;
      ty = more ? ty : 0;
    }
    load_b(0);
    load_a();
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(c_ty, t2);
    split_a();
    __syncthreads();
    store_b();
    store_a();
    __syncthreads();
  }
  // rows 0 and BM - 1 of the tile are halo: clear their accumulators' way out by making them out of range
  if (splits > 1) {
    float* slice = g_ws + (long long)split * (p.M - p.m_off) * p.ld_out;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + row;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int co = n0 + wn * 32 * TN + b * 32 + il;
          if (row >= 1 && row <= BM - 2 && m < p.M && co < ((p.Nout + 3) & ~3)) slice[(long long)(m - p.m_off) * p.ld_out + co] = acc[a][b][r];
        }
      }
    return;
  }
  epilogue3<TM, TN, OP, 4, false, true>(p, acc, smem, m0, n0, tid, wm, wn, il, h, g_bias, g_addend, g_mask, g_out, g_ohi, g_olo);
}

// ---- igemm4x: igemm3x's arithmetic and tap-row reuse on a multi-stage LDS-DMA pipeline (PP_CONV3_DMA=0 turns it off) ----
// Why: the 128 x 128, two-barriers-per-tap loop of igemm3x prefetches one tap ahead through registers; with an L2 hit rate of 90 % a
// wave's four weight loads per tap contain a miss every third tap, and three waves per SIMD cannot cover it (46-59 % of the wave
// cycles in s_waitcnt, MFMA pipe 48 % busy).  Here: ONE workgroup of 8 waves per CU, tile 256 x 128 (wave tile 64 x 64 as
// before), both operands reach LDS by buffer_load ... lds (no staging registers) into rings -- 2 stages of the gathered tile (one
// per (ty, chunk) group, loaded a group = three taps ahead), 3 stages of the weight tile (the stage of a tap is its tx; loaded two
// taps ahead) -- that stay in flight ACROSS the one barrier per tap; the waits are counted (vmcnt 6 / 2 / 2), never 0 in the loop.
//  * A wave-instruction of LDS-DMA writes 64 lane-consecutive 16-byte slots, so the LDS image is shaped on the SOURCE side: the
//    lane at slot s fetches the piece that belongs at s.  Images (planes_fmt.h LAY 1): row-major, XOR-swizzled pieces -- one
//    instruction moves 8 rows x 128 contiguous bytes of the gathered operand ([hi0 lo0 .. hi3 lo3] of a row's 32-channel chunk:
//    full lines; fragment-shaped 16-byte pieces of 64 different lines per instruction cost 15 % of the kernel) or 16 rows x 64
//    bytes of a weight plane.  Padding rows are out-of-range offsets (the DMA writes zeros).
//  * Every stage is an LDS object of its own: the compiler tracks LDS-DMA per object (alias scopes), so the waits it inserts in
//    front of a fragment read are COUNTED and cover the DMA into that stage only (one shared array: vmcnt(0) before every read).
//  * One workgroup per CU makes a last, partly filled round cost a whole round: the host gives whole rounds to this kernel and
//    the remaining rows to igemm3x with the reduction split over a round's worth of workgroups (IgemmParams::m_off).
// Same products in the same order as igemm3x: the rows of the whole rounds are bit-identical to it.  Planes in, planes out,
// 3x3 stride 1, kernel row innermost (x_ty_inner), w_rows % 128 == 0, at least two rounds of tiles (PP_CONV3_DMA_MIN).
// Measured (P16, same box): 422-434 us against 499 on a tail-free 512-channel shape (1.16x), the regression-head launch 486 against
// 577 us (1.19x, 490 TFLOP/s); bf16 pairs 1.04x; training step +3.1 %.
// NWM = 4: the form above (8 waves, 256 x 128, one workgroup per CU, weight ring of three stages = 112 KB of LDS).
// NWM = 2 (round 4): 4 waves, 128 x 128, TWO workgroups per CU -- gathered ring of 2 x 16 KB + weight ring of 2 x 16 KB = 64 KB
// each -- for the launches that cannot fill two rounds of 256-row tiles (the 256-channel heads, the FPN 3x3): the weight tile is
// fetched ONE tap ahead (into the stage the previous tap has just left), the gathered tile two taps ahead; the second workgroup
// of the CU covers the waits.  Same products in the same order as igemm3x either way.
template <bool OP, int VAR = 0, int NWM = 4>
__global__ __launch_bounds__(128 * NWM, NWM == 4 ? 1 : 2) void igemm4x_kernel(
    const IgemmParams p, const void* __restrict__ g_a, const void* __restrict__ g_a1, unsigned a_bytes, const void* __restrict__ g_whi,
    const void* __restrict__ g_wlo, unsigned w_bytes, const float* __restrict__ g_bias, const float* __restrict__ g_addend,
    const float* __restrict__ g_mask, float* __restrict__ g_out, uint2* __restrict__ g_ohi, uint2* __restrict__ g_olo, int w_rows, int w_ld8) {
  constexpr int TM = 2, TN = 2, NW = 2 * NWM, BM = 32 * TM * NWM, BN = 128, BK = 32, NO = BK / 8, ES = 4;
  constexpr int BST = NWM == 4 ? 3 : 2;  // stages of the weight ring
  constexpr int BI = 128 / (16 * NW);    // LDS-DMA instructions per wave and weight plane (16 rows x 64 bytes each)
  constexpr int A_STAGE = 8 * BM + 2, B_STAGE = 2 * NO * BN;  // (gathered: 8 pieces per row + the two zero slots; weights: hi + lo regions)
  // every stage is an LDS object of its own: the compiler's LDS-DMA tracking (alias scopes per object) then inserts COUNTED vmcnt
  // waits in front of a fragment read -- for the DMA into that stage only -- instead of draining everything in flight
  __shared__ __attribute__((aligned(16))) uint4 sA0[A_STAGE];
  __shared__ __attribute__((aligned(16))) uint4 sA1[A_STAGE];
  __shared__ __attribute__((aligned(16))) uint4 sB0[B_STAGE];
  __shared__ __attribute__((aligned(16))) uint4 sB1[B_STAGE];
  __shared__ __attribute__((aligned(16))) uint4 sB2[BST == 3 ? B_STAGE : 1];
  auto stA = [&](auto k) -> uint4* {
    if constexpr (decltype(k)::value == 0) return sA0; else return sA1;
  };
  auto stB = [&](auto k) -> uint4* {
    if constexpr (decltype(k)::value == 0) return sB0; else return sB1;
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = tile_m * (BM - 2) - 1, n0 = tile_n * BN;
  const int il = lane & 31, h = lane >> 5;
  const int n_chunks = p.Cred / BK;
  const int G = p.kh * n_chunks;  // groups: (chunk, ty), ty innermost

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_a), 0, a_bytes, 0x00020000);
  (void)g_a1;  // (packed planes: the lo halves are pieces of the same 128-byte chunks, reached through g_a)
  const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_whi), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_wlo), 0, w_bytes, 0x00020000);

  // DMA lanes (LDS images of planes_fmt.h LAY 1, filled in FULL lines).  Gathered tile: one instruction = 8 rows x 8 pieces (the
  // 128 contiguous bytes [hi0 lo0 .. hi3 lo3] of a row's 32-channel chunk); wave w owns rows 32 w .. 32 w + 31 in four instructions;
  // the lane at LDS piece position q of row r fetches piece q ^ ((r / 2) mod 8).  Weight tile, per plane: one instruction = 16
  // rows x 4 pieces (64 contiguous bytes); wave w owns rows 16 (BI w + i) .. + 15, i < BI; piece q of row n is octet q ^ ((n / 4) mod 4).
  int s_base[4], s_pitch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = 32 * wave + 8 * c + (lane >> 3);
    const int piece = (lane & 7) ^ ((j >> 1) & 7);
    const int q = m0 + j;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    const bool ok = q >= 0 && r.ok;
    const int x = r.xbase - p.off_x;
    s_base[c] = (r.rowbase + r.ybase * r.SW + x) * p.ld_src * ES + 16 * piece;
    int v = 0;
    for (int ty = 0; ty < 3; ++ty)
      if (ok && (unsigned)(r.ybase + ty * p.tsign) < (unsigned)r.SH) v |= 1 << ty;
    s_pitch[c] = (p.tsign * r.SW * p.ld_src * ES) | v;
  }
  int b_dma[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int b_row = 16 * (BI * wave + i) + (lane >> 2);
    const int b_n = n0 + b_row;
    b_dma[i] = b_n < w_rows ? (b_n * w_ld8 + ((lane & 3) ^ ((b_row >> 2) & 3))) * 16 : PP_BUF_OOB;
  }
  const int b_tap = w_rows * w_ld8 * 16;

  unsigned f_valid = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int q = m0 + wm * 32 * TM + a * 32 + il;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    unsigned v = 0;
    int t = 0;
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx, ++t) {
        const int sy = r.ybase + ty * p.tsign, sx = r.xbase + tx * p.tsign;
        if (q >= 0 && r.ok && (unsigned)sy < (unsigned)r.SH && (unsigned)sx < (unsigned)r.SW) v |= 1u << t;
      }
    f_valid |= v << (16 * (a & 1));
  }

  auto dma_a = [&](int g, uint4* st) {  // the gathered tile of group g -> stage st
    const int ty = g % 3, chunk = g / 3;
    uint4* const dst = st + 8 * 32 * wave;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int vo = s_base[c] + __mul24(ty, s_pitch[c] & ~15) + chunk * (BK * ES);
      vo = ((s_pitch[c] >> ty) & 1) ? vo : PP_BUF_OOB;
      dma16(rs_a, dst + 64 * c, vo, 0);
    }
  };
  auto dma_b = [&](int g, int tx, uint4* st) {  // the weight tile of tap (group g, tx) -> stage st
    const int ty = g % 3, chunk = g / 3;
    const int b_uni = ((p.w_ty0 + ty) * p.w_kw + tx) * b_tap + chunk * (BK / 8 * 16);
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      uint4* const hi = st + 4 * 16 * (BI * wave + i);
      dma16(rs_wh, hi, b_dma[i], b_uni);
      dma16(rs_wl, hi + NO * BN, b_dma[i], b_uni);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto mma_tile = [&](int c_ty, int TX, const uint4* Ahi, const uint4* Bhi) {
    const int dx = p.off_x + TX * p.tsign;
    const unsigned tap_bits = (1u << (c_ty * 3 + TX)) * 0x10001u;
    const unsigned okm = f_valid & tap_bits;
    unsigned a_ok = 0;
#pragma unroll
    for (int a = 0; a < TM; ++a) a_ok |= (((okm >> (16 * (a & 1))) & 0xffffu) != 0 ? 1u : 0u) << a;
    if (VAR & 2) __builtin_amdgcn_s_setprio(1);
    mma_step<TM, TN, BM, BN, (VAR & 4) ? 2 : ((VAR & 1) ? 0 : 1), 1>(acc, Ahi, Ahi, Bhi, Bhi + NO * BN, wm * 32 * TM + il + dx, wn * 32 * TN + il, h, a_ok, 8 * BM);
    if (VAR & 2) __builtin_amdgcn_s_setprio(0);
  };
  std::integral_constant<int, 0> c0;
  std::integral_constant<int, 1> c1;
  auto tap_end = [&](auto n) {  // all but the n youngest DMAs of this wave have landed; then every wave's have
    constexpr int N = decltype(n)::value;
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else static_assert(N == 0, "tap_end: count");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // BST == 3: weight ring of THREE stages: the stage of a tap is its tx (compile time without unrolling over groups), the tile of tap
  // t + 2 goes into the stage tap t - 1 has just left; gathered ring of two stages (unrolled over two groups), loaded one group ahead.
  // Issue order at a tap's start: [gathered tile of the next group, at tx == 0], weight tile of tap t + 2.  At the end of tap t the
  // tile of tap t + 1 (issued at tap t - 1) must have landed: instructions issued after it = tx 0: 4 (A) + 2 = 6; tx 1: 2;
  // tx 2: 2 -- and the gathered tile of group g + 1 (issued at tx 0, before the weight tile waited for at tx 1) is then in.
  // BST == 2: the stage of a tap is the parity of its index in a PAIR of groups (u = 3 K + tx); at a tap's start the weight tile of
  // tap t + 1 goes into the other stage (all waves have left it: the barrier that ended tap t - 1), then -- at tx == 0 -- the
  // gathered tile of the next group; at the tap's end the weight tile must be in: everything but the 4 gathered DMAs issued behind
  // it at tx 0 (they have until the end of tx 1), everything at tx 1 and tx 2.
  auto group = [&](int g, auto k) {
    constexpr int K = decltype(k)::value;
    const int c_ty = g % 3;
    const int gn = g + 1 < G ? g + 1 : 0;  // past the end: group 0 again (a harmless re-load)
    uint4* const Ahi = stA(std::integral_constant<int, K & 1>{});
    if constexpr (BST == 3) {
      dma_a(gn, stA(std::integral_constant<int, (K + 1) & 1>{}));
      dma_b(g, 2, sB2);
      mma_tile(c_ty, 0, Ahi, sB0);
      tap_end(std::integral_constant<int, 6>{});
      dma_b(gn, 0, sB0);
      mma_tile(c_ty, 1, Ahi, sB1);
      tap_end(std::integral_constant<int, 2>{});
      dma_b(gn, 1, sB1);
      mma_tile(c_ty, 2, Ahi, sB2);
      tap_end(std::integral_constant<int, 2>{});
    } else {
      constexpr int U = 3 * K;
      dma_b(g, 1, stB(std::integral_constant<int, (U + 1) & 1>{}));
      dma_a(gn, stA(std::integral_constant<int, (K + 1) & 1>{}));
      mma_tile(c_ty, 0, Ahi, stB(std::integral_constant<int, U & 1>{}));
      tap_end(std::integral_constant<int, 4>{});
      dma_b(g, 2, stB(std::integral_constant<int, U & 1>{}));
      mma_tile(c_ty, 1, Ahi, stB(std::integral_constant<int, (U + 1) & 1>{}));
      tap_end(std::integral_constant<int, 0>{});
      dma_b(gn, 0, stB(std::integral_constant<int, (U + 1) & 1>{}));
      mma_tile(c_ty, 2, Ahi, stB(std::integral_constant<int, U & 1>{}));
      tap_end(std::integral_constant<int, 0>{});
    }
  };

  if (tid == 0) {  // the two all-zero slots (hi, lo = slot ^ 1) behind each gathered stage
    sA0[8 * BM] = make_uint4(0u, 0u, 0u, 0u);
    sA0[8 * BM + 1] = make_uint4(0u, 0u, 0u, 0u);
    sA1[8 * BM] = make_uint4(0u, 0u, 0u, 0u);
    sA1[8 * BM + 1] = make_uint4(0u, 0u, 0u, 0u);
  }
  // VAR & 8 (round 4, measured A/B): static priority for the second-dispatched half of an 8-wave workgroup (cdna_hip_programming.md T5,
  // static form: the younger waves lose the issue arbitration on every segment)
  if ((VAR & 8) && NWM == 4 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // prologue: group 0 and the weight tiles of taps 0, 1 (BST == 2: of tap 0)
  dma_a(0, sA0);
  dma_b(0, 0, sB0);
  if constexpr (BST == 3) {
    dma_b(0, 1, sB1);
    tap_end(std::integral_constant<int, 2>{});
  } else {
    tap_end(std::integral_constant<int, 0>{});
  }
  for (int g = 0; g < G; g += 2) {
    group(g, c0);
    if (g + 1 >= G) break;
    group(g + 1, c1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads of a group that never comes: landed before LDS is reused
  __syncthreads();
  // (NWM == 2: a gathered stage holds 16 KB, so the epilogue stages 32 rows at a time)
  epilogue3<TM, TN, OP, 4, false, true, false, NWM, 128 * NWM, (NWM == 2 ? 1 : 0)>(p, acc, sA0, m0, n0, tid, wm, wn, il, h, g_bias, g_addend, g_mask, g_out, g_ohi, g_olo);
}

// epilogue arithmetic of the pointwise finishing kernels: v (+ addend) (masked by the ReLU source) (ReLU) -> f32 and / or planes;
// mo = row of the addend / mask / output tensors, co = first of four columns
__device__ __forceinline__ void finish4(const IgemmParams& p, float4 v, long long mo, int co, bool with_addend, const float* __restrict__ g_addend,
                                        const float* __restrict__ g_mask, float* __restrict__ g_out, void* __restrict__ g_ohi,
                                        void* __restrict__ g_olo) {
  if (with_addend) {
    if (p.add_hi) {
      const float4 a = planes_ld4(p.add_hi, p.add_lo, (mo * p.ld_add + co) >> 2);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    } else if (g_addend) {
      const float4 a = *reinterpret_cast<const float4*>(g_addend + mo * p.ld_add + co);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
  }
  if (p.mask_hi || g_mask) {
    const float4 k = p.mask_hi ? hi_ld4(p.mask_hi, (mo * p.ld_mask + co) >> 2) : *reinterpret_cast<const float4*>(g_mask + mo * p.ld_mask + co);
    v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
  }
  if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (g_out) *reinterpret_cast<float4*>(g_out + mo * p.ld_out + co) = v;
  if (g_ohi) planes_st4(g_ohi, g_olo, (mo * p.ld_out + co) >> 2, v);
}

// parity class without taps (e.g. the odd cells of a 1x1 stride-2 conv): dx = mask?(addend or 0) at the class rows
__global__ void class_fill_kernel(const IgemmParams p, const float* __restrict__ g_addend, const float* __restrict__ g_mask,
                                  float* __restrict__ g_out, void* __restrict__ g_ohi, void* __restrict__ g_olo) {
  const int n4 = (p.Nout + 3) >> 2;
  const long long total = (long long)p.M * n4;
  const int hw = p.seg[0].OH * p.seg[0].OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / n4), co = 4 * (int)(i - (long long)m * n4);
    const int n = m / hw, rem = m - n * hw, yq = rem / p.seg[0].OW, xq = rem - yq * p.seg[0].OW;
    const long long mo = (long long)(n * p.sc_H + 2 * yq + p.sc_cy) * p.sc_W + 2 * xq + p.sc_cx;
    finish4(p, make_float4(0.f, 0.f, 0.f, 0.f), mo, co, true, g_addend, g_mask, g_out, g_ohi, g_olo);
  }
}

// out = relu?(mask?(sum_s ws[s] + bias + addend)) over [M][ceil4(Nout)]
__global__ void splitk_finish_kernel(const IgemmParams p, int splits, const float* __restrict__ ws, const float* __restrict__ g_bias,
                                     const float* __restrict__ g_addend, const float* __restrict__ g_mask, float* __restrict__ g_out,
                                     void* __restrict__ g_ohi, void* __restrict__ g_olo) {
  const int n4 = (p.Nout + 3) >> 2;
  const int rows = p.M - p.m_off;  // (m_off: the launch covers the rows from there on)
  const long long total = (long long)rows * n4, slice = (long long)rows * p.ld_out;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int mr = (int)(i / n4), co = 4 * (int)(i - (long long)mr * n4);
    const int m = p.m_off + mr;
    const float* w = ws + (long long)mr * p.ld_out + co;
    float4 v = *reinterpret_cast<const float4*>(w);
    for (int s = 1; s < splits; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(w + s * slice);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (g_bias) { const float4 b = *reinterpret_cast<const float4*>(g_bias + co); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    finish4(p, v, (long long)m, co, true, g_addend, g_mask, g_out, g_ohi, g_olo);
  }
}

// split-K of the row-list launch: block j of the list <-> compact slice rows 32 j .. 32 j + 31
__global__ void rl_splitk_finish_kernel(const IgemmParams p, int splits, const float* __restrict__ ws, const int* __restrict__ g_rl,
                                        const float* __restrict__ g_bias, const float* __restrict__ g_addend, const float* __restrict__ g_mask,
                                        float* __restrict__ g_out, void* __restrict__ g_ohi, void* __restrict__ g_olo) {
  const int j = blockIdx.x;
  if (j >= g_rl[0]) return;
  const int n4 = (p.Nout + 3) >> 2;
  const long long slice = (((long long)p.M + 31) & ~31ll) * p.ld_out;
  const int m0 = g_rl[1 + j] * 32, nr = min(32, p.M - m0);
  for (int i = threadIdx.x; i < nr * n4; i += blockDim.x) {
    const int r = i / n4, co = 4 * (i - r * n4);
    const float* w = ws + ((long long)j * 32 + r) * p.ld_out + co;
    float4 v = *reinterpret_cast<const float4*>(w);
    for (int s = 1; s < splits; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(w + s * slice);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (g_bias) { const float4 b = *reinterpret_cast<const float4*>(g_bias + co); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    finish4(p, v, (long long)(m0 + r), co, true, g_addend, g_mask, g_out, g_ohi, g_olo);
  }
}

// ---- activation / gradient split: f32 [rows][ld] -> bf16 hi/lo planes with the same geometry ----
__global__ void split_planes_kernel(size_t n8, const float4* __restrict__ src, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                    const float* __restrict__ scale) {
  const float sc = scale ? *scale : 1.f;  // (a power of two: exact)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    uint4 h, l;
    float4 a = src[2 * i], b = src[2 * i + 1];
    a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
    b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
    split8(a, b, &h, &l);
    hi[2 * i] = h;  // packed planes: group i = 32 bytes, hi then lo (lo = hi + 16 bytes)
    lo[2 * i] = l;
  }
}

// The power of two the gradient chain travels multiplied by (halves stop at 6e-8, loss gradients are ~1e-7): every loss
// gradient is bounded by ~1 / max(1, positives of its head) (focal: |dL/dlogit| <= 1 / n, orthogonal_l1: <= 0.125 / n), so
// 2^G = 2^8 * 2^floor(log2(max(1, min positives))) keeps the largest element near 2^8 and leaves 2^8 of headroom for the
// growth through the backward chain; scale2 = {2^G, 2^-G}.
__global__ void grad_scale_kernel(const int* __restrict__ counts, int n_counts, float* __restrict__ scale2, int base_log2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int n = 0x7fffffff;
    for (int i = 0; i < n_counts; ++i) n = counts[i] < n ? counts[i] : n;
    n = n < 1 ? 1 : n;
    int e = 0;
    while ((n >> (e + 1)) != 0) ++e;
    const float s = __uint_as_float((unsigned)(127 + base_log2 + e) << 23);
    scale2[0] = s;
    scale2[1] = 1.f / s;
  }
}

// ---- which 32-row blocks of a gradient tensor hold a non-zero? (input of the row-block skip of bwd-weight / bwd-data) ----
__global__ void row_block_flags_kernel(const float* __restrict__ x, int rows, int ld, int cols4, unsigned char* __restrict__ flags) {
  const int blk = blockIdx.x, r0 = blk * 32;
  const int nr = rows - r0 < 32 ? rows - r0 : 32;
  int any = 0;
  for (int i = threadIdx.x; i < nr * cols4; i += blockDim.x) {
    const int r = i / cols4, c = i - r * cols4;
    const float4 v = *reinterpret_cast<const float4*>(x + (long long)(r0 + r) * ld + 4 * c);
    any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);  // NaN != 0: a block with a NaN is kept
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) flags[blk] = any ? 1 : 0;
}

// the same scan of a tensor stored as bf16 (hi, lo) planes: value != 0 <=> a magnitude bit is set in hi or lo
// within (may be NULL): only the blocks flagged there can hold a non-zero (the caller knows the others are zero): they are not read
__global__ void row_block_flags_planes_kernel(const uint2* __restrict__ hi, const uint2* __restrict__ lo, int rows, int ld4, int cols4,
                                              unsigned char* __restrict__ flags, const unsigned char* __restrict__ within) {
  const int blk = blockIdx.x, r0 = blk * 32;
  if (within && !within[blk]) {  // (uniform)
    if (threadIdx.x == 0) flags[blk] = 0;
    return;
  }
  const int nr = rows - r0 < 32 ? rows - r0 : 32;
  int any = 0;
  for (int i = threadIdx.x; i < nr * cols4; i += blockDim.x) {
    const int r = i / cols4, c = i - r * cols4;
    const long long o = pk4((long long)(r0 + r) * ld4 + c);
    const uint2 h = hi[o], l = lo[o];
    any |= fmt_any_nonzero(h.x | h.y, l.x | l.y) ? 1 : 0;
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) flags[blk] = any ? 1 : 0;
}

// ordered compaction by ONE workgroup: list[0] = count, list[1 ..] = the flagged block indices, ascending
__global__ void row_block_compact_kernel(const unsigned char* __restrict__ flags, int n_blocks, int* __restrict__ list) {
  __shared__ int wave_cnt[4];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n_blocks; i0 += 256) {
    const int i = i0 + tid;
    const bool f = i < n_blocks && flags[i] != 0;
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (f) list[1 + off + before] = i;
    __syncthreads();
    if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) list[0] = base;
}

}  // namespace
extern "C" int PP_API(pp_row_block_list)(pp_ctx* ctx, const float* x, int rows, int ld, int cols, unsigned char* flags, int* list) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x && flags && list && rows > 0 && cols > 0 && cols <= ld && ld % 4 == 0 && pp_is_aligned16(x), PP_ERR_ARG,
               "pp_row_block_list: bad tensor (ld %% 4 == 0, 16-byte aligned)");
  const int nb = (rows + 31) / 32, cols4 = (cols + 3) / 4;
  PP_CHECK_ARG(ctx, 4 * cols4 <= ld, PP_ERR_SHAPE, "pp_row_block_list: cols rounded up to 4 exceed ld");
  hipLaunchKernelGGL(row_block_flags_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, x, rows, ld, cols4, flags);
  hipLaunchKernelGGL(row_block_compact_kernel, dim3(1), dim3(256), 0, ctx->stream, (const unsigned char*)flags, nb, list);
  PP_CHECK_LAUNCH(ctx, "pp_row_block_list");
  return PP_OK;
}
namespace {

}  // namespace
extern "C" int PP_API(pp_row_block_list_planes_within)(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols,
                                               const unsigned char* within, unsigned char* flags, int* list);
namespace {
}  // namespace
extern "C" int PP_API(pp_row_block_list_planes)(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, unsigned char* flags,
                                        int* list) {
  return PP_API(pp_row_block_list_planes_within)(ctx, x_hi, x_lo, rows, ld, cols, nullptr, flags, list);
}
namespace {

}  // namespace
extern "C" int PP_API(pp_row_block_list_planes_within)(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols,
                                               const unsigned char* within, unsigned char* flags, int* list) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x_hi && x_lo && flags && list && rows > 0 && cols > 0 && cols <= ld && ld % 8 == 0 && pp_is_packed(x_hi, x_lo), PP_ERR_ARG,
               "pp_row_block_list_planes: bad tensor (ld %% 8 == 0, packed planes)");
  const int nb = (rows + 31) / 32, cols4 = (cols + 3) / 4;
  PP_CHECK_ARG(ctx, 4 * cols4 <= ld, PP_ERR_SHAPE, "pp_row_block_list_planes: cols rounded up to 4 exceed ld");
  hipLaunchKernelGGL(row_block_flags_planes_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const uint2*)x_hi, (const uint2*)x_lo, rows,
                     ld / 4, cols4, flags, within);
  hipLaunchKernelGGL(row_block_compact_kernel, dim3(1), dim3(256), 0, ctx->stream, (const unsigned char*)flags, nb, list);
  PP_CHECK_LAUNCH(ctx, "pp_row_block_list_planes");
  return PP_OK;
}
namespace {

static int split_planes_impl(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale);
}  // namespace
extern "C" int PP_API(pp_split_planes_bf16x3)(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo) {
  return split_planes_impl(ctx, n, src, hi, lo, nullptr);
}
namespace {
}  // namespace
extern "C" int PP_API(pp_split_planes_scaled_bf16x3)(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale_dev) {
  return split_planes_impl(ctx, n, src, hi, lo, scale_dev);
}
namespace {
}  // namespace
extern "C" int PP_API(pp_grad_scale_from_counts_adj)(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev, int log2_adjust) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, counts_dev && scale2_dev && n_counts >= 1 && n_counts <= 16 && log2_adjust >= -16 && log2_adjust <= 16, PP_ERR_ARG,
               "pp_grad_scale_from_counts: bad arguments");
  static const int base_log2 = []() { const char* e = getenv("PP_GSCALE_LOG2"); const int v = e ? atoi(e) : 8; return v < -20 ? -20 : (v > 30 ? 30 : v); }();
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(64), 0, ctx->stream, counts_dev, n_counts, scale2_dev, base_log2 + log2_adjust);
  PP_CHECK_LAUNCH(ctx, "pp_grad_scale_from_counts");
  return PP_OK;
}
extern "C" int PP_API(pp_grad_scale_from_counts)(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev) {
  return PP_API(pp_grad_scale_from_counts_adj)(ctx, counts_dev, n_counts, scale2_dev, 0);
}
namespace {
static int split_planes_impl(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && hi && lo && n % 8 == 0, PP_ERR_ARG, "pp_split_planes_bf16x3: n must be a multiple of 8");
  PP_CHECK_ARG(ctx, pp_is_aligned16(src) && pp_is_packed(hi, lo), PP_ERR_ALIGN, "pp_split_planes_bf16x3: alignment / packed planes (lo = hi + 16 bytes)");
  if (n == 0) return PP_OK;
  size_t blocks = (n / 8 + 255) / 256;
  const size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n / 8, (const float4*)src, (uint4*)hi, (uint4*)lo, scale);
  PP_CHECK_LAUNCH(ctx, "pp_split_planes_bf16x3");
  return PP_OK;
}

// ---- weight split: f32 HWIO [tap*cin + ci][ld_w] -> bf16 hi/lo planes in both k-contiguous layouts ----
__device__ __forceinline__ void split_weights_tile(int tap, int ci0, int co0, int cin, int cout, int ld_w, const float* __restrict__ w,
                                                   unsigned short* __restrict__ fwd_hi, unsigned short* __restrict__ fwd_lo, int cout_rows,
                                                   unsigned short* __restrict__ dg_hi, unsigned short* __restrict__ dg_lo, int dg_ld) {
  // 32x32 (ci x co) tiles through LDS so that both layouts are written with contiguous rows
  __shared__ unsigned short t_hi[32][33], t_lo[32][33];
  const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int ci = ci0 + j, co = co0 + tx;
    float v = 0.f;
    if (ci < cin && co < cout) v = w[((long long)tap * cin + ci) * ld_w + co];
    unsigned short uh, ul;
    fmt_encode_weight(v, &uh, &ul);
    t_hi[j][tx] = uh;
    t_lo[j][tx] = ul;
    if (dg_hi && ci < cin && co < dg_ld) {  // bwd-data layout: [tap][ci][co], co contiguous
      const long long o = ((long long)tap * cin + ci) * dg_ld + co;
      dg_hi[o] = uh;
      dg_lo[o] = ul;
    }
  }
  __syncthreads();
  if (fwd_hi) {
    for (int j = ty; j < 32; j += 8) {  // forward layout: [tap][co][ci], ci contiguous
      const int co = co0 + j, ci = ci0 + tx;
      if (co < cout_rows && ci < cin) {
        const long long o = ((long long)tap * cout_rows + co) * cin + ci;
        fwd_hi[o] = t_hi[tx][j];
        fwd_lo[o] = t_lo[tx][j];
      }
    }
  }
}

__global__ void split_weights_kernel(int taps, int cin, int cout, int ld_w, const float* __restrict__ w,
                                     unsigned short* __restrict__ fwd_hi, unsigned short* __restrict__ fwd_lo, int cout_rows,
                                     unsigned short* __restrict__ dg_hi, unsigned short* __restrict__ dg_lo, int dg_ld) {
  split_weights_tile(blockIdx.z, blockIdx.y * 32, blockIdx.x * 32, cin, cout, ld_w, w, fwd_hi, fwd_lo, cout_rows, dg_hi, dg_lo, dg_ld);
}

// every tensor of a model in ONE launch (the optimizer step re-splits ~80 tensors; 80 launches of ~5 us were 2 % of a step)
__global__ void split_weights_batch_kernel(int n_jobs, const pp_split_job* __restrict__ jobs) {
  int lo = 0, hi = n_jobs - 1;  // last job with tile_begin <= blockIdx.x (uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pp_split_job j = jobs[lo];
  const int dg_ld = (j.cout + 31) / 32 * 32;
  const int tiles_co = dg_ld / 32, tiles_ci = j.cin / 32;
  int t = (int)blockIdx.x - j.tile_begin;
  const int co_t = t % tiles_co;
  t /= tiles_co;
  const int ci_t = t % tiles_ci, tap = t / tiles_ci;
  split_weights_tile(tap, ci_t * 32, co_t * 32, j.cin, j.cout, j.ld_w, j.w, (unsigned short*)j.fwd_hi, (unsigned short*)j.fwd_lo, j.cout,
                     (unsigned short*)j.dg_hi, (unsigned short*)j.dg_lo, dg_ld);
}

}  // namespace
extern "C" int PP_API(pp_conv_split_weights_bf16x3_batch)(pp_ctx* ctx, int n_jobs, const pp_split_job* jobs_dev, int total_tiles) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_jobs >= 0 && total_tiles >= 0 && (n_jobs == 0 || jobs_dev), PP_ERR_ARG, "pp_conv_split_weights_bf16x3_batch: bad arguments");
  if (n_jobs == 0 || total_tiles == 0) return PP_OK;
  hipLaunchKernelGGL(split_weights_batch_kernel, dim3((unsigned)total_tiles), dim3(32, 8), 0, ctx->stream, n_jobs, jobs_dev);
  PP_CHECK_LAUNCH(ctx, "pp_conv_split_weights_bf16x3_batch");
  return PP_OK;
}
namespace {

}  // namespace
extern "C" int PP_API(pp_conv_split_weights_bf16x3)(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd_hi, void* fwd_lo,
                                            void* dg_hi, void* dg_lo) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv_split_weights_bf16x3");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, w && ((fwd_hi && fwd_lo) || (dg_hi && dg_lo)), PP_ERR_ARG, "pp_conv_split_weights_bf16x3: null tensor");
  PP_CHECK_ARG(ctx, d->cin % 32 == 0, PP_ERR_SHAPE, "pp_conv_split_weights_bf16x3: cin %d must be a multiple of 32", d->cin);
  const int taps = d->kh * d->kw;
  const int dg_ld = (d->cout + 31) / 32 * 32;
  dim3 grid((unsigned)((dg_ld + 31) / 32), (unsigned)((d->cin + 31) / 32), (unsigned)taps);
  hipLaunchKernelGGL(split_weights_kernel, grid, dim3(32, 8), 0, ctx->stream, taps, d->cin, d->cout, d->ld_w, w,
                     (unsigned short*)fwd_hi, (unsigned short*)fwd_lo, d->cout, (unsigned short*)dg_hi, (unsigned short*)dg_lo, dg_ld);
  PP_CHECK_LAUNCH(ctx, "pp_conv_split_weights_bf16x3");
  return PP_OK;
}
namespace {

static bool igemm3_fast_ok(const IgemmParams& p, bool planes, int w_rows, int w_ld8) {
  // the branch-free loop needs a tap-linear gather and 31-bit byte offsets (see igemm3f_kernel)
  static const bool fast_on = []() { const char* e = getenv("PP_CONV3_FAST"); return !(e && e[0] == '0'); }();
  const long long a_bytes = p.src_rows * (long long)p.ld_src * 4;  // (packed planes take the same 4 bytes per element)
  (void)planes;
  const long long w_bytes = (long long)p.w_taps * w_rows * w_ld8 * 16;
  int max_sw = 0;
  for (int i = 0; i < p.n_seg; ++i) max_sw = p.seg[i].SW > max_sw ? p.seg[i].SW : max_sw;
  return fast_on && p.div == 1 && p.kh * p.kw <= 31 && a_bytes < (1ll << 31) && w_bytes < (1ll << 31) &&
         (long long)max_sw * p.ld_src * 4 < (1ll << 23) && p.src_rows > 0;
}

// the tap-row-reuse kernel applies: 3-wide stride-1 "same" geometry on an f32 operand (conditions of igemm3x_kernel)
static bool igemm3x_ok(const IgemmParams& p, bool planes_in, bool planes_out, int w_rows, int w_ld8) {
  static const bool x_on = []() { const char* e = getenv("PP_CONV3_XREUSE"); return !(e && e[0] == '0'); }();
  bool same = x_on && igemm3_fast_ok(p, planes_in, w_rows, w_ld8) && !p.sc_on && p.kw == 3 && p.mul == 1 && p.div == 1 &&
              (p.src != nullptr || planes_in) && (!planes_in || p.ld_src % 8 == 0) && p.w_tstep == 1 && p.w_tx0 == 0;
  (void)planes_out;
  for (int i = 0; i < p.n_seg && same; ++i)
    same = p.seg[i].OH == p.seg[i].SH && p.seg[i].OW == p.seg[i].SW && p.seg[i].row_begin == p.seg[i].src_row_begin;
  return same;
}

// ---- sparse data gradient of a 3x3 stride-1 conv: which 32-row OUTPUT blocks can a non-zero of dy reach? ----
// Output row m reads dy rows m + dy * W + dx, dy, dx in {-1, 0, 1} (W = width of m's level): block b is live when a flagged
// dy block intersects one of the three 34-row windows [32 b - 1 + j W, 32 b + 32 + j W], j = -1, 0, 1 -- exact in 2-D up to
// the 32-cell granularity (image / level boundaries are ignored: conservative).
__global__ void rl_dilate_kernel(const IgemmParams p, const unsigned char* __restrict__ in_flags, unsigned char* __restrict__ out_flags,
                                 int n_blocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const int r_lo = 32 * b, r_hi = min(32 * b + 31, p.M - 1);
  int any = 0;
  for (int s = 0; s < p.n_seg; ++s) {
    const int seg_lo = p.seg[s].row_begin, seg_hi = (s + 1 < p.n_seg ? p.seg[s + 1].row_begin : p.M) - 1;
    if (r_hi < seg_lo || r_lo > seg_hi) continue;
    const int W = p.seg[s].OW;
    for (int j = -1; j <= 1; ++j) {
      int lo = r_lo - 1 + j * W, hi = r_hi + 1 + j * W;
      lo = lo < 0 ? 0 : lo;
      hi = hi > p.M - 1 ? p.M - 1 : hi;
      for (int q = lo >> 5; q <= (hi >> 5); ++q) any |= in_flags[q];
    }
  }
  out_flags[b] = any ? 1 : 0;
}

// rows of the blocks that no non-zero reaches: dx = mask?(addend or 0)
__global__ void rl_fill_kernel(const IgemmParams p, const unsigned char* __restrict__ live, const float* __restrict__ g_addend,
                               const float* __restrict__ g_mask, float* __restrict__ g_out, void* __restrict__ g_ohi, void* __restrict__ g_olo) {
  const int b = blockIdx.x;
  if (live[b]) return;
  const int n4 = (p.Nout + 3) >> 2;
  const int r0 = 32 * b, nr = min(32, p.M - r0);
  if (g_ohi && !g_out && !g_addend && p.Nout == p.ld_out && (p.ld_out & 7) == 0 && (p.add_hi == nullptr || (p.ld_add & 7) == 0) &&
      (p.mask_hi == nullptr || (p.ld_mask & 7) == 0)) {
    // planes in and out, whole rows: the block is nr * ld_out / 8 contiguous 32-byte groups -> 16-byte accesses throughout
    // (the 8-byte form below ran at half the store rate: 46 us per launch in the step)
    const int n8 = nr * (p.ld_out >> 3);
    uint4* dst = reinterpret_cast<uint4*>(g_ohi) + 2 * (long long)r0 * (p.ld_out >> 3);
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    for (int i = threadIdx.x; i < n8; i += blockDim.x) {
      uint4 oh = z4, ol = z4;
      if (p.add_hi) {
        const int r = i / (p.ld_out >> 3), g8 = i - r * (p.ld_out >> 3);
        const uint4* a = reinterpret_cast<const uint4*>(p.add_hi) + 2 * ((long long)(r0 + r) * (p.ld_add >> 3) + g8);
        oh = a[0];
        ol = a[1];
        if (p.mask_hi) {  // keep the addend where the ReLU source is positive (hi > 0: sign bit clear and not zero)
          const uint4 k = reinterpret_cast<const uint4*>(p.mask_hi)[2 * ((long long)(r0 + r) * (p.ld_mask >> 3) + g8)];
          const unsigned kk[4] = {k.x, k.y, k.z, k.w};
          unsigned hh[4] = {oh.x, oh.y, oh.z, oh.w}, ll[4] = {ol.x, ol.y, ol.z, ol.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned lo_ok = ((kk[j] & 0x8000u) == 0u && (kk[j] & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
            const unsigned hi_ok = ((kk[j] & 0x80000000u) == 0u && (kk[j] & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
            hh[j] &= (lo_ok | hi_ok);
            ll[j] &= (lo_ok | hi_ok);
          }
          oh = make_uint4(hh[0], hh[1], hh[2], hh[3]);
          ol = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        }
      }
      dst[2 * i] = oh;
      dst[2 * i + 1] = ol;
    }
    return;
  }
  for (int i = threadIdx.x; i < nr * n4; i += blockDim.x) {
    const int r = i / n4, co = 4 * (i - r * n4);
    const long long m = r0 + r;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_addend || p.add_hi) {
      finish4(p, z, m, co, true, g_addend, g_mask, g_out, g_ohi, g_olo);
    } else {  // (without an addend the row is zero whatever the mask says: nothing to read)
      if (g_out) *reinterpret_cast<float4*>(g_out + m * p.ld_out + co) = z;
      if (g_ohi) {
        reinterpret_cast<uint2*>(g_ohi)[pk4((m * p.ld_out + co) >> 2)] = make_uint2(0u, 0u);
        reinterpret_cast<uint2*>(g_olo)[pk4((m * p.ld_out + co) >> 2)] = make_uint2(0u, 0u);
      }
    }
  }
}

__global__ void row_block_compact_kernel(const unsigned char* __restrict__ flags, int n_blocks, int* __restrict__ list);

// the whole sparse bwd-data: dilate -> compact -> igemm3f over the listed blocks -> fill the others.  scratch: n_blocks bytes
// + (n_blocks + 1) ints behind the caller's flags / list (pp_row_block_list documents the sizes).
// ws / ws_bytes: the split-K scratch (may be NULL).  The listed tiles are few (a fifth of the rows on the bench targets: ~320
// workgroups of 144 k-steps each on 256 CUs, every one of them alone with its load latencies): with a scratch buffer the
// reduction is split two ways (PP_SPARSE_DGRAD_SPLITS) and rl_splitk_finish_kernel adds the slices of the listed blocks.
// dy_flags == NULL (forward, pp_ctx_set_row_block_out): out_flags are given -- no dilation, and the rows of the other blocks
// are left as they are (fill = false).
template <int TM, int TN>
static void launch_igemm3_rowlist(hipStream_t st, IgemmParams& p, const void* ahi, const void* alo, const void* whi, const void* wlo, int w_rows,
                                  int w_ld8, void* ohi, void* olo, const unsigned char* dy_flags, unsigned char* out_flags, int* out_list,
                                  float* ws, size_t ws_bytes, bool fill = true) {
  // (dy_flags also masks the gather: rows of the gathered tensor outside its flagged blocks are never fetched)
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int nb = (p.M + 31) / 32;
  static const int want_splits = []() { const char* e = getenv("PP_SPARSE_DGRAD_SPLITS"); return e ? atoi(e) : 2; }();
  const int n_steps = p.kh * p.kw * (p.Cred / 32);
  int splits = 1;
  if (ws && want_splits > 1 && (ohi || p.out)) {
    const long long slice_bytes = (long long)nb * 32 * p.ld_out * 4;
    splits = want_splits;
    while (splits > 1 && (n_steps / splits < 24 || slice_bytes * splits > (long long)ws_bytes)) --splits;
  }
  if (dy_flags) hipLaunchKernelGGL(rl_dilate_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, p, dy_flags, out_flags, nb);
  hipLaunchKernelGGL(row_block_compact_kernel, dim3(1), dim3(256), 0, st, (const unsigned char*)out_flags, nb, out_list);
  p.n_tiles_n = (p.Nout + BN - 1) / BN;
  const int n_tiles_m = (nb + BM / 32 - 1) / (BM / 32);
  const long long a_bytes = p.src_rows * (long long)p.ld_src * 4, w_bytes = (long long)p.w_taps * w_rows * w_ld8 * 16;
  const dim3 grid((unsigned)(n_tiles_m * p.n_tiles_n * splits));
  if (ahi && !ohi)  // planes in, float32 out (a head's last conv in the forward pass)
    hipLaunchKernelGGL((igemm3f_kernel<TM, TN, true, false, false, true>), grid, dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi, wlo,
                       (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr, w_rows, w_ld8, splits,
                       splits > 1 ? ws : (float*)nullptr, (const int*)out_list, dy_flags);
  else if (ahi)  // planes in, planes out
    hipLaunchKernelGGL((igemm3f_kernel<TM, TN, true, true, false, true>), grid, dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi, wlo,
                       (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo, w_rows, w_ld8, splits,
                       splits > 1 ? ws : (float*)nullptr, (const int*)out_list, dy_flags);
  else
    hipLaunchKernelGGL((igemm3f_kernel<TM, TN, false, false, false, true>), grid, dim3(256), 0, st, p, (const void*)p.src, nullptr,
                       (unsigned)a_bytes, whi, wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr,
                       w_rows, w_ld8, splits, splits > 1 ? ws : (float*)nullptr, (const int*)out_list, dy_flags);
  if (splits > 1)
    hipLaunchKernelGGL(rl_splitk_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, splits, (const float*)ws, (const int*)out_list, p.bias,
                       p.addend, p.mask_src, p.out, ohi, olo);
  // In place on the addend (dx == the running sum of the other data gradients of this tensor, no ReLU mask): the rows that no
  // non-zero reaches already hold their value -- nothing to fill (the shared pyramid features' gradient: 103 MB not moved).
  const bool in_place = (ohi != nullptr && p.add_hi == (const void*)ohi && p.ld_add == p.ld_out && !p.out && !p.addend && !p.mask_hi &&
                         !p.mask_src && !p.relu) ||
                        (p.out != nullptr && p.addend == p.out && p.ld_add == p.ld_out && !ohi && !p.add_hi && !p.mask_hi && !p.mask_src && !p.relu);
  if (fill && !in_place)
    hipLaunchKernelGGL(rl_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, (const unsigned char*)out_flags, p.addend, p.mask_src, p.out, ohi,
                       olo);
}

static void split_capture_pass(hipStream_t st, const IgemmParams& p, void* chi, void* clo) {
  const size_t n8 = (size_t)(p.src_rows * (long long)p.ld_src / 8);
  size_t blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n8, (const float4*)p.src, (uint4*)chi, (uint4*)clo, (const float*)nullptr);
}

// chi / clo (may be NULL): also write the bf16 split of the gathered f32 operand (pp_ctx_set_split_capture) -- inside the
// tap-row-reuse kernel where that one runs, by a separate pass over the operand otherwise
template <int TM, int TN>
static void launch_igemm3(hipStream_t st, IgemmParams& p, const void* ahi, const void* alo, const void* whi, const void* wlo, int w_rows,
                          int w_ld8, void* ohi, void* olo, int splits, float* ws, void* chi = nullptr, void* clo = nullptr,
                          const unsigned char* flags = nullptr, float* ws_any = nullptr, size_t ws_any_bytes = 0, int n_cu = 256) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.n_tiles_n = (p.Nout + BN - 1) / BN;
  const int n_tiles_m = (p.M + BM - 1) / BM;
  const dim3 grid((unsigned)(n_tiles_m * p.n_tiles_n * splits));
  const bool fast = igemm3_fast_ok(p, ahi != nullptr, w_rows, w_ld8);
  const long long a_bytes = p.src_rows * (long long)p.ld_src * 4;
  const long long w_bytes = (long long)p.w_taps * w_rows * w_ld8 * 16;
  if (p.sc_on) {  // parity-class launch of a stride-2 bwd-data (host guarantees: fast, no split, planes on both sides or on neither)
    if (ahi)
      hipLaunchKernelGGL((igemm3f_kernel<TM, TN, true, true, true>), grid, dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi, wlo,
                         (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo, w_rows, w_ld8, 1, nullptr);
    else
      hipLaunchKernelGGL((igemm3f_kernel<TM, TN, false, false, true>), grid, dim3(256), 0, st, p, (const void*)p.src, nullptr, (unsigned)a_bytes,
                         whi, wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr, w_rows, w_ld8, 1,
                         nullptr);
    return;
  }
  {
    // (the 256-row tile has the tap-row-reuse kernel for plane-stored operands only: its f32 form would need 32 more staging registers)
    if (igemm3x_ok(p, ahi != nullptr, ohi != nullptr, w_rows, w_ld8) && (TM <= 2 || (ahi != nullptr && !chi))) {
      const int n_tiles_mx = (p.M + BM - 3) / (BM - 2);  // tiles overlap by two rows
      int skip_halo = 0;
      for (int i = 0; i < p.n_seg; ++i) skip_halo = p.seg[i].SW + 1 > skip_halo ? p.seg[i].SW + 1 : skip_halo;
      const dim3 gridx((unsigned)(n_tiles_mx * p.n_tiles_n * splits));
      static const int x_order = []() { const char* e = getenv("PP_CONV3_X_ORDER"); return e ? atoi(e) : 1; }();
      p.x_ty_inner = x_order;
      // with split-K the partial sums go to the f32 scratch and splitk_finish_kernel writes the output (planes included)
      const bool op = ohi != nullptr && splits == 1;
      if constexpr (TM <= 2) {
        if (chi) {
          hipLaunchKernelGGL((igemm3x_kernel<TM, TN, true>), gridx, dim3(256), 0, st, p, (const void*)p.src, nullptr, (unsigned)a_bytes, whi, wlo,
                             (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr, w_rows, w_ld8, splits,
                             ws, chi, clo, flags, skip_halo);
          return;
        }
      }
      if constexpr (TM == 2 && TN == 2) {
        // the multi-stage LDS-DMA form of this launch (igemm4x_kernel: planes in and out, at least two rounds of 256 x 128 tiles);
        // PP_CONV3_DMA=0 keeps everything on igemm3x
        // (read at every launch, unlike the other knobs: tests compare the two kernels inside one process)
        const char* const dma_env = getenv("PP_CONV3_DMA");
        const bool dma_on = !(dma_env && dma_env[0] == '0');
        if (dma_on && ahi && op && !flags && splits == 1 && p.kh == 3 && x_order == 1 && w_rows % 128 == 0 && p.Cred % 32 == 0) {
          constexpr int BM4 = 256;
          const int n_tiles_m4 = (p.M + BM4 - 3) / (BM4 - 2);
          const int ntn = (p.Nout + 127) / 128;
          // (one workgroup per CU: a launch of less than two rounds is better off with three 128 x 128 workgroups per CU)
          // between one and two rounds (the 256-channel heads: 398 tiles) one launch CAN pay in isolation when the second round is
          // at least half full (PP_CONV3_DMA_FRAC=50: class head 149 us against 162; 1.19 rounds do not: mask head 137 / 115) -- but in
          // the training step those launches run beside the regression head's on the other lane, and two 115 KB workgroups cannot
          // share a CU where a 33 KB one fits next to this kernel's: the step LOSES 2.4 % (594 vs 608 images/s).  Off by default.
          static const int dma_min = []() { const char* e = getenv("PP_CONV3_DMA_MIN"); return e ? atoi(e) : 512; }();
          static const int dma_frac = []() { const char* e = getenv("PP_CONV3_DMA_FRAC"); return e ? atoi(e) : 1000; }();  // percent
          const int n_blk = n_tiles_m4 * ntn;
          const bool one_and_a_bit = n_blk > n_cu && n_blk < 2 * n_cu && (n_blk - n_cu) * 100 >= dma_frac * n_cu;
          if (n_blk >= dma_min || one_and_a_bit) {
            // Whole rounds of 256 x 128 tiles go to igemm4x; with one workgroup per CU a last, partly filled round would cost a
            // full round's time, so the rows of that round are a second launch: 128 x 128 tiles of igemm3x with the reduction
            // split over enough workgroups to fill the chip once (partial sums -> scratch, splitk_finish_kernel writes the rows)
            static const bool tail_on = []() { const char* e = getenv("PP_CONV3_DMA_TAIL"); return !(e && e[0] == '0'); }();
            int full_rt = n_tiles_m4;
            const int rounds = (n_tiles_m4 * ntn) / n_cu;
            // (a last round that is mostly full stays in the one launch: 720 x 540 gives 1 016 tiles = 3.97 rounds)
            static const int tail_frac = []() { const char* e = getenv("PP_CONV3_DMA_TAIL_FRAC"); return e ? atoi(e) : 60; }();  // percent
            const int last = (n_tiles_m4 * ntn) % n_cu;
            if (tail_on && last != 0 && last * 100 < tail_frac * n_cu && rounds >= 2 && ws_any != nullptr) full_rt = rounds * n_cu / ntn;
            const int m_split = full_rt * (BM4 - 2);
            int tail_splits = 0;
            if (full_rt < n_tiles_m4) {
              const int rem = p.M - m_split;
              const int n_rt = (rem + BM - 3) / (BM - 2);
              const int groups = p.kh * (p.Cred / 32);
              int sp = (3 * n_cu + n_rt * ntn - 1) / (n_rt * ntn);  // ~ one full round of three workgroups per CU
              while (sp > 1 && (groups / sp < 4 || (size_t)sp * rem * p.ld_out * 4 > ws_any_bytes)) --sp;
              if (sp > 1) tail_splits = sp;
              else full_rt = n_tiles_m4;  // (no room to split: everything in the one launch)
            }
            p.n_tiles_n = ntn;
            // scheduling variants (measured on the regression-head launch, P16): 1 = no iglp_opt hint in the k-step (default: 442 us
            // against 465 with igemm3x's hint, 0), 2 / 3 = s_setprio around the MFMAs of 0 / 1 (466 / 461), 4 = igemm3f's sched_barrier
            const char* const e_var = getenv("PP_CONV3_DMA_VAR");  // (per launch: A/B in one process)
            const int var = e_var ? atoi(e_var) : 1;
            auto go4 = [&](auto v) {
              hipLaunchKernelGGL((igemm4x_kernel<true, decltype(v)::value>), dim3((unsigned)(full_rt * ntn)), dim3(512), 0, st, p, ahi, alo,
                                 (unsigned)a_bytes, whi, wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo, w_rows,
                                 w_ld8);
            };
            if (var == 0) go4(std::integral_constant<int, 0>{});
            else if (var == 2) go4(std::integral_constant<int, 2>{});
            else if (var == 3) go4(std::integral_constant<int, 3>{});
            else if (var == 4) go4(std::integral_constant<int, 4>{});
            else if (var == 9) go4(std::integral_constant<int, 9>{});
            else go4(std::integral_constant<int, 1>{});
            if (tail_splits > 1) {
              IgemmParams q = p;
              q.m_off = full_rt * (BM4 - 2);
              const int rem = q.M - q.m_off;
              const int n_rt = (rem + BM - 3) / (BM - 2);
              hipLaunchKernelGGL((igemm3x_kernel<TM, TN, false, true, false>), dim3((unsigned)(n_rt * ntn * tail_splits)), dim3(256), 0, st, q, ahi,
                                 alo, (unsigned)a_bytes, whi, wlo, (unsigned)w_bytes, q.bias, q.addend, q.mask_src, q.out, (uint2*)nullptr,
                                 (uint2*)nullptr, w_rows, w_ld8, tail_splits, ws_any, nullptr, nullptr, (const unsigned char*)nullptr, skip_halo);
              const long long total = (long long)rem * ((q.Nout + 3) >> 2);
              long long blocks = (total + 255) / 256;
              if (blocks > (long long)n_cu * 8) blocks = (long long)n_cu * 8;
              hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q, tail_splits, (const float*)ws_any, q.bias, q.addend,
                                 q.mask_src, q.out, ohi, olo);
            }
            return;
          }
          // Round 4: launches too small for two rounds of 256-row tiles (the 256-channel heads, the FPN 3x3: 1.2-1.6 rounds of 128-row
          // tiles on 2 x 256 slots) CAN take the 128 x 128 form of the same pipeline, two workgroups per CU (PP_CONV3_DMA2=1: opt-in;
          // PP_CONV3_DMA2_MIN: fewest 128-row tiles, default one round)
          const char* const e_dma2 = getenv("PP_CONV3_DMA2");  // (per launch, like PP_CONV3_DMA)
          // (off by default: +8 % on the class-head launch, -8 % on the mask head / FPN launches -- 610 tiles on 512 slots --, -2 % on
          // the training step, profiles/r04_lds_dma_128_row_tiles.txt)
          const bool dma2_on = e_dma2 && e_dma2[0] == '1';
          static const int dma2_min = []() { const char* e = getenv("PP_CONV3_DMA2_MIN"); return e ? atoi(e) : 512; }();
          if (dma2_on && n_tiles_mx * ntn >= dma2_min) {
            p.n_tiles_n = ntn;
            hipLaunchKernelGGL((igemm4x_kernel<true, 1, 2>), dim3((unsigned)(n_tiles_mx * ntn)), dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi,
                               wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo, w_rows, w_ld8);
            return;
          }
        }
      }
      if (ahi && op)
        hipLaunchKernelGGL((igemm3x_kernel<TM, TN, false, true, true>), gridx, dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi, wlo,
                           (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo, w_rows, w_ld8, splits, ws, nullptr,
                           nullptr, flags, skip_halo);
      else if (ahi)
        hipLaunchKernelGGL((igemm3x_kernel<TM, TN, false, true, false>), gridx, dim3(256), 0, st, p, ahi, alo, (unsigned)a_bytes, whi, wlo,
                           (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr, w_rows, w_ld8, splits, ws,
                           nullptr, nullptr, flags, skip_halo);
      else if constexpr (TM <= 2) {
        if (op)
          hipLaunchKernelGGL((igemm3x_kernel<TM, TN, false, false, true>), gridx, dim3(256), 0, st, p, (const void*)p.src, nullptr,
                             (unsigned)a_bytes, whi, wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo,
                             w_rows, w_ld8, splits, ws, nullptr, nullptr, flags, skip_halo);
        else
          hipLaunchKernelGGL((igemm3x_kernel<TM, TN, false>), gridx, dim3(256), 0, st, p, (const void*)p.src, nullptr, (unsigned)a_bytes, whi,
                             wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)nullptr, (uint2*)nullptr, w_rows, w_ld8, splits,
                             ws, nullptr, nullptr, flags, skip_halo);
      }
      return;
    }
  }
  if (chi) split_capture_pass(st, p, chi, clo);
  if (splits > 1) ohi = olo = nullptr;  // partial sums -> scratch; splitk_finish_kernel writes the planes
  auto go = [&](auto ap, auto op) {
    constexpr bool AP = decltype(ap)::value, OP = decltype(op)::value;
    if (fast)
      hipLaunchKernelGGL((igemm3f_kernel<TM, TN, AP, OP>), grid, dim3(256), 0, st, p, AP ? ahi : (const void*)p.src, AP ? alo : nullptr,
                         (unsigned)a_bytes, whi, wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)ohi, (uint2*)olo,
                         w_rows, w_ld8, splits, ws);
    else
      hipLaunchKernelGGL((igemm3_kernel<TM, TN, AP, OP>), grid, dim3(256), 0, st, p, p.src, (const uint4*)(AP ? ahi : nullptr),
                         (const uint4*)(AP ? alo : nullptr), (const uint4*)whi, (const uint4*)wlo, p.bias, p.addend, p.mask_src, p.out,
                         (uint2*)ohi, (uint2*)olo, w_rows, w_ld8);
  };
  if (ahi && ohi) go(std::true_type{}, std::true_type{});
  else if (ahi) go(std::true_type{}, std::false_type{});
  else if (ohi) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
}

static void pick_tile3(const pp_ctx* ctx, int M, int Nout, int ld_out, int n_steps, bool may_split, bool x_ok, int* tm, int* tn, int* splits) {
  // measured in-flight rates relative to 128x128 (tools/conv_bench.py): the LDS store path (ds_write_b128 of the
  // staged tiles) costs ~30 % of the loop, so the tile with the fewest staged bytes per MFMA wins when it fills the chip.
  // A workgroup costs (its k-steps + ~8 steps' worth of prologue, epilogue and launch ramp) x tile area; splitting the
  // reduction S ways multiplies the workgroups and pays S x the output in f32 atomics plus the finishing pass.
  static const int cand[5][2] = {{4, 2}, {2, 2}, {1, 2}, {2, 1}, {1, 1}};
  static const double eff[5] = {1.1, 1.0, 0.9, 0.9, 0.8};
  static const int slots[5] = {2, 3, 4, 4, 4};
  static const int split_cand[8] = {1, 2, 3, 4, 6, 8, 12, 16};
  const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const long long ws_bytes = (long long)ctx->ws_bytes;
  double best = 1e300;
  *splits = 1;
  for (int i = 0; i < 5; ++i) {
    const int bm = 64 * cand[i][0], bn = 64 * cand[i][1];
    const long long blocks = (long long)((M + bm - 1) / bm) * ((Nout + bn - 1) / bn);
    for (int si = 0; si < (may_split ? 8 : 1); ++si) {
      const int sp = split_cand[si];
      if (sp > 1 && (n_steps / sp < 8 || (long long)sp * M * ld_out * 4 > ws_bytes)) break;
      const double steps = (double)((n_steps + sp - 1) / sp) + 8.0;
      double t = est_rounds(blocks * sp, slots[i], cus) * steps * cand[i][0] * cand[i][1] / eff[i];
      if (sp > 1) {
        // slices: sp x M x N x 4 B written and read back (~4 TB/s), the epilogue operands, one more launch; one k-step
        // of a 128x128 tile ~ 2.1 us when the chip is full  ->  convert microseconds to the same step units
        const double us = (double)M * Nout * (8.0 * sp + 12.0) / 4.0e6 + 4.0;
        t += us / 2.1 * 4.0;
      }
      if (t < best * 0.97) {
        best = t;
        *tm = cand[i][0];
        *tn = cand[i][1];
        *splits = sp;
      }
    }
  }
  // the 256x128 tile has no tap-row-reuse variant: where that kernel applies, 128x128 with reuse wins (measured on the
  // 256-channel class head, 50400 rows: 194-207 us against 205-227 us; a reuse bonus inside the model above instead sent
  // that shape to 64x128 and the step lost 2 %)
  static const bool x4 = []() { const char* e = getenv("PP_CONV3_X4"); return e && e[0] == '1'; }();
  if (x_ok && *tm == 4 && *splits == 1 && !x4) {
    *tm = 2;
    *tn = 2;
  }
  const char* e = getenv("PP_CONV3_TILE");
  if (e && e[0] && e[1] == ',' && e[2]) {
    *tm = e[0] - '0';
    *tn = e[2] - '0';
  }
  if (const char* es = getenv("PP_CONV3_SPLITS")) {
    const int v = atoi(es);
    if (v >= 1 && (v == 1 || (may_split && n_steps / v >= 1 && (long long)v * M * ld_out * 4 <= ws_bytes))) *splits = v;
  }
}

static void dispatch3(pp_ctx* ctx, IgemmParams& p, const void* ahi, const void* alo, const void* whi, const void* wlo, int w_rows,
                      int w_ld8, void* ohi, void* olo, void* chi = nullptr, void* clo = nullptr, const unsigned char* flags = nullptr) {
  int tm, tn, splits;
  const int n_steps = p.kh * p.kw * (p.Cred / 32);
  {
    // Round 4: 1x1 convolutions on plane-stored operands (the bottleneck branches of the backbone, the FPN laterals; forward and
    // stride-1 data gradient) -> the persistent LDS-DMA GEMM of conv4.hip.  OPT-IN (PP_CONV4P=1): bit-identical, but 1.05-1.9x
    // SLOWER than igemm3f per launch and -7 % on the training step (profiles/r04_1x1_persistent_dma_gemm.txt);
    // PP_CONV4P_NST: stages of its ring (4 = all of the CU's LDS, default; 3); PP_CONV4P_SPLITS forces the reduction split.
    // (read at every launch: tests compare the two kernels inside one process)
    const char* const e_on = getenv("PP_CONV4P");
    const char* const e_nst = getenv("PP_CONV4P_NST");
    const char* const e_sp = getenv("PP_CONV4P_SPLITS");
    const bool p_on = e_on && e_on[0] == '1';  // (off by default: measured slower than igemm3f on every backbone shape, see conv4.hip)
    const int p_nst = e_nst && atoi(e_nst) == 3 ? 3 : 4;
    const int p_splits = e_sp ? atoi(e_sp) : 0;
    const long long a_bytes = p.src_rows * (long long)p.ld_src * 4, w_bytes = (long long)p.w_taps * w_rows * w_ld8 * 16;
    if (p_on && p.kh == 1 && p.kw == 1 && ahi && !chi && !flags && !p.sc_on && p.div == 1 && p.Cred % 32 == 0 && p.ld_src % 8 == 0 &&
        a_bytes < (1ll << 31) && w_bytes < (1ll << 31) && p.src_rows > 0 && (ohi || p.out)) {
      const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
      const long long tiles = (long long)((p.M + 127) / 128) * ((p.Nout + 127) / 128);
      int sp = 1;
      if (tiles * 4 < (long long)n_cu * 3 && ctx->ws != nullptr) {  // under three quarters of a round: split the reduction to fill the chip
        sp = (int)(n_cu / tiles);
        while (sp > 1 && (n_steps / sp < 4 || (long long)sp * p.M * p.ld_out * 4 > (long long)ctx->ws_bytes)) --sp;
      }
      if (p_splits >= 1 && (p_splits == 1 || (ctx->ws != nullptr && n_steps / p_splits >= 1 && (long long)p_splits * p.M * p.ld_out * 4 <= (long long)ctx->ws_bytes)))
        sp = p_splits;
      if (getenv("PP_CONV_DEBUG"))
        fprintf(stderr, "igemm4p M %d N %d steps %d -> %lld tiles x %d splits on %d CUs, %d stages\n", p.M, p.Nout, n_steps, tiles, sp, n_cu, p_nst);
      PP_API(pp4_launch_igemm4p)(ctx->stream, p, ahi, whi, wlo, w_rows, w_ld8, ohi, olo, sp, sp > 1 ? ctx->ws : nullptr, n_cu, p_nst);
      if (sp > 1) {
        const long long total = (long long)p.M * ((p.Nout + 3) >> 2);
        long long blocks = (total + 255) / 256;
        if (blocks > (long long)n_cu * 8) blocks = (long long)n_cu * 8;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, sp, ctx->ws, p.bias, p.addend, p.mask_src, p.out,
                           ohi, olo);
      }
      return;
    }
  }
  const bool may_split = ctx->ws != nullptr && (ohi || p.out != nullptr) && !p.sc_on && igemm3_fast_ok(p, ahi != nullptr, w_rows, w_ld8);
  pick_tile3(ctx, p.M, p.Nout, p.ld_out, n_steps, may_split, igemm3x_ok(p, ahi != nullptr, ohi != nullptr, w_rows, w_ld8), &tm, &tn, &splits);
  if (getenv("PP_CONV_DEBUG"))
    fprintf(stderr, "igemm3 M %d N %d steps %d -> tile %dx%d splits %d (may_split %d: ws %d planes_out %d out %d)\n", p.M, p.Nout, n_steps, 64 * tm,
            64 * tn, splits, (int)may_split, (int)(ctx->ws != nullptr), (int)(ohi != nullptr), (int)(p.out != nullptr));
  float* ws = splits > 1 ? ctx->ws : nullptr;
  if (tm == 4) launch_igemm3<4, 2>(ctx->stream, p, ahi, alo, whi, wlo, w_rows, w_ld8, ohi, olo, splits, ws, chi, clo, flags);
  else if (tm == 2 && tn == 2)
    launch_igemm3<2, 2>(ctx->stream, p, ahi, alo, whi, wlo, w_rows, w_ld8, ohi, olo, splits, ws, chi, clo, flags, ctx->ws, (size_t)ctx->ws_bytes,
                        ctx->n_cu > 0 ? ctx->n_cu : 256);
  else if (tm == 1 && tn == 2) launch_igemm3<1, 2>(ctx->stream, p, ahi, alo, whi, wlo, w_rows, w_ld8, ohi, olo, splits, ws, chi, clo, flags);
  else if (tm == 2 && tn == 1) launch_igemm3<2, 1>(ctx->stream, p, ahi, alo, whi, wlo, w_rows, w_ld8, ohi, olo, splits, ws, chi, clo, flags);
  else launch_igemm3<1, 1>(ctx->stream, p, ahi, alo, whi, wlo, w_rows, w_ld8, ohi, olo, splits, ws, chi, clo, flags);
  if (splits > 1) {
    const long long total = (long long)p.M * ((p.Nout + 3) >> 2);
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, splits, ctx->ws, p.bias, p.addend, p.mask_src,
                       p.out, ohi, olo);
  }
}

}  // namespace
extern "C" int PP_API(pp_conv2d_nhwc_fwd_bf16x3)(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const void* x_hi, const void* x_lo,
                                         const void* w_hi, const void* w_lo, const float* bias, const float* residual, int ld_res,
                                         int relu, float* y, void* y_hi, void* y_lo) {
  PP_REQUIRE_CTX(ctx);
  void *chi = ctx->cap_hi, *clo = ctx->cap_lo;  // one-shot (pp_ctx_set_split_capture)
  ctx->cap_hi = ctx->cap_lo = nullptr;
  const void *ep_ah = ctx->ep_add_hi, *ep_al = ctx->ep_add_lo;  // one-shot (pp_ctx_set_epilogue_planes): the residual as planes
  ctx->ep_add_hi = ctx->ep_add_lo = ctx->ep_mask_hi = nullptr;
  const unsigned char* out_flags = ctx->out_flags;  // one-shot (pp_ctx_set_row_block_out)
  int* out_list = ctx->out_list;
  ctx->out_flags = nullptr;
  ctx->out_list = nullptr;
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_fwd_bf16x3");
  PP_CHECK_ARG(ctx, !(ep_ah && residual), PP_ERR_ARG, "pp_conv2d_nhwc_fwd_bf16x3: residual given both as f32 and as planes");
  PP_CHECK_ARG(ctx, !ep_ah || (ld_res % 4 == 0 && ld_res >= ((d->cout + 3) & ~3)), PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_bf16x3: residual planes");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, !chi || (x && !x_hi && pp_is_packed(chi, clo)), PP_ERR_ARG, "pp_conv2d_nhwc_fwd_bf16x3: split capture needs the f32 operand");
  PP_CHECK_ARG(ctx, (x || (x_hi && x_lo)) && w_hi && w_lo && (y || (y_hi && y_lo)), PP_ERR_ARG, "pp_conv2d_nhwc_fwd_bf16x3: null tensor");
  PP_CHECK_ARG(ctx, (x_hi == nullptr) == (x_lo == nullptr), PP_ERR_ARG, "pp_conv2d_nhwc_fwd_bf16x3: x_hi and x_lo go together");
  PP_CHECK_ARG(ctx, d->cin % 32 == 0 && d->ld_x % 8 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_bf16x3: cin %d must be a multiple of 32, ld_x of 8", d->cin);
  PP_CHECK_ARG(ctx, pp_is_packed(x_hi, x_lo) && pp_is_packed(y_hi, y_lo) && pp_is_packed(ep_ah, ep_al), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_fwd_bf16x3: planes must be packed (lo = hi + 16 bytes, 16-byte aligned)");
  PP_CHECK_ARG(ctx, !y_hi || (d->cout % 8 == 0 && d->ld_y % 8 == 0), PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_bf16x3: output planes need cout, ld_y %% 8 == 0");
  PP_CHECK_ARG(ctx, !ep_ah || (ld_res % 8 == 0 && d->cout % 8 == 0), PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_bf16x3: residual planes need cout, ld_res %% 8 == 0");
  PP_CHECK_ARG(ctx, (!x || pp_is_aligned16(x)) && pp_is_aligned16(w_hi) && pp_is_aligned16(w_lo) && (!y || pp_is_aligned16(y)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_fwd_bf16x3: tensors must be 16-byte aligned");
  PP_CHECK_ARG(ctx, !residual || (ld_res % 4 == 0 && ld_res >= ((d->cout + 3) & ~3) && pp_is_aligned16(residual)), PP_ERR_SHAPE,
               "pp_conv2d_nhwc_fwd_bf16x3: residual");
  PP_CHECK_ARG(ctx, !bias || pp_is_aligned16(bias), PP_ERR_ALIGN, "pp_conv2d_nhwc_fwd_bf16x3: bias alignment");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = x; p.out = y; p.bias = bias; p.addend = residual; p.mask_src = nullptr;
  p.add_hi = ep_ah; p.add_lo = ep_al;
  p.ld_src = d->ld_x; p.ld_w = d->ld_w; p.ld_out = d->ld_y; p.ld_add = ld_res; p.ld_mask = 0;
  p.relu = relu;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M, &p.src_rows);
  p.Cred = d->cin; p.Nout = d->cout; p.w_tap_rows = d->cin;
  p.kh = d->kh; p.kw = d->kw;
  p.mul = d->stride; p.tsign = 1; p.off_y = -d->pad_t; p.off_x = -d->pad_l; p.div = 1;
  p.w_ty0 = 0; p.w_tx0 = 0; p.w_tstep = 1; p.w_kw = d->kw; p.w_taps = d->kh * d->kw;
  PP_CHECK_ARG(ctx, (y_hi == nullptr) == (y_lo == nullptr) && (!y_hi || (d->ld_y % 4 == 0 && pp_is_aligned16(y_hi) && pp_is_aligned16(y_lo))),
               PP_ERR_ARG, "pp_conv2d_nhwc_fwd_bf16x3: output planes");
  if (out_flags) {
    // only the flagged 32-row output blocks (the listed-block launch of the sparse data gradient, without dilation and fill):
    // plane-stored input, 3x3 stride 1 pad 1 on an unchanged grid, no residual
    bool ok = x_hi != nullptr && ((y != nullptr) != (y_hi != nullptr)) && !chi && !residual && !ep_ah && d->stride == 1 && d->kh == 3 && d->kw == 3 && d->pad_t == 1 &&
              d->pad_l == 1 && igemm3_fast_ok(p, true, d->cout, d->cin / 8);
    for (int i = 0; i < p.n_seg && ok; ++i)
      ok = p.seg[i].OH == p.seg[i].SH && p.seg[i].OW == p.seg[i].SW && p.seg[i].row_begin == p.seg[i].src_row_begin;
    PP_CHECK_ARG(ctx, ok, PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_bf16x3: the row-block-out hint needs a 3x3 stride-1 'same' conv on plane-stored input");
    launch_igemm3_rowlist<2, 2>(ctx->stream, p, x_hi, x_lo, w_hi, w_lo, d->cout, d->cin / 8, y_hi, y_lo, nullptr, const_cast<unsigned char*>(out_flags),
                                out_list, ctx->ws, ctx->ws_bytes, false);
    PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_fwd_bf16x3");
    return PP_OK;
  }
  dispatch3(ctx, p, x_hi, x_lo, w_hi, w_lo, d->cout, d->cin / 8, y_hi, y_lo, chi, clo);
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_fwd_bf16x3");
  return PP_OK;
}
namespace {

// out_flags[b] = 1 when a flagged block of in_flags lies within one pixel (2-D, 32-row granularity) of block b of the row
// space of a 3x3 stride-1 'same' conv d (the blocks of its input that the flagged blocks of its output read, and vice versa)
}  // namespace
extern "C" int PP_API(pp_row_block_dilate)(pp_ctx* ctx, const pp_conv_desc* d, const unsigned char* in_flags, unsigned char* out_flags) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_row_block_dilate");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, in_flags && out_flags && in_flags != out_flags, PP_ERR_ARG, "pp_row_block_dilate: two distinct flag buffers");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M, &p.src_rows);
  bool same = d->stride == 1 && d->kh == 3 && d->kw == 3;
  for (int i = 0; i < p.n_seg && same; ++i) same = p.seg[i].OH == p.seg[i].SH && p.seg[i].OW == p.seg[i].SW && p.seg[i].row_begin == p.seg[i].src_row_begin;
  PP_CHECK_ARG(ctx, same, PP_ERR_SHAPE, "pp_row_block_dilate: 3x3 stride-1 conv on an unchanged grid");
  const int nb = (p.M + 31) / 32;
  hipLaunchKernelGGL(rl_dilate_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, p, in_flags, out_flags, nb);
  PP_CHECK_LAUNCH(ctx, "pp_row_block_dilate");
  return PP_OK;
}
namespace {

// The 7x7 stride-2 RGB stem on the bf16 path.  Input: the packed 4-channel image inside a zero frame [n_img][Hp][Wp][4] with
// the image at (3, 3) (pp_pack_rgb_to_4_padded / pp_preprocess_caffe_u8_padded), Hp >= H + 6, Wp >= W + 8, Wp even.  Each
// kernel ROW (7 taps x 4 channels = 28 contiguous floats of the frame, padded to 32) is one tap of a 7x1 convolution over
// 32 "channels" that overlap between neighbouring pixels -- the gather of igemm3f needs nothing else: base offset of pixel
// (2 oy + ty, 2 ox), 32 consecutive floats.  Weights: planes [7][cout][32] with plane row ty, column tx * 4 + c (zeros for
// c == 3 and tx == 7), see Engine._build_stem.  224 of the executed 7 x 32 reduction steps are 147 algorithmic.
}  // namespace
extern "C" int PP_API(pp_stem7x7s2_fwd_bf16x3)(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, const float* x4p, const void* w_hi,
                                       const void* w_lo, int cout, const float* bias, int relu, float* y, int ld_y) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x4p && w_hi && w_lo && y && pp_is_aligned16(x4p) && pp_is_aligned16(y) && pp_is_aligned16(w_hi) && pp_is_aligned16(w_lo),
               PP_ERR_ARG, "pp_stem7x7s2_fwd_bf16x3: null / unaligned tensor");
  PP_CHECK_ARG(ctx, n_img > 0 && H > 0 && W > 0 && Hp >= H + 6 && Wp >= W + 8 && Wp % 2 == 0 && cout > 0 && ld_y % 4 == 0 && ld_y >= ((cout + 3) & ~3),
               PP_ERR_SHAPE, "pp_stem7x7s2_fwd_bf16x3: bad geometry");
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = x4p; p.out = y; p.bias = bias;
  p.ld_src = 4; p.ld_out = ld_y;
  p.relu = relu;
  p.n_seg = 1;
  p.seg[0].row_begin = 0; p.seg[0].src_row_begin = 0;
  p.seg[0].OH = OH; p.seg[0].OW = OW; p.seg[0].SH = Hp; p.seg[0].SW = Wp;
  p.M = n_img * OH * OW;
  p.src_rows = (long long)n_img * Hp * Wp;
  p.Cred = 32; p.Nout = cout; p.w_tap_rows = 32;
  p.kh = 7; p.kw = 1;
  p.mul = 2; p.tsign = 1; p.off_y = 0; p.off_x = 0; p.div = 1;
  p.w_ty0 = 0; p.w_tx0 = 0; p.w_tstep = 1; p.w_kw = 1; p.w_taps = 7;
  PP_CHECK_ARG(ctx, igemm3_fast_ok(p, false, cout, 4), PP_ERR_SHAPE, "pp_stem7x7s2_fwd_bf16x3: frame too large for 31-bit offsets");
  dispatch3(ctx, p, nullptr, nullptr, w_hi, w_lo, cout, 4, nullptr, nullptr);
  PP_CHECK_LAUNCH(ctx, "pp_stem7x7s2_fwd_bf16x3");
  return PP_OK;
}
namespace {

}  // namespace
extern "C" int PP_API(pp_conv2d_nhwc_bwd_data_bf16x3)(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const void* dy_hi, const void* dy_lo,
                                              const void* w_hi, const void* w_lo, const float* addend, int ld_add,
                                              const float* relu_src, int ld_rs, float* dx, void* dx_hi, void* dx_lo) {
  PP_REQUIRE_CTX(ctx);
  void *chi = ctx->cap_hi, *clo = ctx->cap_lo;  // one-shot (pp_ctx_set_split_capture)
  ctx->cap_hi = ctx->cap_lo = nullptr;
  const unsigned char* skip_flags = ctx->skip_flags;  // one-shot (pp_ctx_set_row_block_skip)
  const int* skip_list_in = ctx->skip_list;
  const bool skip_scratch_ok = skip_flags != nullptr && skip_list_in != nullptr;
  ctx->skip_flags = nullptr;
  ctx->skip_list = nullptr;
  const bool lazy_out = ctx->lazy_out != 0, lazy_in = ctx->lazy_in != 0;  // one-shot (pp_ctx_set_row_block_lazy)
  ctx->lazy_out = ctx->lazy_in = 0;
  const void *ep_ah = ctx->ep_add_hi, *ep_al = ctx->ep_add_lo, *ep_mh = ctx->ep_mask_hi;  // one-shot (pp_ctx_set_epilogue_planes)
  ctx->ep_add_hi = ctx->ep_add_lo = ctx->ep_mask_hi = nullptr;
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_bwd_data_bf16x3");
  PP_CHECK_ARG(ctx, !(ep_ah && addend) && !(ep_mh && relu_src), PP_ERR_ARG,
               "pp_conv2d_nhwc_bwd_data_bf16x3: addend / relu_src given both as f32 and as planes");
  PP_CHECK_ARG(ctx, (!ep_ah || (ld_add >= d->cin && ld_add % 4 == 0)) && (!ep_mh || (ld_rs >= d->cin && ld_rs % 4 == 0)), PP_ERR_SHAPE,
               "pp_conv2d_nhwc_bwd_data_bf16x3: addend / relu_src planes");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, !chi || (dy && !dy_hi && pp_is_packed(chi, clo)), PP_ERR_ARG, "pp_conv2d_nhwc_bwd_data_bf16x3: split capture needs the f32 operand");
  PP_CHECK_ARG(ctx, (dy || (dy_hi && dy_lo)) && w_hi && w_lo && (dx || (dx_hi && dx_lo)), PP_ERR_ARG, "pp_conv2d_nhwc_bwd_data_bf16x3: null tensor");
  PP_CHECK_ARG(ctx, (dy_hi == nullptr) == (dy_lo == nullptr) && d->ld_y % 8 == 0, PP_ERR_ARG, "pp_conv2d_nhwc_bwd_data_bf16x3: planes / ld_y");
  PP_CHECK_ARG(ctx, pp_is_packed(dy_hi, dy_lo) && pp_is_packed(dx_hi, dx_lo) && pp_is_packed(ep_ah, ep_al) && pp_is_aligned16(ep_mh), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_data_bf16x3: planes must be packed (lo = hi + 16 bytes, 16-byte aligned)");
  PP_CHECK_ARG(ctx, (!dx_hi || (d->cin % 8 == 0 && d->ld_x % 8 == 0)) && (!ep_ah || ld_add % 8 == 0) && (!ep_mh || ld_rs % 8 == 0), PP_ERR_SHAPE,
               "pp_conv2d_nhwc_bwd_data_bf16x3: planes need channel counts and leading dimensions %% 8 == 0");
  const int cred = (d->cout + 31) / 32 * 32;
  PP_CHECK_ARG(ctx, d->cin % 16 == 0 && d->ld_y >= cred && d->ld_y % 4 == 0, PP_ERR_SHAPE,
               "pp_conv2d_nhwc_bwd_data_bf16x3: dy needs ld_y >= %d (cout rounded up to 32, zero padded)", cred);
  PP_CHECK_ARG(ctx, (!dy || pp_is_aligned16(dy)) && pp_is_aligned16(w_hi) && pp_is_aligned16(w_lo) && (!dx || pp_is_aligned16(dx)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_data_bf16x3: tensors must be 16-byte aligned");
  PP_CHECK_ARG(ctx, (!addend || (ld_add >= d->cin && ld_add % 4 == 0 && pp_is_aligned16(addend))) &&
                        (!relu_src || (ld_rs >= d->cin && ld_rs % 4 == 0 && pp_is_aligned16(relu_src))),
               PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_data_bf16x3: addend / relu_src");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = dy; p.out = dx; p.bias = nullptr; p.addend = addend; p.mask_src = relu_src;
  p.add_hi = ep_ah; p.add_lo = ep_al; p.mask_hi = ep_mh;
  p.ld_src = d->ld_y; p.ld_w = d->ld_w; p.ld_out = d->ld_x; p.ld_add = ld_add; p.ld_mask = ld_rs;
  p.relu = 0;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, false, p.seg, &p.M, &p.src_rows);
  p.Cred = cred; p.Nout = d->cin; p.w_tap_rows = d->cin;
  p.kh = d->kh; p.kw = d->kw;
  p.mul = 1; p.tsign = -1; p.off_y = d->pad_t; p.off_x = d->pad_l; p.div = d->stride;
  p.w_ty0 = 0; p.w_tx0 = 0; p.w_tstep = 1; p.w_kw = d->kw; p.w_taps = d->kh * d->kw;
  PP_CHECK_ARG(ctx, (dx_hi == nullptr) == (dx_lo == nullptr) && (!dx_hi || (d->ld_x % 4 == 0 && pp_is_aligned16(dx_hi) && pp_is_aligned16(dx_lo))),
               PP_ERR_ARG, "pp_conv2d_nhwc_bwd_data_bf16x3: output planes");
  static const bool s2_classes = []() { const char* e = getenv("PP_CONV3_S2CLASSES"); return !(e && e[0] == '0'); }();
  const bool all_f32 = dy && dx && !dy_hi && !dx_hi, all_planes = dy_hi && dx_hi && !dx;  // (the parity-class / row-list kernels exist for these two)
  // Row-block skip: does the launch over the LISTED blocks (launch_igemm3_rowlist below) apply?  PP_SPARSE_DGRAD: 22 (default)
  // / 12 = tile of that launch (equal within noise on the bench step), 0 = keep the dense grid and skip the k-loop of tiles
  // that see no flagged block (coarser: a 126-row tile spans 1.6 image rows)
  static const int rl_mode = []() { const char* e = getenv("PP_SPARSE_DGRAD"); return e ? atoi(e) : 22; }();
  bool rl_ok = skip_flags && skip_scratch_ok && rl_mode && (all_f32 || all_planes) && !chi && d->stride == 1 && d->kh == 3 && d->kw == 3 &&
               d->pad_t == 1 && d->pad_l == 1 && p.bias == nullptr && igemm3_fast_ok(p, all_planes, d->cin, cred / 8);
  for (int i = 0; i < p.n_seg && rl_ok; ++i)
    rl_ok = p.seg[i].OH == p.seg[i].SH && p.seg[i].OW == p.seg[i].SW && p.seg[i].row_begin == p.seg[i].src_row_begin;
  // Contract of the hint's scratch (pp_row_block_list_planes_within relies on it): after this call the second n_blocks bytes
  // of the flags buffer flag every 32-row block of dx that may hold a non-zero -- the dilated list when the listed launch
  // runs, everything otherwise.
  PP_CHECK_ARG(ctx, !lazy_in || rl_ok, PP_ERR_ARG,
               "pp_conv2d_nhwc_bwd_data_bf16x3: dy was declared lazy (unwritten outside its flagged blocks) but the listed-block launch "
               "does not apply to this call");
  if (skip_scratch_ok && !rl_ok)
    PP_HIP(ctx, hipMemsetAsync(const_cast<unsigned char*>(skip_flags) + (p.M + 31) / 32, 1, (size_t)((p.M + 31) / 32), ctx->stream));
  if (d->stride == 2 && s2_classes && (all_f32 || all_planes) && d->in.n_seg == 1) {
    // Stride-2 bwd-data as four stride-1 launches, one per parity class (cy, cx) of the input grid: input cell
    // (2y'+cy, 2x'+cx) only receives the taps ty = ty0 + 2i with ty0 = (cy + pad_t) & 1 (x alike), from output cell
    // y' + (cy + pad_t - ty0)/2 - i.  The single-launch form spends 3/4 of its MFMAs on taps that the divisibility
    // mask zeroes; here every MFMA is useful and classes without taps (1x1 kernels) only run the epilogue.
    IgemmParams q = p;
    q.div = 1;
    const int H = d->in.h[0], W = d->in.w[0];
    bool ok = true;
    IgemmParams cls[4];
    for (int c = 0; c < 4 && ok; ++c) {
      const int cy = c >> 1, cx = c & 1;
      IgemmParams& r = cls[c];
      r = q;
      const int ty0 = (cy + d->pad_t) & 1, tx0 = (cx + d->pad_l) & 1;
      r.kh = ty0 < d->kh ? (d->kh - ty0 + 1) / 2 : 0;
      r.kw = tx0 < d->kw ? (d->kw - tx0 + 1) / 2 : 0;
      r.off_y = (cy + d->pad_t - ty0) / 2;
      r.off_x = (cx + d->pad_l - tx0) / 2;
      r.w_ty0 = ty0; r.w_tx0 = tx0; r.w_tstep = 2; r.w_kw = d->kw;
      if (r.kh == 0 || r.kw == 0) r.kh = r.kw = 0;  // no taps: class_fill_kernel below
      r.seg[0].OH = (H - cy + 1) / 2;
      r.seg[0].OW = (W - cx + 1) / 2;
      r.M = d->in.n_img * r.seg[0].OH * r.seg[0].OW;
      r.sc_on = 1; r.sc_H = H; r.sc_W = W; r.sc_cy = cy; r.sc_cx = cx;
      if (r.M > 0 && r.kh > 0 && !(igemm3_fast_ok(r, all_planes, d->cin, cred / 8) && r.M < (1 << 24))) ok = false;
    }
    if (ok) {
      if (chi) split_capture_pass(ctx->stream, p, chi, clo);
      for (int c = 0; c < 4; ++c) {
        if (cls[c].M <= 0) continue;
        if (cls[c].kh > 0) {
          dispatch3(ctx, cls[c], dy_hi, dy_lo, w_hi, w_lo, d->cin, cred / 8, dx_hi, dx_lo);
        } else {
          const long long total = (long long)cls[c].M * ((cls[c].Nout + 3) >> 2);
          long long blocks = (total + 255) / 256;
          const long long cap = (long long)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
          if (blocks > cap) blocks = cap;
          hipLaunchKernelGGL(class_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, cls[c], addend, relu_src, dx, dx_hi, dx_lo);
        }
      }
      PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_data_bf16x3");
      return PP_OK;
    }
  }
  if (rl_ok) {
    const int nb = (p.M + 31) / 32;
    unsigned char* out_flags = const_cast<unsigned char*>(skip_flags) + nb;
    int* out_list = const_cast<int*>(skip_list_in) + nb + 1;
    if (rl_mode == 22)
      launch_igemm3_rowlist<2, 2>(ctx->stream, p, dy_hi, dy_lo, w_hi, w_lo, d->cin, cred / 8, dx_hi, dx_lo, skip_flags, out_flags, out_list, ctx->ws,
                                  ctx->ws_bytes, !lazy_out);
    else
      launch_igemm3_rowlist<1, 2>(ctx->stream, p, dy_hi, dy_lo, w_hi, w_lo, d->cin, cred / 8, dx_hi, dx_lo, skip_flags, out_flags, out_list, ctx->ws,
                                  ctx->ws_bytes, !lazy_out);
    PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_data_bf16x3");
    return PP_OK;
  }
  dispatch3(ctx, p, dy_hi, dy_lo, w_hi, w_lo, d->cin, cred / 8, dx_hi, dx_lo, chi, clo, skip_flags);
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_data_bf16x3");
  return PP_OK;
}
namespace {

// ------------------------------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix cores:  dw[tap][ci][co] += sum_m x[gather(m,tap)][ci] * dy[m][co]
// The reduction index is the pixel m, but NHWC tensors are channel-contiguous, so both MFMA operands need a
// transpose: the 32-pixel x 128-channel f32 tiles are split to bf16 (hi, lo) while they are staged into LDS
// pixel-major ([pixel][channel], pitch = row + 64 B) and the k-contiguous fragments are fetched with the
// transposing LDS read ds_read_b64_tr_b16 (two reads = 8 pixels of one channel per lane; lane map verified by
// tools/ubench/tr16_check.hip).  With the +64 B pitch the four rows of a 16-lane read group fall in four disjoint
// 16-dword bank windows: conflict free.  Split over pixel ranges + f32 atomics like wgrad_kernel (conv.hip).
typedef short shortx4 __attribute__((ext_vector_type(4)));


template <int TM, int TN, bool AP>
__global__ __launch_bounds__(256, (TM * TN == 4) ? 2 : 3) void wgrad3_kernel(const Wgrad3Params p, const float* __restrict__ g_src,
                                                                            const float* __restrict__ g_dy,
                                                                            const uint4* __restrict__ g_xhi, const uint4* __restrict__ g_xlo,
                                                                            const uint4* __restrict__ g_dhi, const uint4* __restrict__ g_dlo,
                                                                            float* __restrict__ g_dw, float* __restrict__ g_dbias) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int PA = BM + 32, PB = BN + 32;  // LDS pitches in bf16 elements (row + 64 bytes)
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BK * (PA + PB)];
  unsigned short* Xhi = smem;
  unsigned short* Xlo = Xhi + BK * PA;
  unsigned short* Ghi = Xlo + BK * PA;
  unsigned short* Glo = Ghi + BK * PB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_g = p.inv_scale ? *p.inv_scale : 1.f;
  const int wm = wave >> 1, wn = wave & 1;
  int b = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = b % p.n_tiles_n;
  b /= p.n_tiles_n;
  const int tile_k = b % p.n_tiles_k;
  const int split = b / p.n_tiles_k;
  const int tap = tile_k / p.k_tiles_per_tap;
  const int ci0 = (tile_k - tap * p.k_tiles_per_tap) * BM;
  const int ty = tap / p.kw, tx = tap - ty * p.kw;
  const int n0 = tile_n * BN;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int n_steps = (m_end - m_begin + BK - 1) / BK;

  // staging: 8 threads per pixel row, thread loads the float4 channel quads q8 + 8*j
  const int prow = tid >> 3, q8 = tid & 7;
  constexpr int QA = BM / 32, QB = BN / 32;  // quads per thread
  struct WRow { int m, n, y, x, seg_end, sb, OH, OW, SH, SW; } w;
  auto decode = [&](int m) {
    w.m = m;
    int rbeg = p.seg[0].row_begin;
    w.sb = p.seg[0].src_row_begin; w.OH = p.seg[0].OH; w.OW = p.seg[0].OW; w.SH = p.seg[0].SH; w.SW = p.seg[0].SW;
    w.seg_end = p.n_seg > 1 ? p.seg[1].row_begin : p.M;
    for (int s = 1; s < p.n_seg; ++s) {
      if (m >= p.seg[s].row_begin) {
        rbeg = p.seg[s].row_begin; w.sb = p.seg[s].src_row_begin; w.OH = p.seg[s].OH; w.OW = p.seg[s].OW;
        w.SH = p.seg[s].SH; w.SW = p.seg[s].SW;
        w.seg_end = (s + 1 < p.n_seg) ? p.seg[s + 1].row_begin : p.M;
      }
    }
    const int local = m < p.M ? m - rbeg : 0;
    const int hw = w.OH * w.OW;
    w.n = local / hw;
    const int rem = local - w.n * hw;
    w.y = rem / w.OW;
    w.x = rem - w.y * w.OW;
  };
  decode(m_begin + prow);

  constexpr int OA = (BM / 64 > 0) ? BM / 64 : 1, OB = (BN / 64 > 0) ? BN / 64 : 1;  // 8-channel chunks per thread (AP)
  float4 ra[QA], rb[QB];
  uint4 pah[OA], pal[OA], pbh[OB], pbl[OB];
  float4 bsum[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = (g_dbias != nullptr) && (tile_k == 0);

  auto load_step = [&]() {
    const int sy = w.y * p.stride + ty - p.pad_t;
    const int sx = w.x * p.stride + tx - p.pad_l;
    const bool in_rng = w.m < m_end;
    const bool ok = in_rng && ((unsigned)sy < (unsigned)w.SH) && ((unsigned)sx < (unsigned)w.SW);
    const long long xrow = (long long)(w.sb + w.n * w.SH * w.SW + sy * w.SW + sx) * p.ld_src + ci0;
    if (AP) {
#pragma unroll
      for (int j = 0; j < OA; ++j) {
        const long long o8 = 2 * ((xrow >> 3) + q8 + 8 * j);  // packed planes: 32-byte groups
        pah[j] = ok ? g_xhi[o8] : make_uint4(0u, 0u, 0u, 0u);
        pal[j] = ok ? g_xlo[o8] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const int c = n0 + 8 * (q8 + 8 * j);
        const bool okb = in_rng && c < p.ld_dy;
        const long long o8 = 2 * ((((long long)w.m * p.ld_dy + n0) >> 3) + q8 + 8 * j);
        pbh[j] = okb ? g_dhi[o8] : make_uint4(0u, 0u, 0u, 0u);
        pbl[j] = okb ? g_dlo[o8] : make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
      const float* src = g_src + xrow;
#pragma unroll
      for (int j = 0; j < QA; ++j)
        ra[j] = ok ? *reinterpret_cast<const float4*>(src + 4 * (q8 + 8 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float* dyr = g_dy + (long long)w.m * p.ld_dy + n0;
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const int c = n0 + 4 * (q8 + 8 * j);
        rb[j] = (in_rng && c < p.ld_dy) ? *reinterpret_cast<const float4*>(dyr + 4 * (q8 + 8 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // advance this thread's pixel row by one step
    const int m2 = w.m + BK;
    if (m2 >= w.seg_end) {
      decode(m2);
    } else {
      w.m = m2;
      w.x += BK;
      while (w.x >= w.OW) { w.x -= w.OW; ++w.y; }
      while (w.y >= w.OH) { w.y -= w.OH; ++w.n; }
    }
  };
  auto store_step = [&]() {
    if (AP) {
#pragma unroll
      for (int j = 0; j < OA; ++j) {
        const int off = prow * PA + 8 * (q8 + 8 * j);
        if (8 * (q8 + 8 * j) < BM) {
          *reinterpret_cast<uint4*>(Xhi + off) = pah[j];
          *reinterpret_cast<uint4*>(Xlo + off) = pal[j];
        }
      }
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const int off = prow * PB + 8 * (q8 + 8 * j);
        if (8 * (q8 + 8 * j) < BN) {
          *reinterpret_cast<uint4*>(Ghi + off) = pbh[j];
          *reinterpret_cast<uint4*>(Glo + off) = pbl[j];
          if (do_bias) {  // dy = hi + lo (2^-17): only the workgroups of the first k-tile pay for this
            const unsigned int hw[4] = {pbh[j].x, pbh[j].y, pbh[j].z, pbh[j].w}, lw[4] = {pbl[j].x, pbl[j].y, pbl[j].z, pbl[j].w};
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) fmt_value2(hw[e], lw[e], &v[2 * e], &v[2 * e + 1]);
            bsum[2 * j].x += v[0]; bsum[2 * j].y += v[1]; bsum[2 * j].z += v[2]; bsum[2 * j].w += v[3];
            bsum[2 * j + 1].x += v[4]; bsum[2 * j + 1].y += v[5]; bsum[2 * j + 1].z += v[6]; bsum[2 * j + 1].w += v[7];
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < QA; ++j) {
      uint2 hi, lo;
      split4(ra[j], &hi, &lo);
      const int off = prow * PA + 4 * (q8 + 8 * j);
      *reinterpret_cast<uint2*>(Xhi + off) = hi;
      *reinterpret_cast<uint2*>(Xlo + off) = lo;
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      uint2 hi, lo;
      split4(rb[j], &hi, &lo);
      const int off = prow * PB + 4 * (q8 + 8 * j);
      *reinterpret_cast<uint2*>(Ghi + off) = hi;
      *reinterpret_cast<uint2*>(Glo + off) = lo;
      if (do_bias) { bsum[j].x += rb[j].x; bsum[j].y += rb[j].y; bsum[j].z += rb[j].z; bsum[j].w += rb[j].w; }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // transposed-read lane roles (16-lane groups): group g -> columns 16*(g&1).., pixel half h = g>>1
  const int grp = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
  const int cbase = 16 * (grp & 1), hh = grp >> 1;
  const int il = lane & 31, h = lane >> 5;

  if (n_steps > 0) {
    load_step();
    store_step();
  }
  __syncthreads();
  for (int step = 0; step < n_steps; ++step) {
    const bool more = step + 1 < n_steps;
    if (more) load_step();
    wgrad_step<TM, TN>(acc, Xhi, Xlo, Ghi, Glo, PA, PB, wm * 32 * TM + cbase + 4 * gp, wn * 32 * TN + cbase + 4 * gp, hh, gq);
    __syncthreads();
    if (more) {
      store_step();
      __syncthreads();
    }
  }

#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float* dst = g_dw + (long long)(tap * p.Cin + ci) * p.ld_w;
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int co = n0 + wn * 32 * TN + c * 32 + il;
        if (co < p.Cout) atomicAdd(dst + co, acc[a][c][r] * inv_g);
      }
    }
  }
  if (do_bias) {
    // 32 threads (one per pixel row of the step) hold partial sums of the same channel quad: reduce through LDS
    float* red = reinterpret_cast<float*>(smem);  // [32][BN] floats <= LDS size
    __syncthreads();
    if (AP) {
#pragma unroll
      for (int j = 0; j < OB; ++j)
        if (8 * (q8 + 8 * j) < BN) {
          *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j)) = bsum[2 * j];
          *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j) + 4) = bsum[2 * j + 1];
        }
    } else {
#pragma unroll
      for (int j = 0; j < QB; ++j) *reinterpret_cast<float4*>(red + prow * BN + 4 * (q8 + 8 * j)) = bsum[j];
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s += red[r * BN + tid];
      if (n0 + tid < p.Cout) atomicAdd(g_dbias + n0 + tid, s * inv_g);
    }
  }
}

// ---- wgrad3f: the same tiles with a branch-free k-loop (see igemm3f_kernel) ----
// Every step each staging thread re-derives (image, y, x) of its pixel row from m with two reciprocal divisions
// (exact for m < 2^24, host-checked) instead of the incremental walk with its data-dependent loops, addresses both
// operands with 32-bit buffer offsets (out-of-range = zeros: padding taps, rows past the split, columns past ld_dy)
// and converts the loaded tile while its MFMAs drain.  No 64-bit address registers -> 3 workgroups per CU at 128x128.

// SP: the reduction walks a LIST of 32-row blocks (pp_row_block_list: the blocks of dy that hold a non-zero) instead of all
// rows -- the gradient of the 3D-box head is non-zero only around positive anchors (losses.py:332-333 keeps state == 1
// rows), and a block of zero rows adds exactly nothing to dW.  list[0] = number of blocks, list[1..] ascending; the
// splits share the list evenly, so the launch stays balanced whatever the data.
template <int TM, int TN, bool AP, bool SP>
__global__ __launch_bounds__(256, (TM * TN == 4) ? 3 : 4) void wgrad3f_kernel(const Wgrad3Params p, const void* __restrict__ g_x0,
                                                                            const void* __restrict__ g_x1, unsigned x_bytes,
                                                                            const void* __restrict__ g_d0, const void* __restrict__ g_d1,
                                                                            unsigned d_bytes, float* __restrict__ g_dw,
                                                                            float* __restrict__ g_dbias, const int* __restrict__ g_list,
                                                                            float* __restrict__ g_ws, long long ws_slice) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int PA = BM + 32, PB = BN + 32;  // LDS pitches in bf16 elements (row + 64 bytes)
  constexpr int ES = 4;              // f32, or packed planes (32-byte groups of 8 channels: hi, then lo = the *1 resources)
  constexpr int CB = AP ? 32 : 16;   // bytes between the 16-byte chunks a staging thread fetches from one resource
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BK * (PA + PB)];
  unsigned short* Xhi = smem;
  unsigned short* Xlo = Xhi + BK * PA;
  unsigned short* Ghi = Xlo + BK * PA;
  unsigned short* Glo = Ghi + BK * PB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_g = p.inv_scale ? *p.inv_scale : 1.f;
  const int wm = wave >> 1, wn = wave & 1;
  int b = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = b % p.n_tiles_n;
  b /= p.n_tiles_n;
  const int tile_k = b % p.n_tiles_k;
  const int split = b / p.n_tiles_k;
  const int tap = tile_k / p.k_tiles_per_tap;
  const int ci0 = (tile_k - tap * p.k_tiles_per_tap) * BM;
  const int ty = tap / p.kw, tx = tap - ty * p.kw;
  const int n0 = tile_n * BN;
  int m_begin = split * p.rows_per_split;
  int m_end = min(p.M, m_begin + p.rows_per_split);
  int n_steps = (m_end - m_begin + BK - 1) / BK;
  int s_idx = 0, s_end = 0;
  const int blk_past = (p.M + BK - 1) / BK;  // SP: "no more blocks" (its rows are >= M: loads return zeros)
  if (SP) {
    // The grid was sized on the host for the dense reduction; a short list is shared by FEWER splits (uniform early exit of
    // the others, before any barrier): every split adds a full |dW| tile set with float atomics, which at 12 blocks per
    // split cost more than the reduction itself (3D-box head on the bench targets: 16 splits x 9.4 MB of atomics for ~200
    // listed blocks).  The slices of the deterministic mode need every split's slice written: no reduction there.
    const int n_act = g_list[0];
    int s_eff = (n_act + p.sp_min_steps - 1) / p.sp_min_steps;
    s_eff = s_eff < 1 ? 1 : (s_eff > p.splits ? p.splits : s_eff);
    if (g_ws) s_eff = p.splits;
    if (split >= s_eff) return;
    s_idx = (int)((long long)n_act * split / s_eff);
    s_end = (int)((long long)n_act * (split + 1) / s_eff);
    n_steps = s_end - s_idx;
    m_begin = (n_steps > 0 ? g_list[1 + s_idx] : blk_past) * BK;
    m_end = p.M;
  }

  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_x0), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AP ? g_x1 : g_x0), 0, AP ? x_bytes - 16 : x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_d0), 0, d_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(AP ? g_d1 : g_d0), 0, AP ? d_bytes - 16 : d_bytes, 0x00020000);

  // staging: 8 threads per pixel row; thread q8 loads the 16-byte chunks q8 + 8*j of its row
  const int prow = tid >> 3, q8 = tid & 7;
  constexpr int QA = AP ? (BM / 64) : (BM / 32), QB = AP ? (BN / 64) : (BN / 32);  // 16-byte chunks per thread and operand
  constexpr int CH = AP ? 8 : 4;                                                    // channels per chunk
  uint4 ra[QA], rb[QB];                // f32 quads, or bf16 hi chunks (AP)
  uint4 ral[AP ? QA : 1], rbl[AP ? QB : 1];  // bf16 lo chunks (AP)
  uint2 sah[AP ? 1 : QA], sal[AP ? 1 : QA], sbh[AP ? 1 : QB], sbl[AP ? 1 : QB];  // converted tile (f32 path)
  float4 bsum[BN / 32];
#pragma unroll
  for (int j = 0; j < BN / 32; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = (g_dbias != nullptr) && (tile_k == 0);
  // Per-thread walk over its pixel rows m = m_begin + prow + 32 * step: (image base, y, x) advance incrementally
  // (x += 32 with a bounded number of row wraps; wraps = ceil(32 / narrowest level)); only when a lane crosses into
  // the next pyramid level does the wave take the (wave-uniform, rare) full decode.
  int m_cur = m_begin + prow;
  int c_OW, c_OH, c_SW, c_SH, c_end, r_x, r_y, r_img;  // level constants and position of row m_cur
  auto decode = [&](int m) {
    int rbeg = p.seg[0].row_begin, sb = p.seg[0].src_row_begin;
    c_OH = p.seg[0].OH; c_OW = p.seg[0].OW; c_SH = p.seg[0].SH; c_SW = p.seg[0].SW;
    c_end = p.n_seg > 1 ? p.seg[1].row_begin : 0x7fffffff;
    for (int s = 1; s < p.n_seg; ++s) {
      const bool in = m >= p.seg[s].row_begin;
      rbeg = in ? p.seg[s].row_begin : rbeg;
      sb = in ? p.seg[s].src_row_begin : sb;
      c_OH = in ? p.seg[s].OH : c_OH;
      c_OW = in ? p.seg[s].OW : c_OW;
      c_SH = in ? p.seg[s].SH : c_SH;
      c_SW = in ? p.seg[s].SW : c_SW;
      c_end = in ? ((s + 1 < p.n_seg) ? p.seg[s + 1].row_begin : 0x7fffffff) : c_end;
    }
    const int local = m < p.M ? m - rbeg : 0;
    const int hw = c_OH * c_OW;
    int rem;
    const int n = div_small(local, hw, __frcp_rn((float)hw), &rem);
    r_y = div_small(rem, c_OW, __frcp_rn((float)c_OW), &r_x);
    r_img = sb + n * c_SH * c_SW;
  };
  decode(m_cur);
  const int n_wraps = p.max_wraps;
  int q_next = blk_past;  // SP: the listed block of the step after the one being loaded
  if (SP) q_next = (s_idx + 1 < s_end) ? g_list[1 + s_idx + 1] : blk_past;

  auto load_step = [&]() {
    const int m = m_cur;
    const bool in_rng = m < m_end;
    const int sy = r_y * p.stride + ty - p.pad_t, sx = r_x * p.stride + tx - p.pad_l;
    const bool ok = in_rng && ((unsigned)sy < (unsigned)c_SH) && ((unsigned)sx < (unsigned)c_SW);
    int xo = ((r_img + sy * c_SW + sx) * p.ld_src + ci0) * ES + CB * q8;
    xo = ok ? xo : PP_BUF_OOB;
#pragma unroll
    for (int j = 0; j < QA; ++j) {
      ra[j] = buf_load16(rs_x0, xo, 8 * CB * j);
      if (AP) ral[j] = buf_load16(rs_x1, xo, 8 * CB * j);
    }
    const int dyo = (m * p.ld_dy + n0) * ES + CB * q8;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const bool okb = in_rng && (n0 + CH * (q8 + 8 * j) < p.ld_dy);
      const int o = okb ? dyo : PP_BUF_OOB;
      rb[j] = buf_load16(rs_d0, o, 8 * CB * j);
      if (AP) rbl[j] = buf_load16(rs_d1, o, 8 * CB * j);
    }
  };
  auto next_row = [&]() {  // m_cur += 32
    if (SP) {  // jump to the next listed block: full decode (two reciprocal divisions per step).  The list entry was fetched one
               // step ahead (q_next): a dependent global load in front of the operand loads cost ~1 us per step
      ++s_idx;
      m_cur = q_next * BK + prow;
      q_next = (s_idx + 1 < s_end) ? g_list[1 + s_idx + 1] : blk_past;
      decode(m_cur);
      return;
    }
    m_cur += BK;
    if (__builtin_amdgcn_ballot_w64(m_cur >= c_end) != 0) {
      decode(m_cur);
    } else {
      r_x += BK;
      for (int i = 0; i < n_wraps; ++i) {
        const bool w = r_x >= c_OW;
        r_x -= w ? c_OW : 0;
        r_y += w ? 1 : 0;
      }
      const bool wy = r_y >= c_OH;  // (32 consecutive cells never span more than one image boundary: OH * OW >= 32 is host-checked)
      r_y -= wy ? c_OH : 0;
      r_img += wy ? c_SH * c_SW : 0;
    }
  };
  auto split_step = [&](auto with_bias) {  // conversion (f32 path) and the bias partial sums: VALU only, under the MFMAs
    constexpr bool WB = decltype(with_bias)::value;
    const float bmask = 1.f;
    if (AP) {
      if (!WB) return;
#pragma unroll
      for (int j = 0; j < QB; ++j) {  // dy = hi + lo (2^-17)
        const unsigned int hw4[4] = {rb[j].x, rb[j].y, rb[j].z, rb[j].w}, lw4[4] = {rbl[j].x, rbl[j].y, rbl[j].z, rbl[j].w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) fmt_value2(hw4[e], lw4[e], &v[2 * e], &v[2 * e + 1]);
        bsum[2 * j].x = fmaf(v[0], bmask, bsum[2 * j].x); bsum[2 * j].y = fmaf(v[1], bmask, bsum[2 * j].y);
        bsum[2 * j].z = fmaf(v[2], bmask, bsum[2 * j].z); bsum[2 * j].w = fmaf(v[3], bmask, bsum[2 * j].w);
        bsum[2 * j + 1].x = fmaf(v[4], bmask, bsum[2 * j + 1].x); bsum[2 * j + 1].y = fmaf(v[5], bmask, bsum[2 * j + 1].y);
        bsum[2 * j + 1].z = fmaf(v[6], bmask, bsum[2 * j + 1].z); bsum[2 * j + 1].w = fmaf(v[7], bmask, bsum[2 * j + 1].w);
      }
    } else {
#pragma unroll
      for (int j = 0; j < QA; ++j) split4(*reinterpret_cast<const float4*>(&ra[j]), &sah[j], &sal[j]);
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(&rb[j]);
        split4(v, &sbh[j], &sbl[j]);
        if (WB) { bsum[j].x += v.x; bsum[j].y += v.y; bsum[j].z += v.z; bsum[j].w += v.w; }
      }
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int j = 0; j < QA; ++j) {
      const int off = prow * PA + CH * (q8 + 8 * j);
      if (AP) {
        *reinterpret_cast<uint4*>(Xhi + off) = ra[j];
        *reinterpret_cast<uint4*>(Xlo + off) = ral[j];
      } else {
        *reinterpret_cast<uint2*>(Xhi + off) = sah[j];
        *reinterpret_cast<uint2*>(Xlo + off) = sal[j];
      }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int off = prow * PB + CH * (q8 + 8 * j);
      if (AP) {
        *reinterpret_cast<uint4*>(Ghi + off) = rb[j];
        *reinterpret_cast<uint4*>(Glo + off) = rbl[j];
      } else {
        *reinterpret_cast<uint2*>(Ghi + off) = sbh[j];
        *reinterpret_cast<uint2*>(Glo + off) = sbl[j];
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // transposed-read lane roles (16-lane groups): group g -> columns 16*(g&1).., pixel half h = g>>1
  const int grp = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
  const int cbase = 16 * (grp & 1), hh = grp >> 1;
  const int il = lane & 31, h = lane >> 5;

  auto k_loop = [&](auto with_bias) {
    load_step();  // (n_steps == 0: every row is out of range -> zeros)
    split_step(with_bias);
    store_step();
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
      next_row();
      load_step();  // past the last step m >= m_end: zeros, never stored
      __builtin_amdgcn_sched_barrier(0);
      wgrad_step<TM, TN>(acc, Xhi, Xlo, Ghi, Glo, PA, PB, wm * 32 * TM + cbase + 4 * gp, wn * 32 * TN + cbase + 4 * gp, hh, gq);
      split_step(with_bias);
      __syncthreads();
      store_step();
      __syncthreads();
    }
  };
  if (do_bias) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  // Reduction over the row splits.  With a scratch buffer (pp_ctx_set_workspace): split s writes its partial tile -- plain
  // stores, every column of the tile below ld_w, zeros past Cout -- into slice s, a full-size copy of dW (+ one bias row), and
  // wgrad_finish_kernel adds the slices in a fixed order: deterministic, and the partial sums leave the chip at store speed
  // instead of the ~1.3 TB/s of float atomics (MI355X_MICROARCH.md: they execute at the memory side).  Without it: atomics.
  float* const slice = g_ws ? g_ws + (long long)split * ws_slice : nullptr;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long long row = (long long)(tap * p.Cin + ci) * p.ld_w;
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int co = n0 + wn * 32 * TN + c * 32 + il;
        if (slice) {
          if (co < p.ld_w) slice[row + co] = co < p.Cout ? acc[a][c][r] * inv_g : 0.f;
        } else if (co < p.Cout) {
          atomicAdd(g_dw + row + co, acc[a][c][r] * inv_g);
        }
      }
    }
  }
  if (do_bias) {
    // 32 threads (one per pixel row of the step) hold partial sums of the same channels: reduce through LDS
    float* red = reinterpret_cast<float*>(smem);  // [32][BN] floats <= LDS size
    __syncthreads();
    if (AP) {
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j)) = bsum[2 * j];
        *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j) + 4) = bsum[2 * j + 1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < QB; ++j) *reinterpret_cast<float4*>(red + prow * BN + 4 * (q8 + 8 * j)) = bsum[j];
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s += red[r * BN + tid];
      if (slice) {
        if (n0 + tid < p.ld_w) slice[ws_slice - p.ld_w + n0 + tid] = n0 + tid < p.Cout ? s * inv_g : 0.f;  // the bias row closes the slice
      } else if (n0 + tid < p.Cout) {
        atomicAdd(g_dbias + n0 + tid, s * inv_g);
      }
    }
  }
}

// dw += sum_s slice_s (fixed order), dbias += the slices' last row; n4 = float4 groups of one slice (dW + the bias row)
__global__ void wgrad_finish_kernel(long long n4_dw, int ld4, int splits, long long ws_slice4, const float4* __restrict__ ws, float4* __restrict__ dw,
                                    float4* __restrict__ dbias) {
  const long long total = n4_dw + (dbias ? ld4 : 0);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const bool is_bias = i >= n4_dw;
    const long long src = is_bias ? ws_slice4 - ld4 + (i - n4_dw) : i;
    float4 v = ws[src];
    for (int s = 1; s < splits; ++s) {
      const float4 q = ws[src + s * ws_slice4];
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    float4* dst = is_bias ? dbias + (i - n4_dw) : dw + i;
    float4 o = *dst;
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *dst = o;
  }
}

template <int TM, int TN>
static bool launch_wgrad3(pp_ctx* ctx, Wgrad3Params& p, const float* x, const float* dy, const void* xhi, const void* xlo, const void* dhi,
                          const void* dlo, float* dw, float* dbias, const int* list = nullptr, bool lazy_in = false) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.k_tiles_per_tap = p.Cin / BM;
  p.n_tiles_k = p.kh * p.kw * p.k_tiles_per_tap;
  p.n_tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles = p.n_tiles_k * p.n_tiles_n;
  const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
  int max_splits = (p.M + 511) / 512;  // at least 16 reduction steps of 32 rows per workgroup
  if (max_splits < 1) max_splits = 1;
  if (max_splits > 64) max_splits = 64;
  static const bool fast_on = []() { const char* e = getenv("PP_CONV3_FAST"); return !(e && e[0] == '0'); }();
  const long long x_bytes = p.src_rows * (long long)p.ld_src * 4, d_bytes = (long long)p.M * p.ld_dy * 4;  // f32 or packed planes
  int min_ow = 1 << 30, min_hw = 1 << 30;
  for (int i = 0; i < p.n_seg; ++i) {
    min_ow = p.seg[i].OW < min_ow ? p.seg[i].OW : min_ow;
    min_hw = p.seg[i].OH * p.seg[i].OW < min_hw ? p.seg[i].OH * p.seg[i].OW : min_hw;
  }
  p.max_wraps = (32 + min_ow - 1) / min_ow;
  const bool fast = fast_on && x_bytes < (1ll << 31) && d_bytes < (1ll << 31) && p.M < (1 << 24) && p.src_rows > 0 && min_hw >= 32;
  if (lazy_in && !(fast && list)) return false;  // (only the listed-block reduction never looks at the other rows of dy)
  const int slots = fast ? ((TM * TN == 4) ? 3 : 4) : ((TM * TN == 4) ? 2 : 3);
  const double tile_work = (double)(TM * TN) / 4.0;
  // one slice = a full-size copy of dW plus a bias row (f32).  PP_WGRAD3_DETERMINISTIC=1 (and a scratch buffer): the splits write
  // slices and a finishing pass adds them in a fixed order -- bit-reproducible gradients; default: float atomics into dW, which
  // are fire-and-forget and overlap the k-loops of other workgroups (measured in the training step, same box, slices vs
  // atomics: 531.5 vs 537.1 images/s, dense backward 424 vs 433: the extra pass and its launch cost more than the atomics)
  const long long ws_slice = (long long)p.kh * p.kw * p.Cin * p.ld_w + p.ld_w;
  const bool use_ws = fast && ctx->ws != nullptr && (long long)ctx->ws_bytes >= ws_slice * 4 && ((uintptr_t)dw & 15u) == 0 &&
                      (!dbias || ((uintptr_t)dbias & 15u) == 0) && []() { const char* e = getenv("PP_WGRAD3_DETERMINISTIC"); return e && e[0] == '1'; }();
  if (use_ws) {
    const long long fit = (long long)ctx->ws_bytes / (ws_slice * 4);
    if (max_splits > fit) max_splits = (int)fit;
  }
  // per split: atomics = |dW| at ~1.3 TB/s, half of it hidden under the k-loops of other workgroups (measured: 256-channel
  // head convs 252 -> 234 us going from 14 to 21 splits); slices = |dW| written and read back once at ~4 TB/s
  const double atomic_us_per_split = use_ws ? (double)tiles * BM * BN * 4.0 * 2.0 / 4.0e6 : 0.5 * (double)tiles * BM * BN * 4.0 / 1.3e6;
  int splits = 1;
  double best = 1e300;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const double steps = (double)((p.M + sp - 1) / sp + 31) / 32.0;
    const double cost = est_rounds((long long)tiles * sp, slots, cus) * steps * tile_work * 0.8 + atomic_us_per_split * sp;
    if (cost < best * 0.995) {
      best = cost;
      splits = sp;
    }
  }
  if (const char* e = getenv("PP_WGRAD3_SPLITS")) {  // tuning hook
    const int v = atoi(e);
    if (v >= 1 && v <= max_splits) splits = v;
  }
  if (getenv("PP_CONV_DEBUG")) fprintf(stderr, "wgrad3 tile %dx%d tiles %d splits %d (M %d) fast %d\n", BM, BN, tiles, splits, p.M, (int)fast);
  int rps = (p.M + splits - 1) / splits;
  rps = (rps + 31) / 32 * 32;
  splits = (p.M + rps - 1) / rps;
  p.splits = splits;
  p.rows_per_split = rps;
  {
    static const int sp_steps = []() { const char* e = getenv("PP_WGRAD3_SP_STEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1; }();
    p.sp_min_steps = sp_steps;
  }
  float* const ws = (use_ws && splits > 1) ? ctx->ws : nullptr;  // (a single split adds its tile straight into dW)
  if (fast) {
    if (xhi && list)
      hipLaunchKernelGGL((wgrad3f_kernel<TM, TN, true, true>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, xhi, xlo,
                         (unsigned)x_bytes, dhi, dlo, (unsigned)d_bytes, dw, dbias, list, ws, ws_slice);
    else if (xhi)
      hipLaunchKernelGGL((wgrad3f_kernel<TM, TN, true, false>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, xhi, xlo,
                         (unsigned)x_bytes, dhi, dlo, (unsigned)d_bytes, dw, dbias, (const int*)nullptr, ws, ws_slice);
    else if (list)
      hipLaunchKernelGGL((wgrad3f_kernel<TM, TN, false, true>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, (const void*)x,
                         nullptr, (unsigned)x_bytes, (const void*)dy, nullptr, (unsigned)d_bytes, dw, dbias, list, ws, ws_slice);
    else
      hipLaunchKernelGGL((wgrad3f_kernel<TM, TN, false, false>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, (const void*)x,
                         nullptr, (unsigned)x_bytes, (const void*)dy, nullptr, (unsigned)d_bytes, dw, dbias, (const int*)nullptr, ws, ws_slice);
    if (ws) {
      const long long n4 = (ws_slice - p.ld_w) / 4;
      long long blocks = (n4 + p.ld_w / 4 + 255) / 256;
      if (blocks > (long long)cus * 8) blocks = (long long)cus * 8;
      hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n4, p.ld_w / 4, splits, ws_slice / 4,
                         (const float4*)ws, (float4*)dw, (float4*)dbias);
    }
  } else if (xhi)
    hipLaunchKernelGGL((wgrad3_kernel<TM, TN, true>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, x, dy, (const uint4*)xhi,
                       (const uint4*)xlo, (const uint4*)dhi, (const uint4*)dlo, dw, dbias);
  else
    hipLaunchKernelGGL((wgrad3_kernel<TM, TN, false>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, x, dy,
                       (const uint4*)nullptr, (const uint4*)nullptr, (const uint4*)nullptr, (const uint4*)nullptr, dw, dbias);
  return true;
}

}  // namespace
extern "C" int PP_API(pp_conv2d_nhwc_bwd_weight_bf16x3)(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy, const void* x_hi,
                                                const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, float* dbias) {
  PP_REQUIRE_CTX(ctx);
  const int* skip_list = ctx->skip_list;  // one-shot (pp_ctx_set_row_block_skip)
  ctx->skip_list = nullptr;
  ctx->skip_flags = nullptr;
  const bool lazy_in = ctx->lazy_in != 0;  // one-shot (pp_ctx_set_row_block_lazy): dy may hold anything outside the listed blocks
  ctx->lazy_out = ctx->lazy_in = 0;
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_bwd_weight_bf16x3");
  if (rc) return rc;
  const bool planes = x_hi && x_lo && dy_hi && dy_lo;
  PP_CHECK_ARG(ctx, ((x && dy) || planes) && dw, PP_ERR_ARG, "pp_conv2d_nhwc_bwd_weight_bf16x3: null tensor");
  PP_CHECK_ARG(ctx, planes || !(x_hi || x_lo || dy_hi || dy_lo), PP_ERR_ARG, "pp_conv2d_nhwc_bwd_weight_bf16x3: all four planes or none");
  PP_CHECK_ARG(ctx, !planes || (d->ld_x % 8 == 0 && d->ld_y % 8 == 0), PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_weight_bf16x3: planes need ld % 8 == 0");
  PP_CHECK_ARG(ctx, !planes || (pp_is_packed(x_hi, x_lo) && pp_is_packed(dy_hi, dy_lo)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_weight_bf16x3: planes must be packed (lo = hi + 16 bytes, 16-byte aligned)");
  PP_CHECK_ARG(ctx, d->cin % 64 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_weight_bf16x3: cin %d must be a multiple of 64", d->cin);
  PP_CHECK_ARG(ctx, d->ld_y % 4 == 0 && d->ld_x % 4 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_weight_bf16x3: leading dims must be multiples of 4");
  PP_CHECK_ARG(ctx, (!x || pp_is_aligned16(x)) && (!dy || pp_is_aligned16(dy)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_weight_bf16x3: tensors must be 16-byte aligned");
  Wgrad3Params p;
  memset(&p, 0, sizeof(p));
  p.ld_src = d->ld_x; p.ld_dy = d->ld_y; p.ld_w = d->ld_w;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M, &p.src_rows);
  p.inv_scale = (planes && ctx->grad_scale) ? ctx->grad_scale + 1 : nullptr;  // (f32 operands are never scaled)
  p.Cin = d->cin; p.Cout = d->cout;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  // (tools/sweep_wgrad.sh: the small 1x1 layers of res4 / res5 take ~35 us for ANY tile / split choice: ~19 latency-bound
  // steps of a lone workgroup per CU plus the f32 atomics of splits x |dW|.  Register double-buffering of the staged tiles
  // (loads two steps ahead) shortens such lone-workgroup launches by 10-20 % in isolation, but no measurable step time.)
  {
    // Round 4: 3x3 stride-1 "same" layers on plane-stored operands -> the tap-row-reuse kernels of conv4.hip (one staged tile pair per
    // kernel ROW, 192 accumulators).  PP_WGRAD3R (read per launch: tests compare the kernels in one process):
    //   unset  wgrad3w (producer + consumer waves) for the DENSE reductions of the P16 launches with 512 input or output channels --
    //          the 3D-box head: 5-19 % faster per launch, +2.8 % on the dense-backward step; the 256-wide launches gain 5-13 % in
    //          isolation and LOSE 1 % in the step (a workgroup of 8 waves x 256 registers owns its CU: nothing of the data-gradient
    //          lane runs beside it), the bf16-pair launches of the backbone lose outright (profiles/r04_wgrad_producer_consumer_waves.txt);
    //          PP_WGRAD3W_MIN moves the channel threshold
    //   0      wgrad3f everywhere
    //   1      wgrad3r (one wave per SIMD; dense launches 1.15-1.45x SLOWER than wgrad3f: profiles/r04_wgrad_tap_row_reuse.txt)
    //   2      wgrad3w wherever its conditions hold
    // The deterministic slices mode keeps wgrad3f.
    const char* const e_r = getenv("PP_WGRAD3R");
    const char* const e_det = getenv("PP_WGRAD3_DETERMINISTIC");  // (read per launch, like the slices path itself)
    const bool det = e_det && e_det[0] == '1';
    static const int w_min = []() { const char* e = getenv("PP_WGRAD3W_MIN"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int variant = 0;
    if (e_r && (e_r[0] == '1' || e_r[0] == '2')) variant = e_r[0] - '0';
    else if (!e_r && PP_FMT == 1 && !skip_list && (d->cin >= w_min || d->cout >= w_min)) variant = 2;
    if (variant && planes && !(det && ctx->ws != nullptr) && (!lazy_in || skip_list)) {
      static const int sp_steps = []() { const char* e = getenv("PP_WGRAD3_SP_STEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1; }();
      p.sp_min_steps = sp_steps;
      if (PP_API(pp4_launch_wgrad3r)(ctx->stream, p, x_hi, x_lo, dy_hi, dy_lo, dw, dbias, skip_list, ctx->n_cu > 0 ? ctx->n_cu : 256, variant)) {
        PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_weight_bf16x3");
        return PP_OK;
      }
    }
  }
  bool big_k = (d->cin % 128 == 0);
  bool big_n = ((d->cout + 127) / 128 * 128) <= ((d->cout + 63) / 64 * 64);
  if (const char* e = getenv("PP_WGRAD3_TILE")) {  // tuning hook: "1,1" / "1,2" / "2,1" / "2,2"
    if (e[0] && e[1] == ',' && e[2]) {
      big_k = big_k && e[0] == '2';
      big_n = e[2] == '2';
    }
  }
  bool launched;
  if (big_k && big_n) launched = launch_wgrad3<2, 2>(ctx, p, x, dy, planes ? x_hi : nullptr, x_lo, dy_hi, dy_lo, dw, dbias, skip_list, lazy_in);
  else if (big_k) launched = launch_wgrad3<2, 1>(ctx, p, x, dy, planes ? x_hi : nullptr, x_lo, dy_hi, dy_lo, dw, dbias, skip_list, lazy_in);
  else if (big_n) launched = launch_wgrad3<1, 2>(ctx, p, x, dy, planes ? x_hi : nullptr, x_lo, dy_hi, dy_lo, dw, dbias, skip_list, lazy_in);
  else launched = launch_wgrad3<1, 1>(ctx, p, x, dy, planes ? x_hi : nullptr, x_lo, dy_hi, dy_lo, dw, dbias, skip_list, lazy_in);
  PP_CHECK_ARG(ctx, launched, PP_ERR_ARG,
               "pp_conv2d_nhwc_bwd_weight_bf16x3: dy was declared lazy (unwritten outside its listed blocks) but the listed-block reduction "
               "does not apply to this call");
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_weight_bf16x3");
  return PP_OK;
}
namespace {
}  // namespace
