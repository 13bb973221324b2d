// Round 4: the 1x1 convolutions of the ResNet bottlenecks (models/resnet.py:87-110: branch2a / branch2c / branch1 of every block,
// forward and data gradient) as a PERSISTENT, LDS-DMA-pipelined GEMM -- igemm4p_kernel.
//
// Why.  At batch 8 the backbone is ~120 launches of 25-90 us at 4-7 % of the matrix peak (profiles/r03_ops_one_lane_dense.csv):
// res5c_branch2c moves 48 MB and 5 GFLOP in 32 us where either bound is ~10 us.  The register-staged kernel (igemm3f) prefetches
// ONE 32-channel step ahead; with about one workgroup per CU nothing else covers a step's L2 / HBM latency, so a 16-step tile
// pays 16 latencies, then an epilogue, and every workgroup of the one-round launch does so in lock-step (the chip alternates
// between reading and writing).  Here:
//  * one workgroup of 4 waves per CU, tile 128 x 128 (wave tile 64 x 64, mma_step of planes_fmt.h: the same products in the same
//    order as igemm3f -- results are bit-identical to it for an unsplit reduction);
//  * both operands reach LDS by LDS-DMA into a ring of NST (3 or 4) stages of 32 KB (gathered rows 128 x 128 B in full lines +
//    both weight planes 2 x 128 x 64 B: the images of igemm4x, planes_fmt.h LAY 1), NST - 1 steps in flight ACROSS the one
//    barrier per step, counted vmcnt -- a step costs max(MFMA, bandwidth), not a latency;
//  * PERSISTENT: the grid is min(items, CUs); a workgroup walks its items (tile x reduction split) and the ring runs ahead
//    across item boundaries, so the first stages of the next tile land while the current tile's epilogue reads its residual and
//    writes its 64 KB -- the read and write phases of different CUs drift apart instead of alternating chip-wide;
//  * every stage is an LDS object of its own (the compiler tracks LDS-DMA per object), the epilogue stages through a sixth 32 KB
//    object: 5 x 32 KB = all 163 840 bytes of the CU with NST = 4, 128 KB with NST = 3.
// Conditions (host-checked in conv3.hip's dispatch): 1x1, plane-stored gathered operand, no parity-class scatter, buffers < 2 GiB.
#include "conv_common.h"

#define PP_CAT_(a, b) a##b
#define PP_CAT(a, b) PP_CAT_(a, b)
#if PP_FMT == 1
#define PP_API(name) PP_CAT(name, _fmt1)
#else
#define PP_API(name) PP_CAT(name, _fmt0)
#endif
#include "conv4.h"
#ifndef PP_W3W_LA
#define PP_W3W_LA 2  // wgrad3w: fragment reads in flight ahead of the MFMAs, in units
#endif

namespace {
#include "planes_fmt.h"
#include "conv3_shared.h"

// ---- fragment reads out of the compiler's sight ----
// The k-loop below keeps NST - 1 LDS-DMA batches in flight across its barriers.  hipcc tracks LDS-DMA per LDS object and puts a
// vmcnt wait in front of every ds_read that may alias a pending batch; across the control flow of this kernel (item boundaries,
// the epilogue's own waits) its bookkeeping falls back to "any batch", and the wait it inserts -- vmcnt(8): everything but the
// batch just issued -- turns the ring into a one-step prefetch (first build: 43 us where igemm3f took 28).  So the fragment reads
// are inline assembly, ordered by hand: [s_waitcnt vmcnt(N); s_barrier] in front of them (step()), then a counted lgkmcnt wait
// that names every destination register as read-write, so that no MFMA can be scheduled above it
// (cdna_hip_programming.md section 5.7, form (ii)).
template <int OFF>
__device__ __forceinline__ void lds_rd16(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
#define PP_WAIT8(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])::"memory")
#define PP_WAIT4(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory")

// One 32-deep k-step of a wave's 64 x 64 tile (TM = TN = 2) from a stage at LDS byte address `st` (gathered image at + 0, weight
// planes at + 16 KB / + 24 KB: planes_fmt.h LAY 1); a_hi / a_lo / b_hi: this lane's fragment offsets [s][block] inside the stage.
// Same products in the same order as mma_step<2, 2, 128, 128, *, 1> of planes_fmt.h.
__device__ __forceinline__ void mma_step_asm(floatx16 (&acc)[2][2], unsigned st, const unsigned (&a_hi)[2][2], const unsigned (&a_lo)[2][2],
                                             const unsigned (&b_hi)[2][2]) {
  constexpr int BH = 16384, BL = 16384 + 8192;
#if PP_FMT == 1
  u32x4 lo[8], h0[4], h1[4];  // lo: [Bl s0 b0, Bl s0 b1, Bl s1 b0, Bl s1 b1, Al s0 a0, Al s0 a1, Al s1 a0, Al s1 a1]
  lds_rd16<BL>(lo[0], st + b_hi[0][0]); lds_rd16<BL>(lo[1], st + b_hi[0][1]); lds_rd16<BL>(lo[2], st + b_hi[1][0]); lds_rd16<BL>(lo[3], st + b_hi[1][1]);
  lds_rd16<0>(lo[4], st + a_lo[0][0]); lds_rd16<0>(lo[5], st + a_lo[0][1]); lds_rd16<0>(lo[6], st + a_lo[1][0]); lds_rd16<0>(lo[7], st + a_lo[1][1]);
  lds_rd16<BH>(h0[0], st + b_hi[0][0]); lds_rd16<BH>(h0[1], st + b_hi[0][1]); lds_rd16<0>(h0[2], st + a_hi[0][0]); lds_rd16<0>(h0[3], st + a_hi[0][1]);
  lds_rd16<BH>(h1[0], st + b_hi[1][0]); lds_rd16<BH>(h1[1], st + b_hi[1][1]); lds_rd16<0>(h1[2], st + a_hi[1][0]); lds_rd16<0>(h1[3], st + a_hi[1][1]);
  PP_WAIT8(8, lo);
  {
    intx8 bq[2], aq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bq[i][0] = (int)lo[i][0]; bq[i][1] = (int)lo[i][1]; bq[i][2] = (int)lo[i][2]; bq[i][3] = (int)lo[i][3];
      bq[i][4] = (int)lo[2 + i][0]; bq[i][5] = (int)lo[2 + i][1]; bq[i][6] = (int)lo[2 + i][2]; bq[i][7] = (int)lo[2 + i][3];
      aq[i][0] = (int)lo[4 + i][0]; aq[i][1] = (int)lo[4 + i][1]; aq[i][2] = (int)lo[4 + i][2]; aq[i][3] = (int)lo[4 + i][3];
      aq[i][4] = (int)lo[6 + i][0]; aq[i][5] = (int)lo[6 + i][1]; aq[i][6] = (int)lo[6 + i][2]; aq[i][7] = (int)lo[6 + i][3];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[a], bq[b], acc[a][b], 1, 1, 0, P16_SCALES, 1, P16_SCALES);
  }
  PP_WAIT4(4, h0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, h0[2 + a]), __builtin_bit_cast(halfx8, h0[b]), acc[a][b], 0, 0, 0);
  PP_WAIT4(0, h1);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, h1[2 + a]), __builtin_bit_cast(halfx8, h1[b]), acc[a][b], 0, 0, 0);
#else
  u32x4 f0[8], f1[8];  // per half-step: [Bh b0, Bh b1, Bl b0, Bl b1, Ah a0, Ah a1, Al a0, Al a1]
  lds_rd16<BH>(f0[0], st + b_hi[0][0]); lds_rd16<BH>(f0[1], st + b_hi[0][1]); lds_rd16<BL>(f0[2], st + b_hi[0][0]); lds_rd16<BL>(f0[3], st + b_hi[0][1]);
  lds_rd16<0>(f0[4], st + a_hi[0][0]); lds_rd16<0>(f0[5], st + a_hi[0][1]); lds_rd16<0>(f0[6], st + a_lo[0][0]); lds_rd16<0>(f0[7], st + a_lo[0][1]);
  lds_rd16<BH>(f1[0], st + b_hi[1][0]); lds_rd16<BH>(f1[1], st + b_hi[1][1]); lds_rd16<BL>(f1[2], st + b_hi[1][0]); lds_rd16<BL>(f1[3], st + b_hi[1][1]);
  lds_rd16<0>(f1[4], st + a_hi[1][0]); lds_rd16<0>(f1[5], st + a_hi[1][1]); lds_rd16<0>(f1[6], st + a_lo[1][0]); lds_rd16<0>(f1[7], st + a_lo[1][1]);
  PP_WAIT8(8, f0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0[6 + a]), __builtin_bit_cast(bf16x8, f0[b]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0[4 + a]), __builtin_bit_cast(bf16x8, f0[2 + b]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0[4 + a]), __builtin_bit_cast(bf16x8, f0[b]), acc[a][b], 0, 0, 0);
    }
  PP_WAIT8(0, f1);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1[6 + a]), __builtin_bit_cast(bf16x8, f1[b]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1[4 + a]), __builtin_bit_cast(bf16x8, f1[2 + b]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1[4 + a]), __builtin_bit_cast(bf16x8, f1[b]), acc[a][b], 0, 0, 0);
    }
#endif
}

template <bool OP, int NST>
__global__ __launch_bounds__(256, 1) void igemm4p_kernel(const IgemmParams p, const void* __restrict__ g_a, unsigned a_bytes,
                                                         const void* __restrict__ g_whi, const void* __restrict__ g_wlo, unsigned w_bytes,
                                                         const float* __restrict__ g_bias, const float* __restrict__ g_addend,
                                                         const float* __restrict__ g_mask, float* __restrict__ g_out, uint2* __restrict__ g_ohi,
                                                         uint2* __restrict__ g_olo, int w_rows, int w_ld8, int splits, float* __restrict__ g_ws,
                                                         int n_items, int k_q, int k_r) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, BK = 32, NO = BK / 8, ES = 4;
  constexpr int A_U4 = 8 * BM, B_U4 = 2 * NO * BN, STAGE = A_U4 + B_U4;  // 16 KB + 16 KB
  constexpr int PER_STEP = 8;                                              // LDS-DMA instructions per wave and step: 4 gathered + 2 x 2 weight
  __shared__ __attribute__((aligned(16))) uint4 sE[2 * NO * (BM + BN)];    // the epilogue's staging buffer (never a DMA target)
  __shared__ __attribute__((aligned(16))) uint4 s0[STAGE];
  __shared__ __attribute__((aligned(16))) uint4 s1[STAGE];
  __shared__ __attribute__((aligned(16))) uint4 s2[STAGE];
  __shared__ __attribute__((aligned(16))) uint4 s3[NST == 4 ? STAGE : 1];
  auto stg = [&](auto k) __attribute__((always_inline)) -> uint4* {
    constexpr int K = decltype(k)::value;
    if constexpr (K == 0) return s0;
    else if constexpr (K == 1) return s1;
    else if constexpr (K == 2) return s2;
    else return s3;
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int il = lane & 31, h = lane >> 5;
  const int grid = (int)gridDim.x, bid = (int)blockIdx.x;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_a), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_whi), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_wlo), 0, w_bytes, 0x00020000);

  // round r of this workgroup -> its item (tile x split), or -1 past the end.  Within a round the workgroups cover `grid` consecutive
  // items and every XCD a contiguous run of them (the column tiles of one row tile share the XCD's L2).  Workgroup-uniform.
  auto item_of = [&](int round) __attribute__((always_inline)) -> int {
    const int base = round * grid, rem = n_items - base;
    if (rem <= 0 || bid >= rem) return -1;
    return base + xcd_remap(bid, rem < grid ? rem : grid);
  };

  // ---- the issue cursor: which step's tiles the next LDS-DMA batch fetches (runs NST - 1 steps ahead of the MFMAs) ----
  int iss_round = 0, iss_j = 0, iss_n = 0x40000000, iss_k0 = 0;
  int a_off[4], b_off[2];
  auto setup_issue = [&]() __attribute__((always_inline)) {
    const int it = item_of(iss_round);
    iss_j = 0;
    if (it < 0) {  // nothing left to fetch: the remaining batches are out-of-range offsets (the DMA writes zeros into a free stage)
      iss_n = 0x40000000;
      iss_k0 = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) a_off[c] = PP_BUF_OOB;
      b_off[0] = b_off[1] = PP_BUF_OOB;
      return;
    }
    // (a 32-bit division by a run-time value is VALU work even for uniform operands: without the readfirstlane the quotients --
    // and everything derived from them, the DMA's scalar offset included -- live in VGPRs and every LDS-DMA gets a waterfall loop)
    const int lb = __builtin_amdgcn_readfirstlane(it / splits), split = it - lb * splits;
    const int tile_m = __builtin_amdgcn_readfirstlane(lb / p.n_tiles_n), tile_n = lb - tile_m * p.n_tiles_n;
    iss_k0 = split * k_q + (split < k_r ? split : k_r);  // the reduction's steps dealt out evenly: all_steps = splits * k_q + k_r
    iss_n = k_q + (split < k_r ? 1 : 0);
    // gathered tile: one instruction = 8 rows x 8 pieces (the 128 contiguous bytes [hi0 lo0 .. hi3 lo3] of a row's 32-channel chunk);
    // wave w owns rows 32 w .. 32 w + 31; the lane at LDS piece position q of row r fetches piece q ^ ((r / 2) mod 8)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 32 * wave + 8 * c + (lane >> 3);
      const int piece = (lane & 7) ^ ((j >> 1) & 7);
      const RowPos r = decode_row(p, tile_m * BM + j);
      a_off[c] = r.ok ? (r.rowbase + r.ybase * r.SW + r.xbase) * p.ld_src * ES + 16 * piece : PP_BUF_OOB;
    }
    // weight tile, per plane: one instruction = 16 rows x 4 pieces (64 contiguous bytes); wave w owns rows 32 w .. 32 w + 31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int b_row = 16 * (2 * wave + i) + (lane >> 2);
      const int b_n = tile_n * BN + b_row;
      b_off[i] = b_n < w_rows ? (b_n * w_ld8 + ((lane & 3) ^ ((b_row >> 2) & 3))) * 16 : PP_BUF_OOB;
    }
  };
  auto issue_one = [&](uint4* st) __attribute__((always_inline)) {
    const int chunk = iss_k0 + iss_j;
#pragma unroll
    for (int c = 0; c < 4; ++c) dma16(rs_a, st + 8 * 32 * wave + 64 * c, a_off[c] + chunk * (BK * ES), 0);  // (out of range stays out of range)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4* const hi = st + A_U4 + 4 * 16 * (2 * wave + i);
      dma16(rs_wh, hi, b_off[i], chunk * (BK / 8 * 16));
      dma16(rs_wl, hi + NO * BN, b_off[i], chunk * (BK / 8 * 16));
    }
    if (++iss_j == iss_n) {
      ++iss_round;
      setup_issue();
    }
  };

  // ---- the consume cursor ----
  int con_round = 0, con_j = 0, con_n = 0, m0 = 0, n0 = 0, split = 0;
  auto setup_consume = [&]() __attribute__((always_inline)) -> bool {
    const int it = item_of(con_round);
    if (it < 0) return false;
    const int lb = __builtin_amdgcn_readfirstlane(it / splits);
    split = it - lb * splits;
    const int tile_m = __builtin_amdgcn_readfirstlane(lb / p.n_tiles_n);
    n0 = (lb - tile_m * p.n_tiles_n) * BN;
    m0 = tile_m * BM;
    con_n = k_q + (split < k_r ? 1 : 0);
    con_j = 0;
    return true;
  };
  if (!setup_consume()) return;  // (workgroup-uniform, before any barrier)

  // this lane's fragment offsets inside a stage (bytes; planes_fmt.h LAY 1): [half-step s][32-row block]
  unsigned fa_hi[2][2], fa_lo[2][2], fb_hi[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sa = pf_slot_a<BM, 1>(wm * 32 * TM + il + i * 32, 2 * s + h);
      fa_hi[s][i] = 16u * (unsigned)sa;
      fa_lo[s][i] = 16u * (unsigned)(sa ^ 1);
      fb_hi[s][i] = 16u * (unsigned)pf_slot_b<BN, 1>(wn * 32 * TN + il + i * 32, 2 * s + h);
    }
  auto lds_addr = [&](const uint4* q) __attribute__((always_inline)) -> unsigned {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) uint4*)q;
  };

  floatx16 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  };
  zero_acc();

  auto finish_item = [&]() __attribute__((always_inline)) {
    if (splits > 1) {  // partial sums only (slice `split` of the scratch): splitk_finish_kernel adds the slices in a fixed order
      float* slice = g_ws + (long long)split * p.M * p.ld_out;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int co = n0 + wn * 32 * TN + b * 32 + il;
            if (m < p.M && co < ((p.Nout + 3) & ~3)) slice[(long long)m * p.ld_out + co] = acc[a][b][r];
          }
        }
    } else {
      epilogue3<TM, TN, OP, 4>(p, acc, sE, m0, n0, tid, wm, wn, il, h, g_bias, g_addend, g_mask, g_out, g_ohi, g_olo);
    }
  };

  // one step on stage U: its tiles (issued NST - 1 steps ago) must have landed -- for this wave: all but the (NST - 2) younger
  // batches; for every wave: the barrier, which also says that everybody has finished reading the stage of the previous step, the
  // one the next batch goes into.  Returns true when this workgroup's last item is done.
  auto step = [&](auto u) __attribute__((always_inline)) -> bool {
    constexpr int U = decltype(u)::value;
    if constexpr (NST == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_one(stg(std::integral_constant<int, (U + NST - 1) % NST>{}));
    mma_step_asm(acc, lds_addr(stg(std::integral_constant<int, U>{})), fa_hi, fa_lo, fb_hi);
    if (++con_j == con_n) {
      finish_item();
      ++con_round;
      if (!setup_consume()) return true;
      zero_acc();
    }
    return false;
  };
  static_assert(PER_STEP * (NST - 2) == (NST == 4 ? 16 : 8), "vmcnt of step()");

  // prologue: the first NST - 1 steps
  setup_issue();
  issue_one(s0);
  issue_one(s1);
  if constexpr (NST == 4) issue_one(s2);
  for (;;) {
    if (step(std::integral_constant<int, 0>{})) break;
    if (step(std::integral_constant<int, 1>{})) break;
    if (step(std::integral_constant<int, 2>{})) break;
    if constexpr (NST == 4) {
      if (step(std::integral_constant<int, 3>{})) break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead batches of steps that never come: landed before the wave ends
}

// ============================================================================================================================
// wgrad3r_kernel: the weight gradient of a 3x3 stride-1 "same" convolution with TAP-ROW REUSE (round 4).
//   dW[ty][tx][ci][co] = sum_m x[m + (ty - 1) W + (tx - 1)][ci] * dy[m][co]        (rows of one image row only: edges are padding)
// wgrad3f stages a 32-pixel x tile and a 32-pixel dy tile per TAP: per 32-pixel step and wave 32 transposing reads + 8
// ds_write_b128 for 12 MFMAs -- LDS-bound in P16 (2 700 LDS cycles against 1 536 MFMA cycles per CU step, 42 % MFMA-busy,
// profiles/r03_pmc_stalls_both_formats.txt).  For a fixed kernel row ty the x tile of tap tx is the tile of tx = 1 shifted by one
// PIXEL, i.e. by one row of the pixel-major LDS image, and dy is the same: one staged pair (34 x rows, 32 dy rows) serves the
// three taps -- a third of the global loads and LDS stores per MFMA, and the dy fragments are read once per step.
//  * 3 taps x (64 x 64 wave tile) = 192 accumulator registers per lane; with the staging registers and fragments ~300: ONE wave per
//    SIMD (the 2-waves-per-SIMD form VERDICT r03 sketched would have 256 registers in all: 192 + fragments + addresses do not fit).
//    With one wave per SIMD nothing else covers an LDS-DMA's 60-185 cycles of issue (the lesson of igemm4p above), so the operands
//    are register-staged: plain buffer loads one step ahead, ds_write_b128 into the OTHER of two LDS buffers under the MFMAs of
//    the current step, one barrier per step.
//  * a pixel at the left (right) end of an image row has no tx = 0 (tx = 2) neighbour: the shifted tile holds the previous
//    (next) image row's pixel there.  The 8 pixels of a fragment are 8 x 16 bits of its registers: the fragment is AND-ed with a
//    mask built from two 32-bit words per step (bit p = pixel m + p is at the left / right edge; wave-uniform scalar arithmetic).
//    Rows above / below the image (ty != 1) are out-of-range offsets of the staged x row: zeros.
//  * grid = (kernel row ty) x (128-channel cin tiles) x (128-channel cout tiles) x row splits; f32 atomics into dW like wgrad3f.
//  * SP: the reduction walks the listed 32-row blocks of dy (pp_row_block_list), like wgrad3f<.., SP>.
// Conditions (host-checked): plane-stored operands, 3x3 / stride 1 / pad 1 on an unchanged grid, cin % 128 == 0, every level's row
// count a multiple of 32 (a step never straddles two levels), narrowest level >= 12 pixels wide.
__device__ __forceinline__ void and4(uint4& v, const unsigned (&m)[4]) {
  v.x &= m[0]; v.y &= m[1]; v.z &= m[2]; v.w &= m[3];
}

template <bool SP>
__global__ __launch_bounds__(256, 1) void wgrad3r_kernel(const Wgrad3Params p, const void* __restrict__ g_x0, const void* __restrict__ g_x1,
                                                         unsigned x_bytes, const void* __restrict__ g_d0, const void* __restrict__ g_d1,
                                                         unsigned d_bytes, float* __restrict__ g_dw, float* __restrict__ g_dbias,
                                                         const int* __restrict__ g_list) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, BK = 32;
  // x rows per buffer: 34 are read; rows 32 .. 63 exist so that EVERY thread can store its "extra row" registers (zeros but for the
  // threads of rows 0, 1) without a branch -- the loop body stays one basic block that the scheduler can interleave under the MFMAs
  constexpr int XR = 2 * BK;
  constexpr int PA = BM + 32, PB = BN + 32;  // LDS pitches in 16-bit elements (row + 64 bytes: conflict-free transposing reads)
  constexpr int XB = XR * PA, GB = BK * PB;  // elements per plane and buffer
  constexpr int BUF = 2 * XB + 2 * GB;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];  // 2 x 61 440 bytes

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float inv_g = p.inv_scale ? *p.inv_scale : 1.f;
  const int wm = wave >> 1, wn = wave & 1;
  int b = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = b % p.n_tiles_n;
  b /= p.n_tiles_n;
  const int tile_k = b % p.n_tiles_k;
  const int split = b / p.n_tiles_k;
  const int ty = tile_k / p.k_tiles_per_tap;
  const int ci0 = (tile_k - ty * p.k_tiles_per_tap) * BM;
  const int n0 = tile_n * BN;
  // (dense: the splits cut the PADDED position space, see below; rows_per_split counts positions)
  int m_begin = split * p.rows_per_split;
  int m_end = min(SP ? p.M : p.Mp, m_begin + p.rows_per_split);
  int n_steps = (m_end - m_begin + BK - 1) / BK;
  int s_idx = 0, s_end = 0;
  const int blk_past = (p.M + BK - 1) / BK;
  if (SP) {  // (see wgrad3f_kernel: a short list is shared by fewer splits)
    const int n_act = g_list[0];
    int s_eff = (n_act + p.sp_min_steps - 1) / p.sp_min_steps;
    s_eff = s_eff < 1 ? 1 : (s_eff > p.splits ? p.splits : s_eff);
    if (split >= s_eff) return;
    s_idx = (int)((long long)n_act * split / s_eff);
    s_end = (int)((long long)n_act * (split + 1) / s_eff);
    n_steps = s_end - s_idx;
    m_begin = (n_steps > 0 ? g_list[1 + s_idx] : blk_past) * BK;
    m_end = p.M;
  }

  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_x0), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_x1), 0, x_bytes - 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_d0), 0, d_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_d1), 0, d_bytes - 16, 0x00020000);

  // staging: 8 threads per row, thread q8 moves the 8-channel groups q8 and q8 + 8 (16 bytes of hi + 16 bytes of lo each)
  const int prow = tid >> 3, q8 = tid & 7;
  uint4 rx[2], rxl[2], re[2], rel[2], rd[2], rdl[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) re[j] = rel[j] = make_uint4(0u, 0u, 0u, 0u);
  float4 bsum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = (g_dbias != nullptr) && (tile_k == 0);

  // ---- the reduction space ----
  // SP (listed blocks): the pixels themselves; the fragments of the edge taps are masked (edge words, below).
  // Dense (PADK): the pixels with ONE PAD POSITION behind every image row (row length W + 1): a pad is an out-of-range offset for
  // both operands -- zeros in the x tile and in the dy tile --, so the neighbour of a row's first (last) pixel at tap tx = 0 (2)
  // IS a zero and no fragment needs masking (150 VALU per step less: with ONE wave per SIMD the step is bound by the instructions
  // that wave has to issue, not by the matrix pipe); costs 1 / W more steps (1.3 % at 80 pixels, 5 % at 20).
  // Thread `prow` stages position q = m_cur + prow: dy row prow and x row prow + 1; the threads of rows 0 / 1 also stage x row 0
  // (position m_cur - 1) / x row 33 (position m_cur + 32).
  constexpr bool PADK = !SP;
  int d_OW, d_OH, d_xx, d_y, d_img;
  bool d_ok;
  auto decode = [&](int q) __attribute__((always_inline)) {  // branch-free (fixed trip count, selects): part of the loop's one basic block
    int qbeg = PADK ? p.pos_begin[0] : p.seg[0].row_begin, rbeg = p.seg[0].row_begin;
    d_OH = p.seg[0].OH; d_OW = p.seg[0].OW;
#pragma unroll
    for (int s = 1; s < PP_MAX_SEG; ++s) {
      const int sb = PADK ? p.pos_begin[s] : p.seg[s].row_begin;
      const bool in = s < p.n_seg && q >= sb;
      qbeg = in ? sb : qbeg;
      rbeg = in ? p.seg[s].row_begin : rbeg;
      d_OH = in ? p.seg[s].OH : d_OH;
      d_OW = in ? p.seg[s].OW : d_OW;
    }
    const int q_end = PADK ? p.Mp : p.M;
    const int qc = q < 0 ? 0 : (q < q_end ? q : q_end - 1);
    const int w1 = d_OW + (PADK ? 1 : 0), hw1 = d_OH * w1;
    int rem;
    const int n = div_small(qc - qbeg, hw1, __frcp_rn((float)hw1), &rem);
    d_y = div_small(rem, w1, __frcp_rn((float)w1), &d_xx);
    d_img = rbeg + n * d_OH * d_OW;  // first pixel of the image ("same" geometry: x and dy share the row space)
    d_ok = q >= 0 && q < q_end && d_xx < d_OW;
  };
  int q_next = blk_past;
  if (SP) q_next = (s_idx + 1 < s_end) ? g_list[1 + s_idx + 1] : blk_past;
  int m_cur = m_begin;  // first position of the step being PREPARED (uniform)

  // the edge words of the step at m_cur (SP; wave-uniform): bit j of .x / .y = pixel m_cur + j sits at x == 0 / x == W - 1.
  // Lane 0 of wave w has decoded pixel m_cur + 8 w (same level: a listed block never straddles two levels, host-checked)
  auto edge_bits = [&]() __attribute__((always_inline)) -> uint2 {
    const int W = __builtin_amdgcn_readfirstlane(d_OW);
    int x0 = __builtin_amdgcn_readfirstlane(d_xx) - 8 * wave;
    x0 += x0 < 0 ? W : 0;
    x0 += x0 < 0 ? W : 0;
    x0 += x0 < 0 ? W : 0;
    unsigned l = 0, r = 0;
    int pl = x0 == 0 ? 0 : W - x0, pr = W - 1 - x0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      l |= pl < 32 ? (1u << pl) : 0u;
      r |= pr < 32 ? (1u << pr) : 0u;
      pl += W;
      pr += W;
    }
    return make_uint2(l, r);
  };

  // Software pipeline over THREE steps: prep() -- positions, offsets, edge words of step s + 2 (VALU / SALU only) -- runs in the
  // shadow of the MFMAs of step s; issue() -- the 12 buffer loads of step s + 1, offsets already in registers -- sits at the very
  // top of the step (pinned by a sched_barrier: left alone the scheduler sinks the loads to the ds_writes that need them, and
  // every step pays a global-load latency).
  int o_x = PP_BUF_OOB, o_e = PP_BUF_OOB, o_d[2] = {PP_BUF_OOB, PP_BUF_OOB};
  auto x_off = [&]() __attribute__((always_inline)) -> int {  // byte offset of the x pixel of the decoded position at kernel row ty, or out of range
    const int sy = d_y + ty - 1;
    const int o = ((d_img + sy * d_OW + d_xx) * p.ld_src + ci0) * 4 + 32 * q8;
    return o | ((d_ok && (unsigned)sy < (unsigned)d_OH) ? 0 : PP_BUF_OOB);  // (bit 31 = out of range; no select of the offset: see prep())
  };
  auto prep = [&]() __attribute__((always_inline)) -> uint2 {
    const int q = m_cur + prow;
    decode(q);
    uint2 eb = make_uint2(0u, 0u);
    if (SP) eb = edge_bits();
    o_x = x_off();
    const int dyo = ((d_img + d_y * d_OW + d_xx) * p.ld_dy + n0) * 4 + 32 * q8;
    const bool in_rng = d_ok && q < m_end;
#pragma unroll
    for (int j = 0; j < 2; ++j) o_d[j] = dyo | ((in_rng && (n0 + 8 * (q8 + 8 * j) < p.ld_dy)) ? 0 : PP_BUF_OOB);
    decode(prow == 0 ? m_cur - 1 : m_cur + BK);
    // (the other threads fetch nothing -- an offset with bit 31 set is out of range: zeros, stored into rows that are never read.  The
    // OR keeps the decode above unconditional: behind a select the compiler wraps it in an exec-masked branch, which cuts the step's
    // basic block in two and leaves this whole address computation OUTSIDE the MFMAs' shadow)
    o_e = x_off() | (prow < 2 ? 0 : PP_BUF_OOB);
    if (SP) {
      ++s_idx;
      m_cur = q_next * BK;
      q_next = (s_idx + 1 < s_end) ? g_list[1 + s_idx + 1] : blk_past;
    } else {
      m_cur += BK;
    }
    return eb;
  };
  auto issue = [&]() __attribute__((always_inline)) {
#if defined(PP_ABL) && PP_ABL == 1  // ablation: no global loads
    return;
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rx[j] = buf_load16(rs_x0, o_x, 256 * j);
      rxl[j] = buf_load16(rs_x1, o_x, 256 * j);
      rd[j] = buf_load16(rs_d0, o_d[j], 256 * j);
      rdl[j] = buf_load16(rs_d1, o_d[j], 256 * j);
      re[j] = buf_load16(rs_x0, o_e, 256 * j);
      rel[j] = buf_load16(rs_x1, o_e, 256 * j);
    }
  };
  auto bias_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned hw4[4] = {rd[j].x, rd[j].y, rd[j].z, rd[j].w}, lw4[4] = {rdl[j].x, rdl[j].y, rdl[j].z, rdl[j].w};
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) fmt_value2(hw4[e], lw4[e], &v[2 * e], &v[2 * e + 1]);
      bsum[2 * j].x += v[0]; bsum[2 * j].y += v[1]; bsum[2 * j].z += v[2]; bsum[2 * j].w += v[3];
      bsum[2 * j + 1].x += v[4]; bsum[2 * j + 1].y += v[5]; bsum[2 * j + 1].z += v[6]; bsum[2 * j + 1].w += v[7];
    }
  };
  auto store_step = [&](unsigned short* buf) __attribute__((always_inline)) {
#if defined(PP_ABL) && PP_ABL == 3  // ablation: no LDS stores (keep the loaded registers alive)
    {
      unsigned k = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) k ^= rx[j].x ^ rx[j].w ^ rxl[j].x ^ rxl[j].w ^ rd[j].x ^ rd[j].w ^ rdl[j].x ^ rdl[j].w ^ re[j].x ^ re[j].w ^ rel[j].x ^ rel[j].w;
      asm volatile("" ::"v"(k));
    }
    return;
#endif
    unsigned short* const Xhi = buf;
    unsigned short* const Xlo = buf + XB;
    unsigned short* const Ghi = buf + 2 * XB;
    unsigned short* const Glo = Ghi + GB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 8 * (q8 + 8 * j);
      *reinterpret_cast<uint4*>(Xhi + (prow + 1) * PA + col) = rx[j];
      *reinterpret_cast<uint4*>(Xlo + (prow + 1) * PA + col) = rxl[j];
      uint4 gl = rdl[j];
#if PP_FMT == 1
      // dy's lo units with their bytes swapped once, here, instead of in every fragment of every tap (see wgrad_step)
      gl.x = __builtin_amdgcn_perm(gl.x, gl.x, 0x02030001u); gl.y = __builtin_amdgcn_perm(gl.y, gl.y, 0x02030001u);
      gl.z = __builtin_amdgcn_perm(gl.z, gl.z, 0x02030001u); gl.w = __builtin_amdgcn_perm(gl.w, gl.w, 0x02030001u);
#endif
      *reinterpret_cast<uint4*>(Ghi + prow * PB + col) = rd[j];
      *reinterpret_cast<uint4*>(Glo + prow * PB + col) = gl;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 8 * (q8 + 8 * j);
      const int er = prow == 0 ? 0 : BK + prow;  // thread row 0 -> x row 0, thread row 1 -> x row 33, the others -> rows nobody reads
      *reinterpret_cast<uint4*>(Xhi + er * PA + col) = re[j];
      *reinterpret_cast<uint4*>(Xlo + er * PA + col) = rel[j];
    }
  };

  floatx16 acc[3][TM][TN];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int c = 0; c < TN; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][a][c][r] = 0.f;

  // transposed-read lane roles (16-lane groups): group g -> columns 16 (g & 1) .., pixel half hh = g >> 1 (see wgrad3f_kernel)
  const int grp = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
  const int cbase = 16 * (grp & 1), hh = grp >> 1;
  const int il = lane & 31, h = lane >> 5;
  const int xcol0 = wm * 32 * TM + cbase + 4 * gp, gcol0 = wn * 32 * TN + cbase + 4 * gp;

  // 8 pixels of a fragment (bits 8 hh + 16 s .. + 7 of an edge word, 1 = at the edge) -> the four dwords that keep the others
  auto frag_mask = [&](unsigned bits, int s, unsigned (&m)[4]) __attribute__((always_inline)) {
    const unsigned b8 = ~(bits >> (16 * s + 8 * hh));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)b8, 2 * i, 1), hi = (unsigned)__builtin_amdgcn_sbfe((int)b8, 2 * i + 1, 1);
      m[i] = (lo & 0xffffu) | (hi & 0xffff0000u);
    }
  };

  auto compute = [&](const unsigned short* buf, uint2 eb) __attribute__((always_inline)) {
#if defined(PP_ABL) && PP_ABL == 2  // ablation: no fragment reads, no MFMAs
    return;
#endif
    const unsigned short* const Xhi = buf;
    const unsigned short* const Xlo = buf + XB;
    const unsigned short* const Ghi = buf + 2 * XB;
    const unsigned short* const Glo = Ghi + GB;
    unsigned ml[2][4], mr[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_mask(eb.x, s, ml[s]);
      frag_mask(eb.y, s, mr[s]);
    }
    // x row of dy pixel k at tap tx: k + tx (LDS x row 0 = pixel m - 1)
    auto xfrag = [&](const unsigned short* X, int s, int a, int tx) __attribute__((always_inline)) -> uint4 {
      const int row0 = 16 * s + 8 * hh + gq + tx, col = xcol0 + a * 32;
      const bf16x8 t = tr_frag(X, row0 * PA + col, (row0 + 4) * PA + col);
      uint4 u = *reinterpret_cast<const uint4*>(&t);
      if (SP && tx == 0) and4(u, ml[s]);
      if (SP && tx == 2) and4(u, mr[s]);
      return u;
    };
    auto gfrag = [&](const unsigned short* G, int s, int c) __attribute__((always_inline)) -> uint4 {
      const int row0 = 16 * s + 8 * hh + gq, col = gcol0 + c * 32;
      const bf16x8 t = tr_frag(G, row0 * PB + col, (row0 + 4) * PB + col);
      return *reinterpret_cast<const uint4*>(&t);
    };
#if PP_FMT == 1
    intx8 gq8[TN];
    uint4 gh[2][TN];
#pragma unroll
    for (int c = 0; c < TN; ++c) {
      const uint4 u0 = gfrag(Glo, 0, c), u1 = gfrag(Glo, 1, c);
      gq8[c][0] = (int)u0.x; gq8[c][1] = (int)u0.y; gq8[c][2] = (int)u0.z; gq8[c][3] = (int)u0.w;
      gq8[c][4] = (int)u1.x; gq8[c][5] = (int)u1.y; gq8[c][6] = (int)u1.z; gq8[c][7] = (int)u1.w;
      gh[0][c] = gfrag(Ghi, 0, c);
      gh[1][c] = gfrag(Ghi, 1, c);
    }
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const uint4 u0 = xfrag(Xlo, 0, a, tx), u1 = xfrag(Xlo, 1, a, tx);
        intx8 xq8;
        xq8[0] = (int)u0.x; xq8[1] = (int)u0.y; xq8[2] = (int)u0.z; xq8[3] = (int)u0.w;
        xq8[4] = (int)u1.x; xq8[5] = (int)u1.y; xq8[6] = (int)u1.z; xq8[7] = (int)u1.w;
#pragma unroll
        for (int c = 0; c < TN; ++c)
          acc[tx][a][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xq8, gq8[c], acc[tx][a][c], 1, 1, 0, P16_SCALES, 1, P16_SCALES);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const uint4 u = xfrag(Xhi, s, a, tx);
          const halfx8 xh = *reinterpret_cast<const halfx8*>(&u);
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[tx][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, *reinterpret_cast<const halfx8*>(&gh[s][c]), acc[tx][a][c], 0, 0, 0);
        }
    }
#else
    uint4 gh[2][TN], gl[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        gh[s][c] = gfrag(Ghi, s, c);
        gl[s][c] = gfrag(Glo, s, c);
      }
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const uint4 uh = xfrag(Xhi, s, a, tx), ul = xfrag(Xlo, s, a, tx);
          const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&uh), xl = *reinterpret_cast<const bf16x8*>(&ul);
#pragma unroll
          for (int c = 0; c < TN; ++c) {
            acc[tx][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, *reinterpret_cast<const bf16x8*>(&gh[s][c]), acc[tx][a][c], 0, 0, 0);
            acc[tx][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, *reinterpret_cast<const bf16x8*>(&gl[s][c]), acc[tx][a][c], 0, 0, 0);
            acc[tx][a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, *reinterpret_cast<const bf16x8*>(&gh[s][c]), acc[tx][a][c], 0, 0, 0);
          }
        }
#endif
  };

  auto k_loop = [&](auto with_bias) __attribute__((always_inline)) {
    constexpr bool WB = decltype(with_bias)::value;
    uint2 eb0 = prep();  // step 0
    issue();
    uint2 eb1 = prep();  // step 1
    if (WB) bias_step();
    store_step(smem);
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
      issue();  // step + 1 (past the last step every dy row is out of range: zeros, stored and never multiplied)
      __builtin_amdgcn_sched_barrier(0);
      const uint2 eb2 = prep();  // step + 2
      compute(smem + (step & 1) * BUF, eb0);
      // the order inside the step: left alone the scheduler emits all of prep() (~200 VALU / SALU) and THEN the MFMAs; pipeline
      // them instead -- the first fragments, then per MFMA two transposing reads and six scalar / vector instructions
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int i = 0; i < 36; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (WB) bias_step();
      store_step(smem + ((step + 1) & 1) * BUF);
      __syncthreads();
      eb0 = eb1;
      eb1 = eb2;
    }
  };
  if (do_bias) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  // reduction over the row splits: f32 atomics into dW (fire and forget; see wgrad3f_kernel)
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long row = (long long)((ty * 3 + tx) * p.Cin + ci) * p.ld_w;
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const int co = n0 + wn * 32 * TN + c * 32 + il;
          if (co < p.Cout) atomicAdd(g_dw + row + co, acc[tx][a][c][r] * inv_g);
        }
      }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem);  // [32][BN] floats
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j)) = bsum[2 * j];
      *reinterpret_cast<float4*>(red + prow * BN + 8 * (q8 + 8 * j) + 4) = bsum[2 * j + 1];
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


// ============================================================================================================================
// wgrad3w_kernel: wgrad3r's tiles with the two halves of a step on DIFFERENT WAVES (round 4, second attempt; dense reduction only).
// What wgrad3r showed: with ONE wave per SIMD its step is the SUM of a global-load latency (one step of look-ahead), ~200 address
// instructions and the MFMA block.  Here a workgroup is 8 waves = two per SIMD, 256 registers each:
//  * waves 0-3 ("consumers") hold the 192 accumulators and do nothing but transposing fragment reads and MFMAs -- 18 units per
//    step, the reads of unit U + 2 in front of the MFMAs of unit U (alone they run at the MFMA rate: 0.75 us per step);
//  * waves 4-7 ("producers") have no accumulators.  Waves 4, 5 stage the EVEN steps, waves 6, 7 the ODD ones: a pair stores its
//    step's tiles in one phase (between two barriers) and issues the loads of its next step -- two phases ahead -- right behind the
//    stores, so every load has two full steps to land with ONE register set per wave and plain, compiler-counted waits (a first
//    version kept two sets per wave in flight behind inline-assembly loads: hipcc's own bookkeeping merges the unrolled loop's
//    halves conservatively, and with hand-counted waits the loop-carried, still-in-flight registers can be COPIED at the back edge
//    -- a hang that came and went with unrelated changes).  In its other phase a pair walks its positions: +64 per step, a
//    decode from scratch only when the walk leaves its image.
//  * one s_barrier per step, two LDS buffers: in phase k (the consumers read buffer (k - 1) & 1) step k is stored into buffer k & 1.
// Tile decode, reduction space (positions with a pad behind every image row), f32 atomics and bias sums are wgrad3r's.
// What it reaches (profiles/r04_wgrad_producer_consumer_waves.txt): 5-19 % under wgrad3f per dense P16 launch, +4 % on the
// dense-backward step -- and NOT the 1.5x the split was built for: the consumers alone run a step in 0.75-0.9 us, the producers
// alone in 0.57 us, both together in the SUM of the two.  Staging a step's 42 KB costs each SIMD ~1 100 cycles in which its
// consumer wave issues nothing, whichever wave stages and however (register staging as here; LDS-DMA into a ring of three
// buffers with a table of row offsets: built, parity-green, the same sum; stores spread over the phase: slower).  On this CU the
// bytes staged per MFMA set the time, not who stages them.
__global__ __launch_bounds__(512, 1) void wgrad3w_kernel(const Wgrad3Params p, const void* __restrict__ g_x0, const void* __restrict__ g_x1,
                                                         unsigned x_bytes, const void* __restrict__ g_d0, const void* __restrict__ g_d1,
                                                         unsigned d_bytes, float* __restrict__ g_dw, float* __restrict__ g_dbias) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, BK = 32;
  constexpr int XR = BK + 2;                 // x rows per buffer: positions m - 1 .. m + 32
  constexpr int PA = BM + 32, PB = BN + 32;  // LDS pitches in 16-bit elements (row + 64 bytes: conflict-free transposing reads)
  constexpr int XB = XR * PA, GB = BK * PB;  // elements per plane and buffer
  constexpr int BUF = 2 * XB + 2 * GB;       // 42 240 bytes
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int rw = wave & 3;  // index of the wave within its role
  const float inv_g = p.inv_scale ? *p.inv_scale : 1.f;
  const int wm = rw >> 1, wn = rw & 1;
  int b = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = b % p.n_tiles_n;
  b /= p.n_tiles_n;
  const int tile_k = b % p.n_tiles_k;
  const int split = b / p.n_tiles_k;
  const int ty = tile_k / p.k_tiles_per_tap;
  const int ci0 = (tile_k - ty * p.k_tiles_per_tap) * BM;
  const int n0 = tile_n * BN;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.Mp, m_begin + p.rows_per_split);
  const int n_steps = (m_end - m_begin + BK - 1) / BK;
  const bool do_bias = (g_dbias != nullptr) && (tile_k == 0);
  // producers: 128 threads per pair, 8 per row; thread row `prow` stages positions m + prow and m + prow + 16
  const int ptid = tid & 127, prow = ptid >> 3, q8 = ptid & 7;
  float4 bsum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (producer) {
    const int pair = rw >> 1;  // 0: even steps, 1: odd steps
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_x0), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_x1), 0, x_bytes - 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_d0), 0, d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_d1), 0, d_bytes - 16, 0x00020000);
    uint4 rx[2][2], rxl[2][2], rd[2][2], rdl[2][2], re[2], rel[2];  // [row of the thread][channel group]; re: the thread's extra x row
    int d_OW, d_OH, d_xx, d_y, d_img;
    bool d_ok;
    auto decode = [&](int q) __attribute__((always_inline)) {  // position -> level, image, (y, x); x == W is the pad
      int qbeg = p.pos_begin[0], rbeg = p.seg[0].row_begin;
      d_OH = p.seg[0].OH; d_OW = p.seg[0].OW;
#pragma unroll
      for (int s = 1; s < PP_MAX_SEG; ++s) {
        const bool in = s < p.n_seg && q >= p.pos_begin[s];
        qbeg = in ? p.pos_begin[s] : qbeg;
        rbeg = in ? p.seg[s].row_begin : rbeg;
        d_OH = in ? p.seg[s].OH : d_OH;
        d_OW = in ? p.seg[s].OW : d_OW;
      }
      const int qc = q < 0 ? 0 : (q < p.Mp ? q : p.Mp - 1);
      const int w1 = d_OW + 1, hw1 = d_OH * w1;
      int rem;
      const int n = div_small(qc - qbeg, hw1, __frcp_rn((float)hw1), &rem);
      d_y = div_small(rem, w1, __frcp_rn((float)w1), &d_xx);
      d_img = rbeg + n * d_OH * d_OW;  // first pixel of the image ("same" geometry: x and dy share the row space)
      d_ok = q >= 0 && q < p.Mp && d_xx < d_OW;
    };
    auto x_off = [&]() __attribute__((always_inline)) -> int {  // byte offset of the decoded position's x pixel at kernel row ty, or out of range
      const int sy = d_y + ty - 1;
      const int o = ((d_img + sy * d_OW + d_xx) * p.ld_src + ci0) * 4 + 32 * q8;
      return o | ((d_ok && (unsigned)sy < (unsigned)d_OH) ? 0 : PP_BUF_OOB);
    };
    // the walk: this pair's next step is 2 BK positions on
    int w_q[2], w_xx[2], w_y[2], w_img[2], w_W[2], w_H[2];
    bool w_ok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      w_q[r] = m_begin + (pair - 2) * BK + prow + 16 * r;
      w_xx[r] = 0; w_y[r] = 0; w_img[r] = 0; w_W[r] = 1 << 29; w_H[r] = 0;
      w_ok[r] = false;
    }
    int o_x[2] = {PP_BUF_OOB, PP_BUF_OOB}, o_d[2][2] = {{PP_BUF_OOB, PP_BUF_OOB}, {PP_BUF_OOB, PP_BUF_OOB}}, o_e = PP_BUF_OOB;
    const int cofs = ci0 * 4 + 32 * q8;
    auto prep = [&]() __attribute__((always_inline)) {
      o_e = PP_BUF_OOB;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        w_q[r] += 2 * BK;
        w_xx[r] += 2 * BK;
        while (w_xx[r] > w_W[r]) {  // (row length W + 1: positions 0 .. W, W = the pad)
          w_xx[r] -= w_W[r] + 1;
          ++w_y[r];
        }
        if (w_y[r] >= w_H[r] || w_q[r] >= p.Mp) {
          decode(w_q[r]);
          w_xx[r] = d_xx; w_y[r] = d_y; w_img[r] = d_img; w_W[r] = d_OW; w_H[r] = d_OH;
          w_ok[r] = d_ok;
        } else {
          w_ok[r] = w_xx[r] < w_W[r];
        }
        const int sy = w_y[r] + ty - 1;
        const bool vy = (unsigned)sy < (unsigned)w_H[r];
        const int lin = w_img[r] + w_y[r] * w_W[r] + w_xx[r];  // (at a pad: the first pixel of the next row -- never read)
        const int linx = lin + (ty - 1) * w_W[r];
        o_x[r] = (linx * p.ld_src * 4 + cofs) | ((w_ok[r] && vy) ? 0 : PP_BUF_OOB);
        const int dyo = (lin * p.ld_dy + n0) * 4 + 32 * q8;
        const bool in_rng = w_ok[r] && w_q[r] < m_end;
#pragma unroll
        for (int j = 0; j < 2; ++j) o_d[r][j] = dyo | ((in_rng && (n0 + 8 * (q8 + 8 * j) < p.ld_dy)) ? 0 : PP_BUF_OOB);
        // the two extra x rows are neighbours of positions this thread already has: x row 0 (position m - 1) belongs to thread
        // row 0, x row 33 (position m + 32) to the second row of thread row 15
        if (r == 0 && prow == 0) {  // position q - 1: the previous pixel of this image row, or the pad before it (zeros)
          o_e = ((linx - 1) * p.ld_src * 4 + cofs) | ((w_xx[r] > 0 && vy && w_q[r] < p.Mp) ? 0 : PP_BUF_OOB);
        } else if (r == 1 && prow == 15) {  // position q + 1
          if (w_q[r] + 1 >= p.Mp) {
          } else if (w_xx[r] < w_W[r]) {  // the next pixel of this row, or its pad
            o_e = ((linx + 1) * p.ld_src * 4 + cofs) | ((w_xx[r] + 1 < w_W[r] && vy) ? 0 : PP_BUF_OOB);
          } else if (w_y[r] + 1 < w_H[r]) {  // q is a pad: the first pixel of the next image row
            const int sy2 = w_y[r] + ty;
            o_e = ((w_img[r] + sy2 * w_W[r]) * p.ld_src * 4 + cofs) | (((unsigned)sy2 < (unsigned)w_H[r]) ? 0 : PP_BUF_OOB);
          } else {  // the first pixel of the next image (or level)
            decode(w_q[r] + 1);
            o_e = x_off();
          }
        }
      }
    };
    auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          rx[r][j] = buf_load16(rs_x0, o_x[r], 256 * j);
          rxl[r][j] = buf_load16(rs_x1, o_x[r], 256 * j);
          rd[r][j] = buf_load16(rs_d0, o_d[r][j], 256 * j);
          rdl[r][j] = buf_load16(rs_d1, o_d[r][j], 256 * j);
        }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        re[j] = buf_load16(rs_x0, o_e, 256 * j);
        rel[j] = buf_load16(rs_x1, o_e, 256 * j);
      }
    };
    unsigned short* const buf = smem + pair * BUF;  // (an even step's tiles live in buffer 0, an odd step's in buffer 1)
    unsigned short* const Xhi = buf;
    unsigned short* const Xlo = buf + XB;
    unsigned short* const Ghi = buf + 2 * XB;
    unsigned short* const Glo = Ghi + GB;
    auto store_step = [&]() __attribute__((always_inline)) {
      if (do_bias) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const unsigned hw4[4] = {rd[r][j].x, rd[r][j].y, rd[r][j].z, rd[r][j].w}, lw4[4] = {rdl[r][j].x, rdl[r][j].y, rdl[r][j].z, rdl[r][j].w};
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) fmt_value2(hw4[e], lw4[e], &v[2 * e], &v[2 * e + 1]);
            bsum[2 * j].x += v[0]; bsum[2 * j].y += v[1]; bsum[2 * j].z += v[2]; bsum[2 * j].w += v[3];
            bsum[2 * j + 1].x += v[4]; bsum[2 * j + 1].y += v[5]; bsum[2 * j + 1].z += v[6]; bsum[2 * j + 1].w += v[7];
          }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = 8 * (q8 + 8 * j), row = prow + 16 * r;
          *reinterpret_cast<uint4*>(Xhi + (row + 1) * PA + col) = rx[r][j];
          *reinterpret_cast<uint4*>(Xlo + (row + 1) * PA + col) = rxl[r][j];
          uint4 gl = rdl[r][j];
#if PP_FMT == 1
          // dy's lo units with their bytes swapped once, here, instead of in every fragment of every tap (see wgrad_step)
          gl.x = __builtin_amdgcn_perm(gl.x, gl.x, 0x02030001u); gl.y = __builtin_amdgcn_perm(gl.y, gl.y, 0x02030001u);
          gl.z = __builtin_amdgcn_perm(gl.z, gl.z, 0x02030001u); gl.w = __builtin_amdgcn_perm(gl.w, gl.w, 0x02030001u);
#endif
          *reinterpret_cast<uint4*>(Ghi + row * PB + col) = rd[r][j];
          *reinterpret_cast<uint4*>(Glo + row * PB + col) = gl;
        }
      if (prow == 0 || prow == 15) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = 8 * (q8 + 8 * j);
          const int er = prow == 0 ? 0 : BK + 1;  // x row 0 = position m - 1, x row 33 = position m + 32
          *reinterpret_cast<uint4*>(Xhi + er * PA + col) = re[j];
          *reinterpret_cast<uint4*>(Xlo + er * PA + col) = rel[j];
        }
      }
    };
    // phase 0 (before the first barrier): both pairs issue their first step; pair 0 also stores step 0 and issues step 2
    prep();
    issue();
    prep();
    if (pair == 0) {
      store_step();
      issue();
    }
    barrier();
    // phase ph (the consumers read buffer (ph - 1) & 1): the pair of step ph stores it into buffer ph & 1 and issues step ph + 2
    // (offsets ready since its last idle phase); the other pair has the phase for its address walk (the step after the one in flight)
    for (int ph = 1; ph <= n_steps; ++ph) {
      if ((ph & 1) == pair) {
        store_step();
        issue();
      } else {
        prep();
      }
      barrier();
    }
  } else {
    floatx16 acc[3][TM][TN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][a][c][r] = 0.f;
    // transposed-read lane roles (16-lane groups): group g -> columns 16 (g & 1) .., pixel half hh = g >> 1 (see wgrad3f_kernel)
    const int grp = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    const int cbase = 16 * (grp & 1), hh = grp >> 1;
    const int il = lane & 31, h = lane >> 5;
    const int xcol0 = wm * 32 * TM + cbase + 4 * gp, gcol0 = wn * 32 * TN + cbase + 4 * gp;
    auto compute = [&](const unsigned short* buf) __attribute__((always_inline)) {
      // (one lane pointer per operand and step, everything else a compile-time element offset: the reads take it as their immediate.
      // With the offsets spelled as one integer expression hipcc kept six address registers -- two more than the kernel has)
      const unsigned short* const xl = buf + (8 * hh + gq) * PA + xcol0;
      const unsigned short* const gl_ = buf + (8 * hh + gq) * PB + gcol0;
      constexpr int Xhi = 0, Xlo = XB, Ghi = 2 * XB, Glo = 2 * XB + GB;  // plane offsets in elements
      // x row of dy pixel k at tap tx: k + tx (LDS x row 0 = position m - 1)
      auto xfrag = [&](int plane, int s, int a, int tx) __attribute__((always_inline)) -> uint4 {
        const int c0 = plane + (16 * s + tx) * PA + a * 32;
        const bf16x8 t = tr_frag(xl, c0, c0 + 4 * PA);
        return *reinterpret_cast<const uint4*>(&t);
      };
      auto gfrag = [&](int plane, int s, int c) __attribute__((always_inline)) -> uint4 {
        const int c0 = plane + 16 * s * PB + c * 32;
        const bf16x8 t = tr_frag(gl_, c0, c0 + 4 * PB);
        return *reinterpret_cast<const uint4*>(&t);
      };
      // Left alone hipcc emits [reads of a unit; lgkmcnt(0); its MFMAs]: an LDS latency per pair of MFMAs.  Here the reads of unit
      // U + LA are issued in front of the MFMAs of unit U, unit boundaries are scheduling barriers, and the compiler's own (counted)
      // lgkmcnt waits do the rest.
      constexpr int LA = PP_W3W_LA;
#if PP_FMT == 1
      // 18 units per step: the six (tap, a) pairs of the e5m2 planes (4 transposing reads -> 2 scaled MFMAs), then of the half planes
      // of s = 0, then of s = 1 (2 reads -> 2 MFMAs) -- dy's fragments of the three phases are then live one phase at a time (16 + 8
      // registers instead of 32; with 192 accumulators that is the difference between spilling and not)
      constexpr int NU = 18;
      intx8 gq8[TN];
      uint4 gh[2][TN];
      auto rd_g = [&](int ph) __attribute__((always_inline)) {
        if (ph == 0) {
#pragma unroll
          for (int c = 0; c < TN; ++c) {
            const uint4 u0 = gfrag(Glo, 0, c), u1 = gfrag(Glo, 1, c);
            gq8[c][0] = (int)u0.x; gq8[c][1] = (int)u0.y; gq8[c][2] = (int)u0.z; gq8[c][3] = (int)u0.w;
            gq8[c][4] = (int)u1.x; gq8[c][5] = (int)u1.y; gq8[c][6] = (int)u1.z; gq8[c][7] = (int)u1.w;
          }
        } else {
#pragma unroll
          for (int c = 0; c < TN; ++c) gh[ph - 1][c] = gfrag(Ghi, ph - 1, c);
        }
      };
      uint4 F[NU][2];
      auto rd = [&](int U) __attribute__((always_inline)) {
        const int ph = U / 6, tx = (U % 6) >> 1, a_ = U & 1;
        if (ph == 0) {
          F[U][0] = xfrag(Xlo, 0, a_, tx);
          F[U][1] = xfrag(Xlo, 1, a_, tx);
        } else {
          F[U][0] = xfrag(Xhi, ph - 1, a_, tx);
        }
      };
      auto mm = [&](int U) __attribute__((always_inline)) {
        const int ph = U / 6, tx = (U % 6) >> 1, a_ = U & 1;
        if (ph == 0) {
          intx8 xq8;
          xq8[0] = (int)F[U][0].x; xq8[1] = (int)F[U][0].y; xq8[2] = (int)F[U][0].z; xq8[3] = (int)F[U][0].w;
          xq8[4] = (int)F[U][1].x; xq8[5] = (int)F[U][1].y; xq8[6] = (int)F[U][1].z; xq8[7] = (int)F[U][1].w;
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[tx][a_][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xq8, gq8[c], acc[tx][a_][c], 1, 1, 0, P16_SCALES, 1, P16_SCALES);
        } else {
          const halfx8 xh = *reinterpret_cast<const halfx8*>(&F[U][0]);
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[tx][a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, *reinterpret_cast<const halfx8*>(&gh[ph - 1][c]), acc[tx][a_][c], 0, 0, 0);
        }
      };
      rd_g(0);
#pragma unroll
      for (int U = 0; U < LA; ++U) rd(U);
#pragma unroll
      for (int U = 0; U < NU; ++U) {
        __builtin_amdgcn_sched_barrier(0);
        if (U + LA == 6) rd_g(1);
        if (U + LA == 12) rd_g(2);
        if (U + LA < NU) rd(U + LA);
        mm(U);
      }
      __builtin_amdgcn_sched_barrier(0);
#else
      // bf16 pairs: 12 units per step -- (s, tap, a): 4 reads -> 6 MFMAs; dy's fragments of half-step s read two units early
      constexpr int NU = 12;
      uint4 gh[2][TN], gl[2][TN];
      auto rd_g = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          gh[s][c] = gfrag(Ghi, s, c);
          gl[s][c] = gfrag(Glo, s, c);
        }
      };
      uint4 F[NU][2];
      auto rd = [&](int U) __attribute__((always_inline)) {
        const int s = U / 6, tx = (U % 6) >> 1, a_ = U & 1;
        F[U][0] = xfrag(Xhi, s, a_, tx);
        F[U][1] = xfrag(Xlo, s, a_, tx);
      };
      auto mm = [&](int U) __attribute__((always_inline)) {
        const int s = U / 6, tx = (U % 6) >> 1, a_ = U & 1;
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&F[U][0]), xl2 = *reinterpret_cast<const bf16x8*>(&F[U][1]);
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          acc[tx][a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl2, *reinterpret_cast<const bf16x8*>(&gh[s][c]), acc[tx][a_][c], 0, 0, 0);
          acc[tx][a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, *reinterpret_cast<const bf16x8*>(&gl[s][c]), acc[tx][a_][c], 0, 0, 0);
          acc[tx][a_][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, *reinterpret_cast<const bf16x8*>(&gh[s][c]), acc[tx][a_][c], 0, 0, 0);
        }
      };
      rd_g(0);
#pragma unroll
      for (int U = 0; U < LA; ++U) rd(U);
#pragma unroll
      for (int U = 0; U < NU; ++U) {
        __builtin_amdgcn_sched_barrier(0);
        if (U + LA == 6) rd_g(1);
        if (U + LA < NU) rd(U + LA);
        mm(U);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
    };
    barrier();
    for (int step = 0; step < n_steps; ++step) {
      compute(smem + (step & 1) * BUF);
      barrier();
    }
    // reduction over the row splits: f32 atomics into dW (fire and forget; see wgrad3f_kernel)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = ci0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const long long row = (long long)((ty * 3 + tx) * p.Cin + ci) * p.ld_w;
#pragma unroll
          for (int c = 0; c < TN; ++c) {
            const int co = n0 + wn * 32 * TN + c * 32 + il;
            if (co < p.Cout) atomicAdd(g_dw + row + co, acc[tx][a][c][r] * inv_g);
          }
        }
  }
  if (do_bias) {  // (uniform over the workgroup; the producers hold the sums: 256 threads, 16 rows x 8 column groups per pair)
    float* red = reinterpret_cast<float*>(smem);  // [32][BN] floats
    __syncthreads();
    if (producer) {
      const int rrow = ((tid & 255) >> 7) * 16 + prow;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<float4*>(red + rrow * BN + 8 * (q8 + 8 * j)) = bsum[2 * j];
        *reinterpret_cast<float4*>(red + rrow * BN + 8 * (q8 + 8 * j) + 4) = bsum[2 * j + 1];
      }
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

}  // namespace

void PP_API(pp4_launch_igemm4p)(hipStream_t st, IgemmParams& p, const void* ahi, const void* whi, const void* wlo, int w_rows, int w_ld8,
                                void* ohi, void* olo, int splits, float* ws, int n_cu, int nst) {
  p.n_tiles_n = (p.Nout + 127) / 128;
  const int n_items = ((p.M + 127) / 128) * p.n_tiles_n * splits;
  const int grid = n_items < n_cu ? n_items : n_cu;
  const long long a_bytes = p.src_rows * (long long)p.ld_src * 4;
  const long long w_bytes = (long long)p.w_taps * w_rows * w_ld8 * 16;
  const bool op = ohi != nullptr && splits == 1;
  auto go = [&](auto opc, auto nstc) {
    hipLaunchKernelGGL((igemm4p_kernel<decltype(opc)::value, decltype(nstc)::value>), dim3((unsigned)grid), dim3(256), 0, st, p, ahi, (unsigned)a_bytes, whi,
                       wlo, (unsigned)w_bytes, p.bias, p.addend, p.mask_src, p.out, (uint2*)(op ? ohi : nullptr), (uint2*)(op ? olo : nullptr), w_rows, w_ld8,
                       splits, ws, n_items, (p.Cred / 32) / splits, (p.Cred / 32) % splits);
  };
  if (op) {
    if (nst == 3) go(std::true_type{}, std::integral_constant<int, 3>{});
    else go(std::true_type{}, std::integral_constant<int, 4>{});
  } else {
    if (nst == 3) go(std::false_type{}, std::integral_constant<int, 3>{});
    else go(std::false_type{}, std::integral_constant<int, 4>{});
  }
}

// the tap-row-reuse weight gradient (wgrad3r_kernel); returns false when the launch does not meet its conditions (the caller then
// takes wgrad3f).  list: the listed 32-row blocks of dy (pp_row_block_list) or NULL.
bool PP_API(pp4_launch_wgrad3r)(hipStream_t st, Wgrad3Params& p, const void* xhi, const void* xlo, const void* dhi, const void* dlo, float* dw,
                                float* dbias, const int* list, int n_cu, int variant) {
  if (!(p.kh == 3 && p.kw == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.Cin % 128 == 0 && xhi && dhi)) return false;
  if (variant == 2 && list) return false;  // wgrad3w: dense reductions only
  int min_ow = 1 << 30;
  long long pos = 0;
  for (int i = 0; i < p.n_seg; ++i) {
    if (p.seg[i].OH != p.seg[i].SH || p.seg[i].OW != p.seg[i].SW || p.seg[i].row_begin != p.seg[i].src_row_begin) return false;
    if (list && i + 1 < p.n_seg && p.seg[i + 1].row_begin % 32 != 0) return false;  // (a listed 32-row block never straddles two levels)
    min_ow = p.seg[i].OW < min_ow ? p.seg[i].OW : min_ow;
    // the dense reduction walks positions: every image row followed by one pad
    const long long rows = (i + 1 < p.n_seg ? p.seg[i + 1].row_begin : p.M) - p.seg[i].row_begin;
    p.pos_begin[i] = (int)pos;
    pos += rows / p.seg[i].OW * (p.seg[i].OW + 1);
  }
  p.Mp = (int)pos;
  const long long x_bytes = p.src_rows * (long long)p.ld_src * 4, d_bytes = (long long)p.M * p.ld_dy * 4;
  if (!(min_ow >= (list ? 12 : 2) && x_bytes < (1ll << 31) && d_bytes < (1ll << 31) && pos < (1 << 24) && p.src_rows > 0)) return false;
  p.k_tiles_per_tap = p.Cin / 128;
  p.n_tiles_k = 3 * p.k_tiles_per_tap;
  p.n_tiles_n = (p.Cout + 127) / 128;
  const int tiles = p.n_tiles_k * p.n_tiles_n;
  const int R = list ? p.M : p.Mp;  // length of the reduction space
  // row splits: one workgroup per CU and ~1 us per 3-tap step; every split adds a |dW| of f32 atomics (~1.3 TB/s, about half of it
  // under other workgroups' loops)
  int max_splits = (R + 255) / 256;  // at least 8 steps per workgroup
  if (max_splits > 128) max_splits = 128;
  if (max_splits < 1) max_splits = 1;
  const double atomic_steps_per_split = 0.5 * (double)tiles * 3.0 * 128.0 * 128.0 * 4.0 / 1.3e6 / 1.0;
  int splits = 1;
  double best = 1e300;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const double steps = (double)((R + sp - 1) / sp + 31) / 32.0 + 6.0;
    const double rounds = (double)(((long long)tiles * sp + n_cu - 1) / n_cu);
    const double cost = rounds * steps + atomic_steps_per_split * sp;
    if (cost < best * 0.995) {
      best = cost;
      splits = sp;
    }
  }
  if (const char* e = getenv("PP_WGRAD3R_SPLITS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= max_splits) splits = v;
  }
  int rps = (R + splits - 1) / splits;
  rps = (rps + 31) / 32 * 32;
  splits = (R + rps - 1) / rps;
  p.splits = splits;
  p.rows_per_split = rps;
  if (p.sp_min_steps < 1) p.sp_min_steps = 1;
  if (getenv("PP_CONV_DEBUG")) fprintf(stderr, "wgrad3r tiles %d (3 taps each) splits %d (M %d)%s\n", tiles, splits, p.M, list ? " listed blocks" : "");
  if (variant == 2) {  // wgrad3w: producer / consumer waves (dense reductions; listed blocks stay with wgrad3f)
    hipLaunchKernelGGL(wgrad3w_kernel, dim3((unsigned)(tiles * splits)), dim3(512), 0, st, p, xhi, xlo, (unsigned)x_bytes, dhi, dlo,
                       (unsigned)d_bytes, dw, dbias);
    return true;
  }
  if (list)
    hipLaunchKernelGGL((wgrad3r_kernel<true>), dim3((unsigned)(tiles * splits)), dim3(256), 0, st, p, xhi, xlo, (unsigned)x_bytes, dhi, dlo,
                       (unsigned)d_bytes, dw, dbias, list);
  else
    hipLaunchKernelGGL((wgrad3r_kernel<false>), dim3((unsigned)(tiles * splits)), dim3(256), 0, st, p, xhi, xlo, (unsigned)x_bytes, dhi, dlo,
                       (unsigned)d_bytes, dw, dbias, (const int*)nullptr);
  return true;
}
