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
