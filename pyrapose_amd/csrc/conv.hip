// Implicit-GEMM direct convolution (NHWC, float32) on v_mfma_f32_32x32x2_f32 for gfx950.
//
//   forward   y[m][co]  = sum_{tap,ci} x[gather(m,tap)][ci] * w[tap][ci][co]        (+bias, +residual, relu)
//   bwd-data  dx[m][ci] = sum_{tap,co} dy[gather'(m,tap)][co] * w[tap][ci][co]      (+addend, relu mask)
//   bwd-wgt   dw[tap][ci][co] += sum_m x[gather(m,tap)][ci] * dy[m][co]             (split over m, f32 atomics)
//
// Replaces the TF conv kernels behind every Conv2D of the reference graph
// (models/retinanet.py:9-131,180-214; keras_resnet ResNet50 via models/resnet.py:87).
//
// Design:  no im2col buffer -- the A tile is gathered straight from the NHWC
// activation into LDS, 16 reduction channels per step, stored k-major ([16][BM], rows rotated by
// 8*(k/4) so the transposing ds_write_b32 are conflict-free) so that each lane fetches the TM (TN)
// operands of its interleaved 32x32 sub-tiles with ONE ds_read_b{32,64,128}.  One wave owns a
// (32*TM)x(32*TN) output tile = TM*TN accumulators of v_mfma_f32_32x32x2_f32 (exact f32 fma chain,
// 64 cycles/SIMD each), 4 waves (2x2) per workgroup, LDS double-buffered with register staging
// (global loads for step s+1 are in flight under the MFMAs of step s; one barrier per step).
// Pyramid levels that share weights are ONE launch: rows are numbered level after level ("row
// space", pyrapose_hip.h) and only the gather knows the per-level geometry.
#include "conv_common.h"

// TM, TN in {1,2,4}: per-wave tile (32*TM)x(32*TN); workgroup tile BM=64*TM, BN=64*TN (2x2 waves).
// BT: the B (weight) tile is read transposed -- rows = output channel, 16 contiguous reduction
// channels (bwd-data);  SMALLC: packed-RGB stem, Cred == 4, one tap per float4.
template <int TM, int TN, bool BT, bool SMALLC>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 2 : 4) void igemm_kernel(const IgemmParams p, const float* __restrict__ g_src,
                                                       const float* __restrict__ g_wgt, const float* __restrict__ g_bias,
                                                       const float* __restrict__ g_addend, const float* __restrict__ g_mask,
                                                       float* __restrict__ g_out) {
  // the tensors are separate __restrict__ kernel arguments (not struct members) so that the compiler
  // knows the epilogue stores cannot alias the addend / mask loads and may batch them
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
  float* As = smem;
  float* Bs = smem + 2 * BK * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- A loader: thread -> rows r0 + 64*i, 4 consecutive reduction channels 4*kq ----
  const int kq = tid & 3, r0 = tid >> 2;
  RowPos rows[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rows[i] = decode_row(p, m0 + r0 + 64 * i);
  long long a_off[TM];
  bool a_ok[TM];

  const int n_taps = p.kh * p.kw;
  const int steps_per_tap = SMALLC ? 1 : p.Cred / BK;
  const int n_steps = SMALLC ? (n_taps * 4 + BK - 1) / BK : n_taps * steps_per_tap;

  // ---- B loader ----
  // !BT: rows = k (16), float4 along output channels.  BT: rows = output channel, float4 along k.
  const int b_nq = tid % (BN / 4), b_kb = tid / (BN / 4);

  float4 ra[TM], rb[TN];
  int tap = 0, ty = 0, tx = 0, red0 = 0;  // current tap and reduction-channel offset of the step being loaded

  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) a_ok[i] = tap_offset(p, rows[i], ty, tx, &a_off[i]);
  };

  auto load_step = [&](int step) {
    if (SMALLC) {
      // one tap per float4: this thread's tap = step*4 + kq
      int t = step * 4 + kq;
      int tyy = t / p.kw, txx = t - tyy * p.kw;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        long long off;
        bool ok = (t < n_taps) && tap_offset(p, rows[i], tyy, txx, &off);
        ra[i] = ok ? *reinterpret_cast<const float4*>(g_src + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ra[i] = a_ok[i] ? *reinterpret_cast<const float4*>(g_src + a_off[i] + red0 + 4 * kq)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (!BT) {
      const int wrow0 = SMALLC ? step * BK : tap * p.w_tap_rows + red0;
      const int c = n0 + 4 * b_nq;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int k = b_kb + (1024 / BN) * i;
        rb[i] = (c < p.ld_w) ? *reinterpret_cast<const float4*>(g_wgt + (long long)(wrow0 + k) * p.ld_w + c)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int n = n0 + r0 + 64 * i;
        rb[i] = (n < p.Nout)
                    ? *reinterpret_cast<const float4*>(g_wgt + (long long)(tap * p.w_tap_rows + n) * p.ld_w + red0 + 4 * kq)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  auto advance = [&]() {
    if (!SMALLC) {
      red0 += BK;
      if (red0 >= p.Cred) {
        red0 = 0;
        ++tap;
        ++tx;
        if (tx == p.kw) { tx = 0; ++ty; }
        set_tap();
      }
    }
  };

  auto store_step = [&](int buf) {
    float* A = As + buf * BK * BM;
    float* B = Bs + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int col = (r0 + 64 * i + 8 * kq) & (BM - 1);
      A[(4 * kq + 0) * BM + col] = ra[i].x;
      A[(4 * kq + 1) * BM + col] = ra[i].y;
      A[(4 * kq + 2) * BM + col] = ra[i].z;
      A[(4 * kq + 3) * BM + col] = ra[i].w;
    }
    if (!BT) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int k = b_kb + (1024 / BN) * i;
        *reinterpret_cast<float4*>(B + k * BN + 4 * b_nq) = rb[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int col = (r0 + 64 * i + 8 * kq) & (BN - 1);
        B[(4 * kq + 0) * BN + col] = rb[i].x;
        B[(4 * kq + 1) * BN + col] = rb[i].y;
        B[(4 * kq + 2) * BN + col] = rb[i].z;
        B[(4 * kq + 3) * BN + col] = rb[i].w;
      }
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
  const int a_col = wm * 32 * TM + il * TM;
  const int b_col = wn * 32 * TN + il * TN;
  typedef typename VecT<TM>::type AV;
  typedef typename VecT<TN>::type BV;

  if (!SMALLC) set_tap();
  load_step(0);
  store_step(0);
  __syncthreads();

  for (int step = 0; step < n_steps; ++step) {
    const int buf = step & 1;
    const bool more = step + 1 < n_steps;
    if (more) {
      advance();
      load_step(step + 1);
    }
    const float* A = As + buf * BK * BM;
    const float* B = Bs + buf * BK * BN;
    // Operand fragments are fetched one k-quad (two MFMA k-steps) ahead of the MFMAs that consume them
    // (register double buffer); sched_group_barrier pins the order [reads of quad q+1][MFMAs of quad q] so
    // the LDS latency sits under 2*TM*TN MFMAs instead of in front of them.
    AV av[2][2];
    BV bv[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      av[0][j] = *reinterpret_cast<const AV*>(A + (2 * j + h) * BM + (a_col & (BM - 1)));
      bv[0][j] = *reinterpret_cast<const BV*>(B + (2 * j + h) * BN + (b_col & (BN - 1)));
    }
#pragma unroll
    for (int q = 0; q < BK / 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q + 1 < BK / 4) {
        const int rot = 8 * (q + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 4 * (q + 1) + 2 * j + h;
          av[nxt][j] = *reinterpret_cast<const AV*>(A + k * BM + ((a_col + rot) & (BM - 1)));
          bv[nxt][j] = *reinterpret_cast<const BV*>(B + k * BN + ((b_col + (BT ? rot : 0)) & (BN - 1)));
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_get(av[cur][j], a), vec_get(bv[cur][j], b), acc[a][b], 0, 0, 0);
      if (q + 1 < BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS reads of the next quad first
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);            // then this quad's MFMAs
    }
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: +bias, +addend, relu-mask, relu ----
  // The accumulator tile goes through LDS (the staging buffers are free now) so that every global access
  // of the epilogue -- output store, residual / addend load, ReLU-mask load -- is a 16-byte-per-lane,
  // row-contiguous access (32 lanes = one 512 B row segment), independent of the MFMA register layout.
  // Pass hm handles the 32*TM rows owned by the waves with wm == hm.  Columns are processed in float4
  // groups up to Nout rounded up to 4 (the weight padding columns are zero, so zeros land there).
  constexpr int SUB = (TM * BN > 256) ? 2 : TM;  // accumulator sub-rows staged per pass (LDS holds 32*SUB rows)
  constexpr int ROWS = 32 * SUB;         // rows per pass
  constexpr int C4 = BN / 4;             // float4 groups per row
  constexpr int RPI = 256 / C4;          // rows covered by one sweep of the 256 threads
  constexpr int SWEEPS = ROWS / RPI;
  static_assert(ROWS * BN <= 2 * BK * (BM + BN), "epilogue staging does not fit the LDS buffers");
  float* stage = smem;                   // [ROWS][BN]
  const int e_c4 = tid % C4, e_r = tid / C4;
  const int co = n0 + 4 * e_c4;
  const bool col_ok = co < ((p.Nout + 3) & ~3);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_bias && col_ok) bias4 = *reinterpret_cast<const float4*>(g_bias + co);
  const int m_last = p.M - 1;
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
#pragma unroll
    for (int a0 = 0; a0 < TM; a0 += SUB) {
      __syncthreads();  // staging area free (K loop reads / previous pass reads done)
      if (wm == hm) {
#pragma unroll
        for (int as = 0; as < SUB; ++as)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = ((r & 3) + 8 * (r >> 2) + 4 * h) * SUB + as;
            float* dst = stage + row * BN + b_col;
            const int a = a0 + as;
            if (TN == 1) dst[0] = acc[a][0][r];
            if (TN == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[a][0][r], acc[a][TN > 1 ? 1 : 0][r]);
            if (TN == 4)
              *reinterpret_cast<float4*>(dst) = make_float4(acc[a][0][r], acc[a][TN > 1 ? 1 : 0][r], acc[a][TN > 2 ? 2 : 0][r], acc[a][TN > 3 ? 3 : 0][r]);
          }
      }
      __syncthreads();
      if (col_ok) {
        // straight-line load groups: rows are clamped (never predicated) and the has-addend / has-mask cases are
        // wave-uniform branches, so the compiler can keep a whole group of loads in flight behind ONE wait
        constexpr int G = SWEEPS < 4 ? SWEEPS : 4;
        // staged row sr holds tile row (sr / SUB) * TM + a0 + sr % SUB of the half owned by wm == hm
        auto tile_row = [&](int sr) { return m0 + hm * 32 * TM + (sr / SUB) * TM + a0 + (sr % SUB); };
        auto sweep = [&](auto has_add, auto has_mask) {
#pragma unroll
          for (int s0 = 0; s0 < SWEEPS; s0 += G) {
            float4 ad[G], mk[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int m = min(tile_row(e_r + RPI * (s0 + g)), m_last);
              if (has_add) ad[g] = *reinterpret_cast<const float4*>(g_addend + (long long)m * p.ld_add + co);
              if (has_mask) mk[g] = *reinterpret_cast<const float4*>(g_mask + (long long)m * p.ld_mask + co);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int row = e_r + RPI * (s0 + g);
              const int m = tile_row(row);
              float4 v = *reinterpret_cast<const float4*>(stage + row * BN + 4 * e_c4);
              v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
              if (has_add) { v.x += ad[g].x; v.y += ad[g].y; v.z += ad[g].z; v.w += ad[g].w; }
              if (has_mask) {
                v.x = mk[g].x > 0.f ? v.x : 0.f; v.y = mk[g].y > 0.f ? v.y : 0.f;
                v.z = mk[g].z > 0.f ? v.z : 0.f; v.w = mk[g].w > 0.f ? v.w : 0.f;
              }
              if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              if (m <= m_last) *reinterpret_cast<float4*>(g_out + (long long)m * p.ld_out + co) = v;
            }
          }
        };
        if (g_addend != nullptr && g_mask != nullptr) sweep(std::true_type{}, std::true_type{});
        else if (g_addend != nullptr) sweep(std::true_type{}, std::false_type{});
        else if (g_mask != nullptr) sweep(std::false_type{}, std::true_type{});
        else sweep(std::false_type{}, std::false_type{});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
struct WgradParams {
  const float* src;
  const float* dy;
  float* dw;
  float* dbias;
  int ld_src, ld_dy, ld_w;
  int M, n_seg;
  SegGeo seg[PP_MAX_SEG];
  int Cin, Cout;
  int kh, kw, stride, pad_t, pad_l;
  int k_tiles_per_tap, n_tiles_k, n_tiles_n, splits, rows_per_split;
};

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p, const float* __restrict__ g_src,
                                                       const float* __restrict__ g_dy, float* __restrict__ g_dw,
                                                       float* __restrict__ g_dbias) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
  float* As = smem;
  float* Bs = smem + 2 * BK * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  const int a_cq = tid % (BM / 4), a_rb = tid / (BM / 4);  // rows a_rb + (1024/BM)*i
  const int b_cq = tid % (BN / 4), b_rb = tid / (BN / 4);

  float4 ra[TM], rb[TN];

  // Each thread gathers the same TM rows of every 16-row step; their (image, y, x) position is carried
  // incrementally (+16 rows per step) instead of being re-derived with integer divisions every step.
  struct WRow { int m, n, y, x, seg_end, sb, OH, OW, SH, SW; };
  WRow wr[TM];
  auto decode = [&](WRow& w, int m) {
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
#pragma unroll
  for (int i = 0; i < TM; ++i) decode(wr[i], m_begin + a_rb + (1024 / BM) * i);

  auto load_step = [&](int step) {
    const int mb = m_begin + step * BK;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const WRow& w = wr[i];
      const int sy = w.y * p.stride + ty - p.pad_t;
      const int sx = w.x * p.stride + tx - p.pad_l;
      const bool ok = (w.m < m_end) && ((unsigned)sy < (unsigned)w.SH) && ((unsigned)sx < (unsigned)w.SW);
      ra[i] = ok ? *reinterpret_cast<const float4*>(g_src + (long long)(w.sb + w.n * w.SH * w.SW + sy * w.SW + sx) * p.ld_src + ci0 + 4 * a_cq)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int m = mb + b_rb + (1024 / BN) * i;
      const int c = n0 + 4 * b_cq;
      rb[i] = (m < m_end && c < p.ld_dy) ? *reinterpret_cast<const float4*>(g_dy + (long long)m * p.ld_dy + c)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // advance the gather rows to the next step
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      WRow& w = wr[i];
      const int m2 = w.m + BK;
      if (m2 >= w.seg_end) {
        decode(w, m2);  // crosses into the next pyramid level (or past the end): rare
      } else {
        w.m = m2;
        w.x += BK;
        while (w.x >= w.OW) { w.x -= w.OW; ++w.y; }
        while (w.y >= w.OH) { w.y -= w.OH; ++w.n; }
      }
    }
  };
  auto store_step = [&](int buf) {
    float* A = As + buf * BK * BM;
    float* B = Bs + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) *reinterpret_cast<float4*>(A + (a_rb + (1024 / BM) * i) * BM + 4 * a_cq) = ra[i];
#pragma unroll
    for (int i = 0; i < TN; ++i) *reinterpret_cast<float4*>(B + (b_rb + (1024 / BN) * i) * BN + 4 * b_cq) = rb[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  float bias_sum = 0.f;
  const bool do_bias = (p.dbias != nullptr) && (tile_k == 0) && (tid < BN);

  const int il = lane & 31, h = lane >> 5;
  const int a_col = wm * 32 * TM + il * TM;
  const int b_col = wn * 32 * TN + il;  // blocked columns: + 32*c
  typedef typename VecT<TM>::type AV;

  if (n_steps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int step = 0; step < n_steps; ++step) {
    const int buf = step & 1;
    const bool more = step + 1 < n_steps;
    if (more) load_step(step + 1);
    const float* A = As + buf * BK * BM;
    const float* B = Bs + buf * BK * BN;
    AV av[2][2];
    float bv[2][2][TN];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      av[0][j] = *reinterpret_cast<const AV*>(A + (2 * j + h) * BM + a_col);
#pragma unroll
      for (int c = 0; c < TN; ++c) bv[0][j][c] = B[(2 * j + h) * BN + b_col + 32 * c];
    }
#pragma unroll
    for (int q = 0; q < BK / 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q + 1 < BK / 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 4 * (q + 1) + 2 * j + h;
          av[nxt][j] = *reinterpret_cast<const AV*>(A + k * BM + a_col);
#pragma unroll
          for (int c = 0; c < TN; ++c) bv[nxt][j][c] = B[k * BN + b_col + 32 * c];
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int c = 0; c < TN; ++c)
            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_get(av[cur][j], a), bv[cur][j][c], acc[a][c], 0, 0, 0);
      if (q + 1 < BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
    }
    if (do_bias) {
#pragma unroll
      for (int k = 0; k < BK; ++k) bias_sum += B[k * BN + tid];
    }
    if (more) store_step(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row_local = wm * 32 * TM + ((r & 3) + 8 * (r >> 2) + 4 * h) * TM + a;
      const int ci = ci0 + row_local;
      float* dst = g_dw + (long long)(tap * p.Cin + ci) * p.ld_w;
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        const int co = n0 + b_col + 32 * c;
        if (co < p.Cout) atomicAdd(dst + co, acc[a][c][r]);
      }
    }
  }
  if (do_bias && n0 + tid < p.Cout) atomicAdd(g_dbias + n0 + tid, bias_sum);
}

// ------------------------------------------------------------------------------------------------
// host side
// Pick the workgroup tile (64*tm x 64*tn): minimise est_rounds * tm*tn / eff(tm,tn), eff = in-flight rate of the
// tile shape relative to 128x128 (smaller tiles re-read more operand bytes per flop).  Padded output columns
// (144 -> 192 or 256) enter through the workgroup count.
static void pick_tile(const pp_ctx* ctx, int M, int Nout, int* tm, int* tn) {
  static const int cand[4][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}};
  static const double eff[4] = {1.0, 0.955, 0.92, 0.915};
  static const int slots[4] = {4, 5, 5, 8};
  const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
  double best = 1e300;
  for (int i = 0; i < 4; ++i) {
    const int bm = 64 * cand[i][0], bn = 64 * cand[i][1];
    const long long blocks = (long long)((M + bm - 1) / bm) * ((Nout + bn - 1) / bn);
    const double t = est_rounds(blocks, slots[i], cus) * cand[i][0] * cand[i][1] / eff[i];
    if (t < best * 0.999) {
      best = t;
      *tm = cand[i][0];
      *tn = cand[i][1];
    }
  }
  const char* e = getenv("PP_CONV_TILE");  // tuning hook: "tm,tn"
  if (e && e[0] && e[1] == ',' && e[2]) {
    *tm = e[0] - '0';
    *tn = e[2] - '0';
  }
}

template <int TM, int TN, bool BT, bool SMALLC>
static void launch_igemm(hipStream_t st, IgemmParams& p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.n_tiles_n = (p.Nout + BN - 1) / BN;
  int n_tiles_m = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL((igemm_kernel<TM, TN, BT, SMALLC>), dim3((unsigned)(n_tiles_m * p.n_tiles_n)), dim3(256), 0, st, p, p.src,
                     p.wgt, p.bias, p.addend, p.mask_src, p.out);
}

extern "C" int pp_conv2d_nhwc_fwd(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* w, const float* bias,
                                  const float* residual, int ld_res, int relu, float* y) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_fwd");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, x && w && y, PP_ERR_ARG, "pp_conv2d_nhwc_fwd: null tensor");
  PP_CHECK_ARG(ctx, pp_is_aligned16(x) && pp_is_aligned16(w) && pp_is_aligned16(y), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_fwd: tensors must be 16-byte aligned");
  PP_CHECK_ARG(ctx, !residual || (ld_res % 4 == 0 && ld_res >= ((d->cout + 3) & ~3)), PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd: ld_res");
  PP_CHECK_ARG(ctx, (!residual || pp_is_aligned16(residual)) && (!bias || pp_is_aligned16(bias)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_fwd: bias / residual must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = x; p.wgt = w; p.out = y; p.bias = bias; p.addend = residual; p.mask_src = nullptr;
  p.ld_src = d->ld_x; p.ld_w = d->ld_w; p.ld_out = d->ld_y; p.ld_add = ld_res; p.ld_mask = 0;
  p.relu = relu;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M);
  p.Cred = d->cin; p.Nout = d->cout; p.w_tap_rows = d->cin;
  p.kh = d->kh; p.kw = d->kw;
  p.mul = d->stride; p.tsign = 1; p.off_y = -d->pad_t; p.off_x = -d->pad_l; p.div = 1;
  int tm, tn;
  pick_tile(ctx, p.M, p.Nout, &tm, &tn);
  if (d->cin == 4) {
    if (tn == 1) launch_igemm<2, 1, false, true>(ctx->stream, p);
    else launch_igemm<2, 2, false, true>(ctx->stream, p);
  } else if (tm == 4) {
    launch_igemm<4, 2, false, false>(ctx->stream, p);
  } else if (tm == 2) {
    if (tn == 1) launch_igemm<2, 1, false, false>(ctx->stream, p);
    else launch_igemm<2, 2, false, false>(ctx->stream, p);
  } else {
    if (tn == 1) launch_igemm<1, 1, false, false>(ctx->stream, p);
    else launch_igemm<1, 2, false, false>(ctx->stream, p);
  }
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_fwd");
  return PP_OK;
}

extern "C" int pp_conv2d_nhwc_bwd_data(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const float* w, const float* addend,
                                       int ld_add, const float* relu_src, int ld_rs, float* dx) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_bwd_data");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, dy && w && dx, PP_ERR_ARG, "pp_conv2d_nhwc_bwd_data: null tensor");
  PP_CHECK_ARG(ctx, d->cin % 16 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_data: cin %d", d->cin);
  const int cred = (d->cout + 15) / 16 * 16;
  PP_CHECK_ARG(ctx, d->ld_y >= cred && d->ld_y % 4 == 0, PP_ERR_SHAPE,
               "pp_conv2d_nhwc_bwd_data: dy needs ld_y >= %d (cout rounded up to 16, zero padded)", cred);
  PP_CHECK_ARG(ctx, pp_is_aligned16(dy) && pp_is_aligned16(w) && pp_is_aligned16(dx), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_data: tensors must be 16-byte aligned");
  PP_CHECK_ARG(ctx, (!addend || (ld_add >= d->cin && ld_add % 4 == 0)) && (!relu_src || (ld_rs >= d->cin && ld_rs % 4 == 0)), PP_ERR_SHAPE,
               "pp_conv2d_nhwc_bwd_data: ld");
  PP_CHECK_ARG(ctx, (!addend || pp_is_aligned16(addend)) && (!relu_src || pp_is_aligned16(relu_src)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_bwd_data: addend / relu_src must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = dy; p.wgt = w; p.out = dx; p.bias = nullptr; p.addend = addend; p.mask_src = relu_src;
  p.ld_src = d->ld_y; p.ld_w = d->ld_w; p.ld_out = d->ld_x; p.ld_add = ld_add; p.ld_mask = ld_rs;
  p.relu = 0;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, false, p.seg, &p.M);
  p.Cred = cred; p.Nout = d->cin; p.w_tap_rows = d->cin;
  p.kh = d->kh; p.kw = d->kw;
  p.mul = 1; p.tsign = -1; p.off_y = d->pad_t; p.off_x = d->pad_l; p.div = d->stride;
  int tm, tn;
  pick_tile(ctx, p.M, p.Nout, &tm, &tn);
  if (tm == 4) {
    launch_igemm<4, 2, true, false>(ctx->stream, p);
  } else if (tm == 2) {
    if (tn == 1) launch_igemm<2, 1, true, false>(ctx->stream, p);
    else launch_igemm<2, 2, true, false>(ctx->stream, p);
  } else {
    if (tn == 1) launch_igemm<1, 1, true, false>(ctx->stream, p);
    else launch_igemm<1, 2, true, false>(ctx->stream, p);
  }
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_data");
  return PP_OK;
}

template <int TM, int TN>
static void launch_wgrad(pp_ctx* ctx, WgradParams& p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.k_tiles_per_tap = p.Cin / BM;
  p.n_tiles_k = p.kh * p.kw * p.k_tiles_per_tap;
  p.n_tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles = p.n_tiles_k * p.n_tiles_n;
  const int cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
  // split the row reduction: minimise  est_rounds(tiles*splits) * steps_per_split  (+ the f32-atomic traffic of
  // `splits` partial tiles at ~1.3 TB/s, in units of one 16-row step ~ 1 us), keeping >= 16 steps per workgroup
  int max_splits = (p.M + 255) / 256;
  if (max_splits < 1) max_splits = 1;
  if (max_splits > 64) max_splits = 64;
  const int slots = (TM * TN == 4) ? 3 : (TM * TN == 2 ? 5 : 8);
  const double tile_work = (double)(TM * TN) / 4.0;                       // relative to 128x128
  const double atomic_us_per_split = (double)tiles * BM * BN * 4.0 / 1.3e6;  // bytes / (1.3 TB/s) in us
  int splits = 1;
  double best = 1e300;
  for (int sp = 1; sp <= max_splits; ++sp) {
    const double steps = (double)((p.M + sp - 1) / sp + 15) / 16.0;
    const double cost = est_rounds((long long)tiles * sp, slots, cus) * steps * tile_work * 1.05 + atomic_us_per_split * sp;
    if (cost < best * 0.995) {
      best = cost;
      splits = sp;
    }
  }
  int rps = (p.M + splits - 1) / splits;
  rps = (rps + 15) / 16 * 16;
  splits = (p.M + rps - 1) / rps;
  p.splits = splits;
  p.rows_per_split = rps;
  hipLaunchKernelGGL((wgrad_kernel<TM, TN>), dim3((unsigned)(tiles * splits)), dim3(256), 0, ctx->stream, p, p.src, p.dy, p.dw, p.dbias);
}

extern "C" int pp_conv2d_nhwc_bwd_weight(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy, float* dw,
                                         float* dbias) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_bwd_weight");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, x && dy && dw, PP_ERR_ARG, "pp_conv2d_nhwc_bwd_weight: null tensor");
  PP_CHECK_ARG(ctx, d->cin % 64 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_weight: cin %d must be a multiple of 64", d->cin);
  PP_CHECK_ARG(ctx, d->ld_y % 4 == 0, PP_ERR_SHAPE, "pp_conv2d_nhwc_bwd_weight: ld_y %d must be a multiple of 4", d->ld_y);
  PP_CHECK_ARG(ctx, pp_is_aligned16(x) && pp_is_aligned16(dy), PP_ERR_ALIGN, "pp_conv2d_nhwc_bwd_weight: tensors must be 16-byte aligned");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.src = x; p.dy = dy; p.dw = dw; p.dbias = dbias;
  p.ld_src = d->ld_x; p.ld_dy = d->ld_y; p.ld_w = d->ld_w;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M);
  p.Cin = d->cin; p.Cout = d->cout;
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  const bool big_k = (d->cin % 128 == 0);
  const bool big_n = ((d->cout + 127) / 128 * 128) <= ((d->cout + 63) / 64 * 64);
  if (big_k && big_n) launch_wgrad<2, 2>(ctx, p);
  else if (big_k) launch_wgrad<2, 1>(ctx, p);
  else if (big_n) launch_wgrad<1, 2>(ctx, p);
  else launch_wgrad<1, 1>(ctx, p);
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_bwd_weight");
  return PP_OK;
}
