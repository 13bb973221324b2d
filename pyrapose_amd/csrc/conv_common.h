// Shared pieces of the implicit-GEMM convolution kernels (conv.hip: f32 MFMA; conv3.hip: 3 x bf16 MFMA):
// row-space geometry, the gather (row -> source offset per tap), the XCD-aware tile order, descriptor checks
// and the launch-shape cost model.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "pp_internal.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct SegGeo {
  int row_begin;      // first row (m) of the segment in the space the kernel enumerates
  int src_row_begin;  // first row of the segment in the gathered tensor
  int OH, OW;         // cells per image in the enumerated space
  int SH, SW;         // cells per image in the gathered tensor
};

struct IgemmParams {
  const float* src;
  const float* wgt;
  float* out;
  const float* bias;
  const float* addend;
  const float* mask_src;
  int ld_src, ld_w, ld_out, ld_add, ld_mask;
  int relu;
  int M, n_seg;
  SegGeo seg[PP_MAX_SEG];
  int Cred;        // reduction channels per tap (multiple of 16, or 4 for the packed-RGB stem)
  int Nout;        // output channels
  int w_tap_rows;  // weight rows per tap (= cin of the forward conv)
  int kh, kw;
  int mul, tsign, off_y, off_x, div;  // src = (pos*mul + tap*tsign + off) / div
  int n_tiles_n;
  long long src_rows;  // rows of the gathered tensor (all segments)
  // weight tap of loop tap (ty, tx) = (w_ty0 + w_tstep*ty) * w_kw + (w_tx0 + w_tstep*tx); identity: 0, 1, 0, kw
  int w_ty0, w_tx0, w_tstep, w_kw, w_taps;  // w_taps: taps in the weight planes (kh * kw of the conv)
  // stride-2 bwd-data by parity class: enumerated row (n, y', x') of the class grid seg[0].OH x OW is written to (and
  // reads addend / mask at) row (n*sc_H + 2y' + sc_cy) * sc_W + 2x' + sc_cx of the full tensor
  int sc_on, sc_H, sc_W, sc_cy, sc_cx;
  // epilogue operands stored as bf16 (hi, lo) planes instead of f32 (pp_ctx_set_epilogue_planes): the addend / residual
  // (value = hi + lo, geometry [rows][ld_add]) and the ReLU source (only its hi plane is read: hi > 0 <=> value > 0,
  // geometry [rows][ld_mask]).  When set they replace `addend` / `mask_src`.
  const void* add_hi;
  const void* add_lo;
  const void* mask_hi;
  int x_ty_inner;  // igemm3x: loop order of the (kernel row, channel chunk) groups
  int m_off;       // igemm3x / splitk_finish_kernel: first output row of THIS launch (the rows below belong to another launch:
                   // the remainder of a launch whose full rounds igemm4x_kernel took); split-K slices hold rows m_off .. M - 1
};

// the weight-gradient kernels (conv3.hip: wgrad3 / wgrad3f; conv4.hip: wgrad3r)
struct Wgrad3Params {
  int ld_src, ld_dy, ld_w;
  int M, n_seg;
  SegGeo seg[PP_MAX_SEG];
  int Cin, Cout;
  int kh, kw, stride, pad_t, pad_l;
  int k_tiles_per_tap, n_tiles_k, n_tiles_n, splits, rows_per_split;
  int max_wraps;  // ceil(32 / narrowest level width): row wraps one 32-row step can cross
  int sp_min_steps;  // SP: listed 32-row blocks one workgroup should at least reduce (fewer splits when the list is short)
  long long src_rows;
  const float* inv_scale;  // device scalar 2^-G (NULL: 1): dy travels multiplied by 2^G (pp_ctx_set_grad_scale), dW / dbias leave unscaled
  // wgrad3r (conv4.hip), dense reduction: the position space with one pad behind every image row -- first position of each level, total
  int pos_begin[PP_MAX_SEG];
  int Mp;
};

// Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, each with a private L2).
// Remap so that every XCD walks a CONTIGUOUS range of logical tiles: the N-tiles of one M-tile (same A
// rows) and neighbouring M-tiles (overlapping 3x3 halos, same weight step) then share one L2.
// Bijective for any grid size (cdna_hip_programming.md T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int x = b & 7, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

template <int V>
struct VecT;
template <>
struct VecT<1> { typedef float type; };
template <>
struct VecT<2> { typedef float2 type; };
template <>
struct VecT<4> { typedef float4 type; };

__device__ __forceinline__ float vec_get(float v, int) { return v; }
__device__ __forceinline__ float vec_get(float2 v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float vec_get(float4 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

struct RowPos {
  int ybase, xbase, rowbase, SH, SW;
  bool ok;
};

// m -> (segment, image, y, x) -> gather bases.  n_seg <= PP_MAX_SEG, uniform loop + selects.
__device__ __forceinline__ RowPos decode_row(const IgemmParams& p, int m) {
  RowPos r;
  r.ok = m < p.M;
  int rb = p.seg[0].row_begin, sb = p.seg[0].src_row_begin, OH = p.seg[0].OH, OW = p.seg[0].OW,
      SH = p.seg[0].SH, SW = p.seg[0].SW;
  for (int s = 1; s < p.n_seg; ++s) {
    if (m >= p.seg[s].row_begin) {
      rb = p.seg[s].row_begin; sb = p.seg[s].src_row_begin; OH = p.seg[s].OH; OW = p.seg[s].OW;
      SH = p.seg[s].SH; SW = p.seg[s].SW;
    }
  }
  int local = r.ok ? m - rb : 0;
  int hw = OH * OW;
  int n = local / hw;
  int rem = local - n * hw;
  int y = rem / OW;
  int x = rem - y * OW;
  r.ybase = y * p.mul + p.off_y;
  r.xbase = x * p.mul + p.off_x;
  r.rowbase = sb + n * SH * SW;
  r.SH = SH;
  r.SW = SW;
  return r;
}

__device__ __forceinline__ bool tap_offset(const IgemmParams& p, const RowPos& r, int ty, int tx, long long* off) {
  int sy = r.ybase + ty * p.tsign;
  int sx = r.xbase + tx * p.tsign;
  bool ok = r.ok;
  if (p.div > 1) {
    ok = ok && (sy % p.div == 0) && (sx % p.div == 0);
    sy /= p.div;
    sx /= p.div;
  }
  ok = ok && ((unsigned)sy < (unsigned)r.SH) && ((unsigned)sx < (unsigned)r.SW);
  *off = ok ? (long long)(r.rowbase + sy * r.SW + sx) * p.ld_src : 0;
  return ok;
}


// ---- host side ----
static int fill_segs(pp_ctx* ctx, const pp_conv_desc* d, bool enumerate_out, SegGeo* seg, int* M_out, long long* src_rows_out = nullptr) {
  // enumerate_out: rows enumerate the OUTPUT space and gather from the input (fwd, wgrad);
  // otherwise rows enumerate the INPUT space and gather from the output-space tensor (bwd-data).
  const pp_rowspace* e = enumerate_out ? &d->out : &d->in;
  const pp_rowspace* g = enumerate_out ? &d->in : &d->out;
  long long rb = 0, sb = 0;
  for (int s = 0; s < e->n_seg; ++s) {
    seg[s].row_begin = (int)rb;
    seg[s].src_row_begin = (int)sb;
    seg[s].OH = e->h[s]; seg[s].OW = e->w[s];
    seg[s].SH = g->h[s]; seg[s].SW = g->w[s];
    rb += (long long)e->n_img * e->h[s] * e->w[s];
    sb += (long long)g->n_img * g->h[s] * g->w[s];
  }
  *M_out = (int)rb;
  if (src_rows_out) *src_rows_out = sb;
  (void)ctx;
  return 0;
}

static int check_desc(pp_ctx* ctx, const pp_conv_desc* d, const char* who) {
  PP_CHECK_ARG(ctx, d != nullptr, PP_ERR_ARG, "%s: null descriptor", who);
  PP_CHECK_ARG(ctx, pp_rowspace_ok(&d->in) && pp_rowspace_ok(&d->out), PP_ERR_SHAPE, "%s: bad row space", who);
  PP_CHECK_ARG(ctx, d->in.n_seg == d->out.n_seg && d->in.n_img == d->out.n_img, PP_ERR_SHAPE,
               "%s: in/out row spaces disagree", who);
  PP_CHECK_ARG(ctx, d->kh > 0 && d->kw > 0 && d->kh <= 7 && d->kw <= 7 && d->stride >= 1 && d->stride <= 2, PP_ERR_SHAPE,
               "%s: unsupported kernel %dx%d stride %d", who, d->kh, d->kw, d->stride);
  PP_CHECK_ARG(ctx, d->cin > 0 && d->cout > 0 && (d->cin % 16 == 0 || d->cin == 4), PP_ERR_SHAPE,
               "%s: cin %d must be a multiple of 16 (or 4 for the packed stem)", who, d->cin);
  PP_CHECK_ARG(ctx, d->ld_w % 16 == 0 && d->ld_w >= d->cout, PP_ERR_SHAPE, "%s: ld_w %d (cout %d) must be a multiple of 16", who,
               d->ld_w, d->cout);
  PP_CHECK_ARG(ctx, d->ld_x % 4 == 0 && d->ld_x >= d->cin, PP_ERR_SHAPE, "%s: ld_x %d < cin %d or not a multiple of 4", who, d->ld_x, d->cin);
  PP_CHECK_ARG(ctx, d->ld_y % 4 == 0 && d->ld_y >= ((d->cout + 3) & ~3), PP_ERR_SHAPE,
               "%s: ld_y %d must be a multiple of 4 and >= cout %d rounded up to 4", who, d->ld_y, d->cout);
  PP_CHECK_ARG(ctx, d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < d->kh && d->pad_l < d->kw, PP_ERR_SHAPE, "%s: bad padding", who);
  for (int s = 0; s < d->in.n_seg; ++s) {
    // every output cell must lie inside the (bottom/right zero-extended) input: OH <= ceil((H + pad_t)/stride)
    PP_CHECK_ARG(ctx, (d->out.h[s] - 1) * d->stride - d->pad_t < d->in.h[s] && (d->out.w[s] - 1) * d->stride - d->pad_l < d->in.w[s],
                 PP_ERR_SHAPE, "%s: output %dx%d does not fit input %dx%d", who, d->out.h[s], d->out.w[s], d->in.h[s], d->in.w[s]);
  }
  if (d->in.n_seg > 1) PP_CHECK_ARG(ctx, d->stride == 1, PP_ERR_SHAPE, "%s: multi-level row spaces need stride 1", who);
  return PP_OK;
}


// ---- launch-shape cost model (measured with tools/conv_bench.py on MI355X; DESIGN.md §Tile choice) ----
// A CU that holds c equal workgroups retires them at a relative MFMA rate thr(c) (one 4-wave workgroup per CU
// cannot hide its own staging latency: 0.62; four: 0.82 of the f32-MFMA peak).  Equal-length workgroups
// finish in lock-step rounds, so  time / L = full_rounds * S / thr(S) + c_tail / thr(c_tail)  with S the
// co-resident slots per CU, c_tail = ceil(remainder / CUs) and L the time of one workgroup alone at full rate.
static double est_rounds(long long blocks, int slots, int cus) {
  static const double thr[5] = {1.0, 0.62, 0.78, 0.81, 0.82};
  const long long per_round = (long long)cus * slots;
  const long long full = blocks / per_round;
  const long long rem = blocks - full * per_round;
  const int s_idx = slots < 4 ? slots : 4;
  double t = (double)full * slots / thr[s_idx];
  if (rem > 0) {
    const int c = (int)((rem + cus - 1) / cus);
    t += (double)c / thr[c < 4 ? c : 4];
  }
  return t;
}

