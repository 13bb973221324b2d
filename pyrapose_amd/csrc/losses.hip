// Fused loss forward+backward kernels (float32): no where/gather_nd materialisation -- the anchor
// state column predicates each element in place, wavefront shuffles reduce the loss sum, and the
// backward writes d(loss)/d(logit) directly (sigmoid' folded in).
//   focal          : losses.py:22-68   (cls + mask heads, bin/train.py:98-99)
//   orthogonal_l1  : losses.py:321-408 ('3Dbox' head, bin/train.py:97)
#include "pp_internal.h"

struct HeadGeo {
  int n_seg, n_img;
  int row_begin[PP_MAX_SEG + 1];
  int hw[PP_MAX_SEG];
  int cell_off[PP_MAX_SEG];
  int cells_total;
};

static void fill_head_geo(const pp_rowspace* rs, HeadGeo* g) {
  g->n_seg = rs->n_seg;
  g->n_img = rs->n_img;
  int rb = 0, co = 0;
  for (int s = 0; s < rs->n_seg; ++s) {
    g->row_begin[s] = rb;
    g->hw[s] = rs->h[s] * rs->w[s];
    g->cell_off[s] = co;
    rb += rs->n_img * g->hw[s];
    co += g->hw[s];
  }
  g->row_begin[rs->n_seg] = rb;
  g->cells_total = co;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ void block_accumulate(float v, float* dst) {
  __shared__ float part[16];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
    if (s != 0.f) atomicAdd(dst, s);
  }
}

// ---- positive counts (normalisers) --------------------------------------------------------------
__global__ void count_pos_kernel(size_t rows, int stride, const float* __restrict__ y, int* __restrict__ out) {
  int cnt = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x)
    cnt += (y[i * stride + stride - 1] == 1.0f) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

extern "C" int pp_count_positives(pp_ctx* ctx, size_t rows_box, const float* y_box, size_t rows_cls, int c_cls, const float* y_cls,
                                  size_t rows_mask, int c_mask, const float* y_mask, int* counts) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, counts, PP_ERR_ARG, "pp_count_positives: null counts");
  if (y_box && rows_box)
    hipLaunchKernelGGL(count_pos_kernel, dim3((unsigned)((rows_box + 255) / 256 > 1024 ? 1024 : (rows_box + 255) / 256)), dim3(256), 0,
                       ctx->stream, rows_box, 17, y_box, counts + 0);
  if (y_cls && rows_cls)
    hipLaunchKernelGGL(count_pos_kernel, dim3((unsigned)((rows_cls + 255) / 256 > 1024 ? 1024 : (rows_cls + 255) / 256)), dim3(256), 0,
                       ctx->stream, rows_cls, c_cls + 1, y_cls, counts + 1);
  if (y_mask && rows_mask)
    hipLaunchKernelGGL(count_pos_kernel, dim3((unsigned)((rows_mask + 255) / 256 > 1024 ? 1024 : (rows_mask + 255) / 256)), dim3(256),
                       0, ctx->stream, rows_mask, c_mask + 1, y_mask, counts + 2);
  PP_CHECK_LAUNCH(ctx, "pp_count_positives");
  return PP_OK;
}

// ---- which 32-row blocks of a head's row space hold an anchor with state 1? (pp_positive_row_blocks) ----
__global__ void positive_blocks_kernel(HeadGeo g, int A, int stride, const float* __restrict__ y_true, unsigned char* __restrict__ flags) {
  const size_t total = (size_t)g.n_img * g.cells_total * A;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (y_true[i * stride + (stride - 1)] != 1.0f) continue;
    size_t t = i / A;
    const int cell = (int)(t % g.cells_total);
    const int b = (int)(t / g.cells_total);
    int s = 0;
    for (int k = 1; k < g.n_seg; ++k)
      if (cell >= g.cell_off[k]) s = k;
    const int m = g.row_begin[s] + b * g.hw[s] + (cell - g.cell_off[s]);
    flags[m >> 5] = 1;
  }
}

extern "C" int pp_positive_row_blocks(pp_ctx* ctx, const pp_rowspace* rs, int A, int stride, const float* y_true, unsigned char* flags) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, rs && y_true && flags && A > 0 && stride > 1 && rs->n_seg >= 1 && rs->n_seg <= PP_MAX_SEG, PP_ERR_ARG, "pp_positive_row_blocks: bad arguments");
  HeadGeo g;
  fill_head_geo(rs, &g);
  const size_t nb = ((size_t)g.row_begin[g.n_seg] + 31) / 32;
  PP_HIP(ctx, hipMemsetAsync(flags, 0, nb, ctx->stream));
  const size_t total = (size_t)g.n_img * g.cells_total * A;
  size_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(positive_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, A, stride, y_true, flags);
  PP_CHECK_LAUNCH(ctx, "pp_positive_row_blocks");
  return PP_OK;
}

// ---- sigmoid focal loss ---------------------------------------------------------------------------
// One thread per (row m, padded channel ch).  ch = a*C + c.
__global__ void focal_kernel(HeadGeo g, int A, int C, const float* __restrict__ logits, int ld, const float* __restrict__ y_true,
                             float alpha, float gamma, const int* __restrict__ count, float loss_weight,
                             float* __restrict__ loss_sum, float* __restrict__ dlogits) {
  const int AC = A * C;
  const size_t total = (size_t)g.row_begin[g.n_seg] * ld;
  const float norm = fmaxf(1.0f, (float)(*count));
  const float eps = 1e-7f, one_m_eps = 1.0f - 1e-7f;
  float lsum = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % ld);
    const int m = (int)(i / ld);
    float grad = 0.f;
    if (ch < AC) {
      int s = 0;
      for (int k = 1; k < g.n_seg; ++k)
        if (m >= g.row_begin[k]) s = k;
      const int local = m - g.row_begin[s];
      const int b = local / g.hw[s];
      const int p = local - b * g.hw[s];
      const int a = ch / C, c = ch - a * C;
      const size_t trow = ((size_t)b * g.cells_total + g.cell_off[s] + p) * A + a;
      const float* yt = y_true + trow * (C + 1);
      const float state = yt[C];
      if (state != -1.0f) {
        const float z = yt[c];
        const float x = logits[i];
        const float pr = 1.0f / (1.0f + expf(-x));
        const bool pos = (z == 1.0f);
        const float alpha_t = pos ? alpha : 1.0f - alpha;
        const float q = pos ? 1.0f - pr : pr;
        float fw, dfw;  // focal weight and d/dp
        if (gamma == 2.0f) {
          fw = alpha_t * q * q;
          dfw = alpha_t * 2.0f * q;
        } else {
          fw = alpha_t * powf(q, gamma);
          dfw = alpha_t * gamma * powf(q, gamma - 1.0f);
        }
        if (pos) dfw = -dfw;
        const float pc = fminf(fmaxf(pr, eps), one_m_eps);
        const bool inside = (pr >= eps) && (pr <= one_m_eps);
        // keras binary_crossentropy on probabilities: clip, logit, sigmoid-CE-with-logits
        const float bce = -(z * logf(pc) + (1.0f - z) * logf(1.0f - pc));
        const float dbce = inside ? (-(z / pc) + (1.0f - z) / (1.0f - pc)) : 0.0f;
        lsum += fw * bce;
        grad = (dfw * bce + fw * dbce) * pr * (1.0f - pr) / norm * loss_weight;
      }
    }
    if (dlogits) dlogits[i] = grad;
  }
  if (loss_sum) block_accumulate(lsum / norm, loss_sum);
}

extern "C" int pp_sigmoid_focal_fwd_bwd(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, int n_class, const float* logits, int ld,
                                        const float* y_true, float alpha, float gamma, const int* count, float loss_weight,
                                        float* loss_sum, float* dlogits) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, rs && pp_rowspace_ok(rs) && logits && y_true && count, PP_ERR_ARG, "pp_sigmoid_focal_fwd_bwd: null argument");
  PP_CHECK_ARG(ctx, n_anchor > 0 && n_class > 0 && ld >= n_anchor * n_class, PP_ERR_SHAPE, "pp_sigmoid_focal_fwd_bwd: ld %d < %d", ld,
               n_anchor * n_class);
  HeadGeo g;
  fill_head_geo(rs, &g);
  size_t total = (size_t)g.row_begin[g.n_seg] * ld;
  size_t blocks = (total + 255) / 256;
  size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(focal_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, n_anchor, n_class, logits, ld, y_true, alpha,
                     gamma, count, loss_weight, loss_sum, dlogits);
  PP_CHECK_LAUNCH(ctx, "pp_sigmoid_focal_fwd_bwd");
  return PP_OK;
}

// ---- smooth-L1 + edge-parallelism L1 on positives -------------------------------------------------
// The 12 (a,b,c,d) index quadruples of losses.py:338-361: feature = (r[a]-r[b]) - (r[c]-r[d]) on the x
// coordinates (even indices); the y feature uses every index + 1.
__constant__ int kOrthQuad[12][4] = {{0, 6, 2, 4},    {0, 6, 8, 14},  {0, 2, 6, 4},   {0, 2, 8, 10},
                                     {0, 8, 2, 10},   {0, 8, 6, 14},  {12, 10, 14, 8}, {12, 10, 4, 2},
                                     {12, 4, 10, 2},  {12, 4, 14, 6}, {12, 14, 4, 6},  {12, 14, 10, 8}};

// One thread per (image, anchor row).
__global__ void orth_l1_kernel(HeadGeo g, int A, const float* __restrict__ pred, int ld, const float* __restrict__ y_true,
                               float weight, float sigma_sq, const int* __restrict__ count, float loss_weight,
                               float* __restrict__ loss_sum, float* __restrict__ dpred) {
  const size_t total = (size_t)g.n_img * g.cells_total * A;
  const float norm = fmaxf(1.0f, (float)(*count));
  const float w_xy = 0.8f, w_orth = 0.2f;
  float lsum = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    size_t t = i / A;
    const int cell = (int)(t % g.cells_total);
    const int b = (int)(t / g.cells_total);
    int s = 0;
    for (int k = 1; k < g.n_seg; ++k)
      if (cell >= g.cell_off[k]) s = k;
    const int m = g.row_begin[s] + b * g.hw[s] + (cell - g.cell_off[s]);
    const float* yt = y_true + i * 17;
    const float* pr = pred + (size_t)m * ld + a * 16;
    float* dp = dpred ? dpred + (size_t)m * ld + a * 16 : nullptr;
    float gout[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) gout[j] = 0.f;
    if (yt[16] == 1.0f) {
      float r[16], tg[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { r[j] = pr[j]; tg[j] = yt[j]; }
      float xy = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float d = r[j] - tg[j];
        const float ad = fabsf(d);
        const bool quad = ad < 1.0f / sigma_sq;
        xy += quad ? 0.5f * sigma_sq * ad * ad : ad - 0.5f / sigma_sq;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        gout[j] = w_xy * (quad ? sigma_sq * d : sgn);
      }
      float orth = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int ia = kOrthQuad[k][0] + o, ib = kOrthQuad[k][1] + o, ic = kOrthQuad[k][2] + o, id = kOrthQuad[k][3] + o;
          const float fp = (r[ia] - r[ib]) - (r[ic] - r[id]);
          const float ft = (tg[ia] - tg[ib]) - (tg[ic] - tg[id]);
          const float e = fp - ft;
          orth += fabsf(e);
          const float sg = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * (w_orth / 24.0f);
          gout[ia] += sg; gout[ib] -= sg; gout[ic] -= sg; gout[id] += sg;
        }
      }
      lsum += w_xy * xy + w_orth * (orth / 24.0f);
    }
    if (dp) {
      const float sc = weight * loss_weight / norm;
#pragma unroll
      for (int j = 0; j < 16; j += 4)
        *reinterpret_cast<float4*>(dp + j) = make_float4(gout[j] * sc, gout[j + 1] * sc, gout[j + 2] * sc, gout[j + 3] * sc);
    }
  }
  if (loss_sum) block_accumulate(weight * lsum / norm, loss_sum);
}

extern "C" int pp_orth_smoothl1_fwd_bwd(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, const float* pred, int ld,
                                        const float* y_true, float weight, float sigma, const int* count, float loss_weight,
                                        float* loss_sum, float* dpred) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, rs && pp_rowspace_ok(rs) && pred && y_true && count, PP_ERR_ARG, "pp_orth_smoothl1_fwd_bwd: null argument");
  PP_CHECK_ARG(ctx, n_anchor > 0 && ld >= n_anchor * 16 && ld % 4 == 0, PP_ERR_SHAPE, "pp_orth_smoothl1_fwd_bwd: ld %d", ld);
  HeadGeo g;
  fill_head_geo(rs, &g);
  if (dpred && ld != n_anchor * 16) {
    // padding columns (never produced by the kernel) must be zero for the data-gradient conv
    PP_HIP(ctx, hipMemsetAsync(dpred, 0, (size_t)g.row_begin[g.n_seg] * ld * sizeof(float), ctx->stream));
  }
  size_t total = (size_t)g.n_img * g.cells_total * n_anchor;
  size_t blocks = (total + 255) / 256;
  size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(orth_l1_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, n_anchor, pred, ld, y_true, weight,
                     sigma * sigma, count, loss_weight, loss_sum, dpred);
  PP_CHECK_LAUNCH(ctx, "pp_orth_smoothl1_fwd_bwd");
  return PP_OK;
}
