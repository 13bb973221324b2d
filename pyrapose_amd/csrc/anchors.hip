// Anchor generation, IoU, target assignment and decode -- HBM-bound integer/index work with the
// reference's exact arithmetic (this file is compiled with -ffp-contract=off: no FMA fusion, so
// float64/float32 results are bit-identical to the numpy / Cython / TF op-by-op evaluation).
//   utils/anchors.py:447-478 generate_anchors      :415-444 shift        :372-412 anchors_for_shape
//   utils/compute_overlap.pyx:13-53                utils/anchors.py:290-318 compute_gt_annotations
//   utils/anchors.py:72-287 anchor_targets_bbox    :515-559 box3D_transform  :562-567 toPix_array
//   layers/_misc.py:60-71 Anchors -> backend/common.py:93-116 shift (float32)
//   layers/_misc.py:195-197 RegressBoxes3D -> backend/common.py:25-56 box3D_transform_inv
//   utils/linemod_eval.py:317-319 score threshold;  layers/filter_detections.py:21-118
#include <math.h>
#include <stdlib.h>

#include "pp_internal.h"

// ------------------------------------------------------------------------------------------ T1 (host)
extern "C" int pp_generate_base_anchors_host(int base_size, const float* ratios, int n_ratios, const float* scales, int n_scales,
                                             double* out) {
  if (!ratios || !scales || !out || n_ratios <= 0 || n_scales <= 0 || base_size <= 0) return PP_ERR_ARG;
  for (int r = 0; r < n_ratios; ++r) {
    for (int s = 0; s < n_scales; ++s) {
      // anchors.py:465: python-int * float32 array is a float32 product, widened on assignment
      volatile float side32 = (float)base_size * scales[s];
      double side = (double)side32;
      double area = side * side;                 // :468
      double ratio = (double)ratios[r];
      double w = sqrt(area / ratio);             // :471
      double h = w * ratio;                      // :472
      double* o = out + 4 * (r * n_scales + s);
      o[0] = 0.0 - w * 0.5;                      // :475
      o[2] = w - w * 0.5;
      o[1] = 0.0 - h * 0.5;                      // :476
      o[3] = h - h * 0.5;
    }
  }
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ T2 / D1
#define PP_MAX_LEVELS 5
#define PP_MAX_BASE 16  // ratios x scales per cell (reference: 9, YCB-V variant: 12)
struct ShiftGeo {
  int n_levels, A;
  int fh[PP_MAX_LEVELS], fw[PP_MAX_LEVELS], stride[PP_MAX_LEVELS];
  int anchor_begin[PP_MAX_LEVELS + 1];
  double base[PP_MAX_LEVELS][PP_MAX_BASE][4];
};

template <typename T>
__global__ void anchors_shift_kernel(const ShiftGeo g, T* __restrict__ out) {
  const int total = g.anchor_begin[g.n_levels];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int l = 0;
    for (int k = 1; k < g.n_levels; ++k)
      if (i >= g.anchor_begin[k]) l = k;
    const int local = i - g.anchor_begin[l];
    const int a = local % g.A;
    const int cell = local / g.A;
    const int x = cell % g.fw[l], y = cell / g.fw[l];
    const T sx = ((T)x + (T)0.5) * (T)g.stride[l];
    const T sy = ((T)y + (T)0.5) * (T)g.stride[l];
    T* o = out + (size_t)i * 4;
    o[0] = (T)g.base[l][a][0] + sx;
    o[1] = (T)g.base[l][a][1] + sy;
    o[2] = (T)g.base[l][a][2] + sx;
    o[3] = (T)g.base[l][a][3] + sy;
  }
}

template <typename T>
static int anchors_shift_impl(pp_ctx* ctx, int n_levels, const int* fh, const int* fw, const int* strides, int A,
                              const double* base, T* out, const char* who) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, fh && fw && strides && base && out, PP_ERR_ARG, "%s: null argument", who);
  PP_CHECK_ARG(ctx, n_levels > 0 && n_levels <= PP_MAX_LEVELS && A > 0 && A <= PP_MAX_BASE, PP_ERR_SHAPE, "%s: bad level/anchor count", who);
  ShiftGeo g;
  memset(&g, 0, sizeof(g));
  g.n_levels = n_levels;
  g.A = A;
  long long tot = 0;
  for (int l = 0; l < n_levels; ++l) {
    PP_CHECK_ARG(ctx, fh[l] > 0 && fw[l] > 0 && strides[l] > 0, PP_ERR_SHAPE, "%s: bad level %d", who, l);
    g.fh[l] = fh[l]; g.fw[l] = fw[l]; g.stride[l] = strides[l];
    g.anchor_begin[l] = (int)tot;
    tot += (long long)fh[l] * fw[l] * A;
    for (int a = 0; a < A; ++a)
      for (int c = 0; c < 4; ++c) g.base[l][a][c] = base[((size_t)l * A + a) * 4 + c];
  }
  PP_CHECK_ARG(ctx, tot < (1ll << 31), PP_ERR_SHAPE, "%s: too many anchors", who);
  g.anchor_begin[n_levels] = (int)tot;
  int blocks = (int)((tot + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((anchors_shift_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, g, out);
  PP_CHECK_LAUNCH(ctx, who);
  return PP_OK;
}

extern "C" int pp_anchors_shift_f64(pp_ctx* ctx, int n_levels, const int* fh, const int* fw, const int* strides, int A,
                                    const double* base, double* out) {
  return anchors_shift_impl<double>(ctx, n_levels, fh, fw, strides, A, base, out, "pp_anchors_shift_f64");
}
extern "C" int pp_anchors_shift_f32(pp_ctx* ctx, int n_levels, const int* fh, const int* fw, const int* strides, int A,
                                    const double* base, float* out) {
  return anchors_shift_impl<float>(ctx, n_levels, fh, fw, strides, A, base, out, "pp_anchors_shift_f32");
}

// ------------------------------------------------------------------------------------------ T3
__device__ __forceinline__ double iou_plus1(double bx1, double by1, double bx2, double by2, double qx1, double qy1, double qx2,
                                            double qy2, double q_area) {
  const double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1;
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1;
  if (!(ih > 0)) return 0.0;
  const double ua = (bx2 - bx1 + 1) * (by2 - by1 + 1) + q_area - iw * ih;
  return iw * ih / ua;
}

__global__ void overlap_kernel(int n, const double* __restrict__ boxes, int k, const double* __restrict__ query,
                               double* __restrict__ out) {
  const long long total = (long long)n * k;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % k);
    const long long b = i / k;
    const double* bb = boxes + b * 4;
    const double* qq = query + (long long)q * 4;
    const double q_area = (qq[2] - qq[0] + 1) * (qq[3] - qq[1] + 1);
    out[i] = iou_plus1(bb[0], bb[1], bb[2], bb[3], qq[0], qq[1], qq[2], qq[3], q_area);
  }
}

extern "C" int pp_compute_overlap_f64(pp_ctx* ctx, int n, const double* boxes, int k, const double* query, double* overlaps) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n >= 0 && k >= 0, PP_ERR_SHAPE, "pp_compute_overlap_f64: negative size");
  if (n == 0 || k == 0) return PP_OK;
  PP_CHECK_ARG(ctx, boxes && query && overlaps, PP_ERR_ARG, "pp_compute_overlap_f64: null tensor");
  long long total = (long long)n * k;
  int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(overlap_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, boxes, k, query, overlaps);
  PP_CHECK_LAUNCH(ctx, "pp_compute_overlap_f64");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ T4
__device__ __forceinline__ void best_gt(const double* a4, int k, const double* __restrict__ gt, int* arg, double* best) {
  // numpy argmax: first maximum wins; row of zeros -> 0
  int am = 0;
  double mx = -INFINITY;
  for (int q = 0; q < k; ++q) {
    const double* qq = gt + (size_t)q * 4;
    const double q_area = (qq[2] - qq[0] + 1) * (qq[3] - qq[1] + 1);
    const double v = iou_plus1(a4[0], a4[1], a4[2], a4[3], qq[0], qq[1], qq[2], qq[3], q_area);
    if (v > mx) { mx = v; am = q; }
  }
  *arg = am;
  *best = mx;
}

__global__ void gt_annotations_kernel(int n, const double* __restrict__ anchors, int k, const double* __restrict__ gt, double neg,
                                      double pos, int* __restrict__ argmax, signed char* __restrict__ state) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int am;
    double mx;
    best_gt(anchors + (size_t)i * 4, k, gt, &am, &mx);
    argmax[i] = am;
    const bool p = mx >= pos;
    state[i] = p ? 1 : ((mx > neg) ? -1 : 0);
  }
}

extern "C" int pp_compute_gt_annotations(pp_ctx* ctx, int n, const double* anchors, int k, const double* gt_boxes,
                                         double negative_overlap, double positive_overlap, int* argmax, signed char* state) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n > 0 && k > 0 && anchors && gt_boxes && argmax && state, PP_ERR_ARG, "pp_compute_gt_annotations: bad argument");
  int blocks = (n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256;
  hipLaunchKernelGGL(gt_annotations_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, anchors, k, gt_boxes, negative_overlap,
                     positive_overlap, argmax, state);
  PP_CHECK_LAUNCH(ctx, "pp_compute_gt_annotations");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ T5 (host helpers)
extern "C" int pp_project_box3d_host(const double* pose, const double* box, const double* cam, double* out16) {
  if (!pose || !box || !cam || !out16) return PP_ERR_ARG;
  // transforms3d 0.3.1 quaternions.quat2mat (third-party; published algorithm), q = (w, x, y, z)
  const double w = pose[3], x = pose[4], y = pose[5], z = pose[6];
  const double Nq = w * w + x * x + y * y + z * z;
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (!(Nq < 2.220446049250313e-16)) {
    const double s = 2.0 / Nq;
    const double X = x * s, Y = y * s, Z = z * s;
    const double wX = w * X, wY = w * Y, wZ = w * Z;
    const double xX = x * X, xY = x * Y, xZ = x * Z;
    const double yY = y * Y, yZ = y * Z, zZ = z * Z;
    R[0][0] = 1.0 - (yY + zZ); R[0][1] = xY - wZ;         R[0][2] = xZ + wY;
    R[1][0] = xY + wZ;         R[1][1] = 1.0 - (xX + zZ); R[1][2] = yZ - wX;
    R[2][0] = xZ - wY;         R[2][1] = yZ + wX;         R[2][2] = 1.0 - (xX + yY);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = (double)(float)R[i][j];  // anchors.py:208 float32 rounding
  for (int c = 0; c < 8; ++c) {
    const double* b = box + 3 * c;
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = (R[i][0] * b[0] + R[i][1] * b[1] + R[i][2] * b[2]) + pose[i];  // :210-211
    out16[2 * c + 0] = (t[0] * cam[0]) / t[2] + cam[2];  // :564
    out16[2 * c + 1] = (t[1] * cam[1]) / t[2] + cam[3];  // :565
  }
  return PP_OK;
}

extern "C" int pp_pil_nearest_index_host(int n_in, int n_out, int* out) {
  if (!out || n_in <= 0 || n_out <= 0) return PP_ERR_ARG;
  // Pillow ImagingScaleAffine (NEAREST): start at scale/2, truncate, accumulate += scale in double
  const double scale = (double)n_in / (double)n_out;
  double xo = scale * 0.5;
  for (int i = 0; i < n_out; ++i) {
    int v = (int)xo;
    out[i] = v < n_in - 1 ? v : n_in - 1;
    xo += scale;
  }
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ T5 (device)
#define PP_MAX_BATCH 64
#define PP_MAX_MASK_DIM 160
struct TargetTables {  // small host tables travel as kernel arguments: no staging copy, no sync
  int gt_offset[PP_MAX_BATCH + 1];
  int image_hw[2 * PP_MAX_BATCH];
  int mask_hw[2 * PP_MAX_BATCH];  // valid (h, w) of each id mask inside the padded [mask_h, mask_w] plane
};

struct TargetArgs {
  int N, B, C, G;
  const double* anchors;
  const double* gt_boxes;  // [G,4]
  const int* gt_labels;    // [G]
  const double* gt_box3d;  // [G,16]
  double neg, pos;
  float* regression;       // [B,N,17]
  float* labels;           // [B,N,C+1]
};

// one thread per (image, anchor): writes the whole 17-float regression row and the (C+1)-float label row
__global__ void anchor_targets_kernel(const TargetArgs t, const TargetTables tb) {
  const long long total = (long long)t.B * t.N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / t.N);
    const int n = (int)(i - (long long)b * t.N);
    const double* a = t.anchors + (size_t)n * 4;
    const int g0 = tb.gt_offset[b], g1 = tb.gt_offset[b + 1];
    float* reg = t.regression + (size_t)i * 17;
    float* lab = t.labels + (size_t)i * (t.C + 1);
    float state = 0.f;
    int cls = -1;
    if (g1 > g0) {
      int am;
      double mx;
      best_gt(a, g1 - g0, t.gt_boxes + (size_t)g0 * 4, &am, &mx);
      const bool p = mx >= t.pos;
      state = p ? 1.f : ((mx > t.neg) ? -1.f : 0.f);
      if (p) cls = t.gt_labels[g0 + am];
      // box3D_transform (anchors.py:533-556): (gt - anchor_{x1,y1,x2,y2 alternately}) / (w|h), then (.. - 0) / 0.2
      const double aw = a[2] - a[0], ah = a[3] - a[1];
      const double* gt = t.gt_box3d + (size_t)(g0 + am) * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double d = (gt[j] - a[j & 3]) / ((j & 1) ? ah : aw);
        reg[j] = (float)((d - 0.0) / 0.2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) reg[j] = 0.f;
    }
    // anchors.py:279-285: centre outside the unpadded image -> ignore
    const double cx = (a[0] + a[2]) / 2, cy = (a[1] + a[3]) / 2;
    if (cx >= (double)tb.image_hw[2 * b + 1] || cy >= (double)tb.image_hw[2 * b + 0]) state = -1.f;
    reg[16] = state;
    for (int c = 0; c < t.C; ++c) lab[c] = (c == cls) ? 1.f : 0.f;
    lab[t.C] = state;
  }
}

// mask targets (anchors.py:156-164): one workgroup per (image, gt): count matching level-3 cells, then mark.
// The PIL NEAREST index maps (Pillow ImagingScaleAffine: start at scale/2, truncate, accumulate += scale in
// double) are rebuilt per image in LDS with the same accumulation order as pp_pil_nearest_index_host.
__global__ void mask_targets_kernel(int C, const TargetTables tb, const int* __restrict__ gt_labels,
                                    const int* __restrict__ gt_mask_ids, const unsigned char* __restrict__ id_masks, int mask_h,
                                    int mask_w, int mh, int mw, float* __restrict__ mask_out) {
  const int b = blockIdx.y;
  const int g = tb.gt_offset[b] + blockIdx.x;
  if (g >= tb.gt_offset[b + 1]) return;
  const int id = gt_mask_ids[g], cls = gt_labels[g];
  const unsigned char* img = id_masks + (size_t)b * mask_h * mask_w;
  const int cells = mh * mw;
  __shared__ int s_cnt;
  __shared__ int s_row[PP_MAX_MASK_DIM], s_col[PP_MAX_MASK_DIM];
  if (threadIdx.x == 0) s_cnt = 0;
  for (int t = threadIdx.x; t < mh + mw; t += blockDim.x) {
    const bool is_row = t < mh;
    const int i = is_row ? t : t - mh;
    const int n_in = is_row ? tb.mask_hw[2 * b] : tb.mask_hw[2 * b + 1];
    const double scale = (double)n_in / (double)(is_row ? mh : mw);
    double xo = scale * 0.5;
    for (int k = 0; k < i; ++k) xo += scale;
    int v = (int)xo;
    v = v < n_in - 1 ? v : n_in - 1;
    if (is_row) s_row[i] = v; else s_col[i] = v;
  }
  __syncthreads();
  int cnt = 0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x) {
    const int y = c / mw, x = c - y * mw;
    cnt += ((int)img[(size_t)s_row[y] * mask_w + s_col[x]] == id) ? 1 : 0;
  }
  if (cnt) atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (s_cnt <= 1) return;  // `if len(anchors_spec) > 1`
  float* out = mask_out + (size_t)b * cells * (C + 1);
  for (int c = threadIdx.x; c < cells; c += blockDim.x) {
    const int y = c / mw, x = c - y * mw;
    if ((int)img[(size_t)s_row[y] * mask_w + s_col[x]] == id) {
      out[(size_t)c * (C + 1) + cls] = 1.f;
      out[(size_t)c * (C + 1) + C] = 1.f;
    }
  }
}

extern "C" int pp_anchor_targets(pp_ctx* ctx, int n_anchor_total, const double* anchors, int batch, const int* gt_offset_host,
                                 const double* gt_boxes, const int* gt_labels, const double* gt_box3d, const int* gt_mask_ids,
                                 const unsigned char* id_masks, int mask_h, int mask_w, const int* mask_hw_host,
                                 const int* image_hw_host, int num_classes, double negative_overlap, double positive_overlap,
                                 int out_mh, int out_mw, float* regression, float* labels, float* mask) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_anchor_total > 0 && batch > 0 && num_classes > 0 && anchors && gt_offset_host && image_hw_host && regression &&
                        labels,
               PP_ERR_ARG, "pp_anchor_targets: bad argument");
  PP_CHECK_ARG(ctx, gt_offset_host[0] == 0, PP_ERR_SHAPE, "pp_anchor_targets: gt_offset[0] must be 0");
  int max_g = 0;
  for (int b = 0; b < batch; ++b) {
    int k = gt_offset_host[b + 1] - gt_offset_host[b];
    PP_CHECK_ARG(ctx, k >= 0, PP_ERR_SHAPE, "pp_anchor_targets: gt_offset must be non-decreasing");
    if (k > max_g) max_g = k;
  }
  const int G = gt_offset_host[batch];
  PP_CHECK_ARG(ctx, G == 0 || (gt_boxes && gt_labels && gt_box3d), PP_ERR_ARG, "pp_anchor_targets: null ground truth");
  const bool do_mask = mask != nullptr;
  if (do_mask) PP_CHECK_ARG(ctx, out_mh > 0 && out_mw > 0 && (G == 0 || (id_masks && gt_mask_ids && mask_h > 0 && mask_w > 0)), PP_ERR_ARG,
                            "pp_anchor_targets: bad mask arguments");
  PP_CHECK_ARG(ctx, batch <= PP_MAX_BATCH, PP_ERR_SHAPE, "pp_anchor_targets: batch %d > %d", batch, PP_MAX_BATCH);
  PP_CHECK_ARG(ctx, !do_mask || (out_mh <= PP_MAX_MASK_DIM && out_mw <= PP_MAX_MASK_DIM), PP_ERR_SHAPE,
               "pp_anchor_targets: mask level %dx%d too large", out_mh, out_mw);
  TargetTables tb;
  memset(&tb, 0, sizeof(tb));
  memcpy(tb.gt_offset, gt_offset_host, (batch + 1) * sizeof(int));
  memcpy(tb.image_hw, image_hw_host, 2 * batch * sizeof(int));
  if (do_mask && G > 0) {
    for (int b = 0; b < batch; ++b) {
      tb.mask_hw[2 * b] = mask_hw_host ? mask_hw_host[2 * b] : mask_h;
      tb.mask_hw[2 * b + 1] = mask_hw_host ? mask_hw_host[2 * b + 1] : mask_w;
      PP_CHECK_ARG(ctx, tb.mask_hw[2 * b] > 0 && tb.mask_hw[2 * b] <= mask_h && tb.mask_hw[2 * b + 1] > 0 && tb.mask_hw[2 * b + 1] <= mask_w,
                   PP_ERR_SHAPE, "pp_anchor_targets: mask_hw of image %d outside the padded plane", b);
    }
  }

  TargetArgs t;
  t.N = n_anchor_total; t.B = batch; t.C = num_classes; t.G = G;
  t.anchors = anchors;
  t.gt_boxes = gt_boxes; t.gt_labels = gt_labels; t.gt_box3d = gt_box3d;
  t.neg = negative_overlap; t.pos = positive_overlap;
  t.regression = regression; t.labels = labels;
  long long total = (long long)batch * n_anchor_total;
  int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(anchor_targets_kernel, dim3(blocks), dim3(256), 0, ctx->stream, t, tb);
  if (do_mask) {
    PP_HIP(ctx, hipMemsetAsync(mask, 0, (size_t)batch * out_mh * out_mw * (num_classes + 1) * sizeof(float), ctx->stream));
    if (max_g > 0)
      hipLaunchKernelGGL(mask_targets_kernel, dim3(max_g, batch), dim3(256), 0, ctx->stream, num_classes, tb, gt_labels,
                         gt_mask_ids, id_masks, mask_h, mask_w, out_mh, out_mw, mask);
  }
  PP_CHECK_LAUNCH(ctx, "pp_anchor_targets");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ D2
__global__ void box3d_decode_kernel(long long rows, int n, const float* __restrict__ anchors, const float* __restrict__ reg,
                                    float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x) {
    const float* a = anchors + (size_t)(i % n) * 4;
    const float w = a[2] - a[0], h = a[3] - a[1];
    const float* r = reg + (size_t)i * 16;
    float* o = out + (size_t)i * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // TF evaluates op by op in float32: (delta*std + mean) * size, then + anchor coordinate
      float tmp = r[j] * 0.2f;
      tmp = tmp + 0.0f;
      tmp = tmp * ((j & 1) ? h : w);
      o[j] = a[j & 3] + tmp;
    }
  }
}

extern "C" int pp_box3d_decode(pp_ctx* ctx, int batch, int n, const float* anchors, const float* regression, float* boxes3d) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, batch > 0 && n > 0 && anchors && regression && boxes3d, PP_ERR_ARG, "pp_box3d_decode: bad argument");
  long long rows = (long long)batch * n;
  int blocks = (int)((rows + 255) / 256 > 4096 ? 4096 : (rows + 255) / 256);
  hipLaunchKernelGGL(box3d_decode_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rows, n, anchors, regression, boxes3d);
  PP_CHECK_LAUNCH(ctx, "pp_box3d_decode");
  return PP_OK;
}

// ------------------------------------------------------------------------------------------ D3
// one workgroup per (image, class): ordered (stable) compaction via ballot prefix sums
__global__ void threshold_compact_kernel(int n, int C, const float* __restrict__ scores, float thr, int cap,
                                         int* __restrict__ idx_out, int* __restrict__ counts) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float* s = scores + (size_t)b * n * C + c;
  int* out = idx_out + ((size_t)b * C + c) * cap;
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < n; start += blockDim.x) {
    const int i = start + threadIdx.x;
    const bool hit = (i < n) && (s[(size_t)i * C] > thr);
    const unsigned long long bal = __ballot(hit);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (hit && off + before < cap) out[off + before] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < n_waves; ++w) t += wave_cnt[w];
      base += t;
    }
    __syncthreads();
  }
  const int total = base;
  if (threadIdx.x == 0) counts[b * C + c] = total;
  for (int i = (total < cap ? total : cap) + threadIdx.x; i < cap; i += blockDim.x) out[i] = -1;
}

extern "C" int pp_score_threshold_compact(pp_ctx* ctx, int batch, int n, int n_class, const float* scores, float thr, int cap,
                                          int* idx_out, int* counts) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, batch > 0 && n > 0 && n_class > 0 && cap > 0 && scores && idx_out && counts, PP_ERR_ARG,
               "pp_score_threshold_compact: bad argument");
  hipLaunchKernelGGL(threshold_compact_kernel, dim3(n_class, batch), dim3(1024), 0, ctx->stream, n, n_class, scores, thr, cap, idx_out,
                     counts);
  PP_CHECK_LAUNCH(ctx, "pp_score_threshold_compact");
  return PP_OK;
}
