// filter_detections (layers/filter_detections.py:21-118): score threshold -> per-class greedy NMS
// (tf.image.non_max_suppression, TF 2.1, third-party) -> concat classes -> top_k -> gather -> pad -1.
// The layer is registered but never instantiated by the reference graph; it is built
// because the north star names NMS.  Tie-breaks: higher score first, equal score -> lower index first
// ("parity unpinned").  Compiled with -ffp-contract=off (IoU compares are op-by-op f32).
#include "pp_internal.h"

struct FilterWs {
  unsigned long long* keys;  // [C][npow2] sort keys: (~score_bits << 32) | anchor index
  int* cls_count;            // [C] candidates per class
  int* sel_idx;              // [C][max_det] NMS survivors (anchor index), in selection order
  int* sel_count;            // [C]
};

static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

extern "C" size_t pp_filter_workspace_bytes(int n, int n_class, int max_det) {
  if (n <= 0 || n_class <= 0 || max_det <= 0) return 0;
  size_t np2 = (size_t)next_pow2(n);
  return (size_t)n_class * np2 * 8 + (size_t)n_class * 4 * 2 + (size_t)n_class * max_det * 4 + 256;
}

__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
  unsigned int b = __float_as_uint(score);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotone map of float order onto unsigned order
  return ((unsigned long long)(~b) << 32) | (unsigned int)idx;
}
__device__ __forceinline__ float key_score(unsigned long long k) {
  unsigned int b = ~(unsigned int)(k >> 32);
  b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
  return __uint_as_float(b);
}

// one workgroup per class: stable compaction of score > thr, then bitonic sort (score desc, index asc)
// (blockIdx.y = image: every per-image tensor and workspace section is advanced by its image stride)
__global__ void filter_sort_kernel(int n, int C, int npow2, const float* __restrict__ scores, float thr,
                                   unsigned long long* __restrict__ keys_all, int* __restrict__ cls_count) {
  const int c = blockIdx.x;
  scores += (size_t)blockIdx.y * n * C;
  keys_all += (size_t)blockIdx.y * C * npow2;
  cls_count += (size_t)blockIdx.y * C;
  unsigned long long* keys = keys_all + (size_t)c * npow2;
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < n; start += blockDim.x) {
    const int i = start + threadIdx.x;
    float s = 0.f;
    bool hit = false;
    if (i < n) {
      s = scores[(size_t)i * C + c];
      hit = s > thr;
    }
    const unsigned long long bal = __ballot(hit);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (hit) keys[off + before] = make_key(s, i);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < n_waves; ++w) t += wave_cnt[w];
      base += t;
    }
    __syncthreads();
  }
  const int cnt = base;
  if (threadIdx.x == 0) cls_count[c] = cnt;
  int len = 1;
  while (len < cnt) len <<= 1;
  for (int i = cnt + threadIdx.x; i < len; i += blockDim.x) keys[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= len; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool up = ((i & k) == 0);
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ float tf_iou(const float* a, const float* b) {
  // tensorflow/core/kernels/non_max_suppression_op.cc IOU(): coordinate order agnostic
  const float ay0 = fminf(a[0], a[2]), ax0 = fminf(a[1], a[3]), ay1 = fmaxf(a[0], a[2]), ax1 = fmaxf(a[1], a[3]);
  const float by0 = fminf(b[0], b[2]), bx0 = fminf(b[1], b[3]), by1 = fmaxf(b[0], b[2]), bx1 = fmaxf(b[1], b[3]);
  const float area_a = (ay1 - ay0) * (ax1 - ax0), area_b = (by1 - by0) * (bx1 - bx0);
  if (area_a <= 0.f || area_b <= 0.f) return 0.f;
  const float iy0 = fmaxf(ay0, by0), ix0 = fmaxf(ax0, bx0), iy1 = fminf(ay1, by1), ix1 = fminf(ax1, bx1);
  const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
  return inter / (area_a + area_b - inter);
}

// one workgroup per class: greedy NMS over the sorted candidates, at most max_det survivors
__global__ void filter_nms_kernel(int n, int npow2, const float* __restrict__ boxes, float iou_thr, int max_det,
                                  const unsigned long long* __restrict__ keys_all, const int* __restrict__ cls_count,
                                  int* __restrict__ sel_idx_all, int* __restrict__ sel_count) {
  const int c = blockIdx.x, C = gridDim.x;
  boxes += (size_t)blockIdx.y * n * 4;
  keys_all += (size_t)blockIdx.y * C * npow2;
  cls_count += (size_t)blockIdx.y * C;
  sel_idx_all += (size_t)blockIdx.y * C * max_det;
  sel_count += (size_t)blockIdx.y * C;
  const unsigned long long* keys = keys_all + (size_t)c * npow2;
  int* sel_idx = sel_idx_all + (size_t)c * max_det;
  extern __shared__ float s_boxes[];  // [max_det][4]
  const int cnt = cls_count[c];
  int n_sel = 0;  // uniform across the workgroup
  for (int i = 0; i < cnt && n_sel < max_det; ++i) {
    const int idx = (int)(keys[i] & 0xffffffffu);
    const float* cand = boxes + (size_t)idx * 4;
    int sup = 0;
    for (int j = threadIdx.x; j < n_sel; j += blockDim.x) sup |= (tf_iou(cand, s_boxes + 4 * j) > iou_thr) ? 1 : 0;
    sup = __syncthreads_or(sup);
    if (!sup) {
      if (threadIdx.x < 4) s_boxes[4 * n_sel + threadIdx.x] = cand[threadIdx.x];
      if (threadIdx.x == 0) sel_idx[n_sel] = idx;
      ++n_sel;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sel_count[c] = n_sel;
}

// single workgroup: concat classes, top_k by score (ties: earlier position first), gather, pad with -1
__global__ void filter_topk_kernel(int n, int C, int max_det, const float* __restrict__ boxes, const float* __restrict__ boxes3d,
                                   const float* __restrict__ scores, const int* __restrict__ sel_idx_all,
                                   const int* __restrict__ sel_count, float* __restrict__ out_boxes,
                                   float* __restrict__ out_boxes3d, float* __restrict__ out_scores, int* __restrict__ out_labels) {
  {
    const size_t b = blockIdx.x;
    boxes += b * n * 4; boxes3d += b * n * 16; scores += b * n * C;
    sel_idx_all += b * C * max_det; sel_count += b * C;
    out_boxes += b * max_det * 4; out_boxes3d += b * max_det * 16; out_scores += b * max_det; out_labels += b * max_det;
  }
  extern __shared__ unsigned long long s_keys[];  // [len] (score, position); then position -> (class, slot)
  __shared__ int s_off[64 + 1];
  if (threadIdx.x == 0) {
    int o = 0;
    for (int c = 0; c < C; ++c) { s_off[c] = o; o += sel_count[c]; }
    s_off[C] = o;
  }
  __syncthreads();
  const int total = s_off[C];
  int len = 1;
  while (len < total) len <<= 1;
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < total) {
      int c = 0;
      while (i >= s_off[c + 1]) ++c;
      const int idx = sel_idx_all[(size_t)c * max_det + (i - s_off[c])];
      k = make_key(scores[(size_t)idx * C + c], i);
    }
    s_keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= len; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = s_keys[i], b = s_keys[l];
          const bool up = ((i & k) == 0);
          if ((a > b) == up) { s_keys[i] = b; s_keys[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  const int keep = total < max_det ? total : max_det;
  for (int r = threadIdx.x; r < max_det; r += blockDim.x) {
    if (r < keep) {
      const int pos = (int)(s_keys[r] & 0xffffffffu);
      int c = 0;
      while (pos >= s_off[c + 1]) ++c;
      const int idx = sel_idx_all[(size_t)c * max_det + (pos - s_off[c])];
      for (int j = 0; j < 4; ++j) out_boxes[4 * r + j] = boxes[(size_t)idx * 4 + j];
      for (int j = 0; j < 16; ++j) out_boxes3d[16 * r + j] = boxes3d[(size_t)idx * 16 + j];
      out_scores[r] = key_score(s_keys[r]);
      out_labels[r] = c;
    } else {
      for (int j = 0; j < 4; ++j) out_boxes[4 * r + j] = -1.f;
      for (int j = 0; j < 16; ++j) out_boxes3d[16 * r + j] = -1.f;
      out_scores[r] = -1.f;
      out_labels[r] = -1;
    }
  }
}

extern "C" int pp_filter_detections_batch(pp_ctx* ctx, int n_img, int n, int n_class, const float* boxes, const float* boxes3d,
                                          const float* scores, float score_thr, float iou_thr, int max_det, void* workspace,
                                          float* out_boxes, float* out_boxes3d, float* out_scores, int* out_labels) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_img > 0 && n_img <= 65535 && n > 0 && n_class > 0 && n_class <= 64 && max_det > 0 && max_det <= 1024, PP_ERR_SHAPE,
               "pp_filter_detections: unsupported size (images <= 65535, classes <= 64, max_det <= 1024)");
  PP_CHECK_ARG(ctx, boxes && boxes3d && scores && workspace && out_boxes && out_boxes3d && out_scores && out_labels, PP_ERR_ARG,
               "pp_filter_detections: null argument");
  const int np2 = next_pow2(n);
  char* w = (char*)workspace;  // sections are [image][...], each image's block as pp_filter_workspace_bytes lays it out
  FilterWs ws;
  ws.keys = (unsigned long long*)w;
  w += (size_t)n_img * n_class * np2 * 8;
  ws.cls_count = (int*)w;
  w += (size_t)n_img * n_class * 4;
  ws.sel_count = (int*)w;
  w += (size_t)n_img * n_class * 4;
  ws.sel_idx = (int*)w;
  const int tot_max = next_pow2(n_class * max_det);
  PP_CHECK_ARG(ctx, (size_t)tot_max * 8 <= 64 * 1024, PP_ERR_SHAPE, "pp_filter_detections: classes*max_det too large for the top-k stage");
  hipLaunchKernelGGL(filter_sort_kernel, dim3(n_class, n_img), dim3(1024), 0, ctx->stream, n, n_class, np2, scores, score_thr, ws.keys,
                     ws.cls_count);
  hipLaunchKernelGGL(filter_nms_kernel, dim3(n_class, n_img), dim3(256), (size_t)max_det * 16, ctx->stream, n, np2, boxes, iou_thr, max_det,
                     (const unsigned long long*)ws.keys, (const int*)ws.cls_count, ws.sel_idx, ws.sel_count);
  hipLaunchKernelGGL(filter_topk_kernel, dim3(n_img), dim3(1024), (size_t)tot_max * 8, ctx->stream, n, n_class, max_det, boxes, boxes3d,
                     scores, (const int*)ws.sel_idx, (const int*)ws.sel_count, out_boxes, out_boxes3d, out_scores, out_labels);
  PP_CHECK_LAUNCH(ctx, "pp_filter_detections");
  return PP_OK;
}

extern "C" int pp_filter_detections(pp_ctx* ctx, int n, int n_class, const float* boxes, const float* boxes3d, const float* scores,
                                    float score_thr, float iou_thr, int max_det, void* workspace, float* out_boxes,
                                    float* out_boxes3d, float* out_scores, int* out_labels) {
  return pp_filter_detections_batch(ctx, 1, n, n_class, boxes, boxes3d, scores, score_thr, iou_thr, max_det, workspace, out_boxes,
                                    out_boxes3d, out_scores, out_labels);
}
