// Geometric augmentation of the input pipeline on the device: the reference calls OpenCV for these --
//   utils/image.py:150-216  apply_transform:      cv2.warpAffine(image, M[:2], INTER_LINEAR, border from TransformParameters)
//   utils/image.py:219-230  apply_transform2mask: cv2.warpAffine(mask,  M[:2], INTER_NEAREST, BORDER_CONSTANT 0)
//   utils/image.py:307-323  resize_image:         cv2.resize(img, None, fx=scale, fy=scale)   (INTER_LINEAR)
// OpenCV (opencv-python, unpinned in the reference's setup.py) is a third-party dependency that is neither in the reference
// tree nor installed here: "parity unpinned".  These kernels follow OpenCV 4's PUBLISHED fixed-point scheme for 8-bit
// images -- all integer arithmetic, so the HIP result is bit-exact against oracle/image_np.py, which restates the same
// scheme in numpy:
//   warpAffine: the 2x3 matrix is inverted in double; per destination pixel the source position is
//     X = (round(M00*x*1024) + round((M01*y + M02)*1024) + delta) >> 5   (1/32-pixel units; delta = 16, or 512 and >> 10 for nearest)
//     and the four neighbours are blended with 15-bit weights (32-ax)(32-ay)*32 ..., (sum + 2^14) >> 15.
//   resize: source position (dx + 0.5)/scale - 0.5 (float), 11-bit weights, two passes with the 4 + 16 + 2 bit shifts of
//     cv::VResizeLinear.
// Compiled with -ffp-contract=off: the double / float index arithmetic must round like the host's.
#include "pp_internal.h"

static inline unsigned img_grid(size_t n, pp_ctx* ctx) {
  size_t blocks = (n + 255) / 256;
  const size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 16;
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

struct WarpMat { double m[6]; };  // the INVERSE map (dst -> src), as cv::warpAffine computes it
#define PP_WARP_MAX_IMG 64
struct WarpMats { WarpMat a[PP_WARP_MAX_IMG]; };

__device__ __forceinline__ int sat_i32(double v) {  // cv::saturate_cast<int>(double) = cvRound (round half to even), saturated
  const double r = rint(v);
  return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (-2147483647 - 1) : (int)r);
}
__device__ __forceinline__ int sat_i16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

// border: 0 = BORDER_CONSTANT (cval), 1 = BORDER_REPLICATE
template <int CN>
__global__ void warp_affine_linear_u8_kernel(int n_img, int H, int W, const WarpMats mats, int border, int cval, const unsigned char* __restrict__ src,
                                             unsigned char* __restrict__ dst) {
  const size_t total = (size_t)n_img * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((size_t)W * H));
    const double* M = mats.a[n].m;
    const int X0 = sat_i32((M[1] * y + M[2]) * 1024.0) + 16, Y0 = sat_i32((M[4] * y + M[5]) * 1024.0) + 16;
    const int X = (X0 + sat_i32(M[0] * x * 1024.0)) >> 5, Y = (Y0 + sat_i32(M[3] * x * 1024.0)) >> 5;
    const int sx = sat_i16(X >> 5), sy = sat_i16(Y >> 5), ax = X & 31, ay = Y & 31;
    const int w0 = (32 - ay) * (32 - ax) * 32, w1 = (32 - ay) * ax * 32, w2 = ay * (32 - ax) * 32, w3 = ay * ax * 32;
    const unsigned char* S = src + (size_t)n * H * W * CN;
    unsigned char* D = dst + i * CN;
    if ((unsigned)sx < (unsigned)(W - 1) && (unsigned)sy < (unsigned)(H - 1)) {
      const unsigned char* p = S + ((size_t)sy * W + sx) * CN;
#pragma unroll
      for (int c = 0; c < CN; ++c)
        D[c] = (unsigned char)((p[c] * w0 + p[c + CN] * w1 + p[(size_t)W * CN + c] * w2 + p[(size_t)W * CN + CN + c] * w3 + (1 << 14)) >> 15);
    } else if (border == 0 && (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
#pragma unroll
      for (int c = 0; c < CN; ++c) D[c] = (unsigned char)cval;
    } else {
      int xs[2], ys[2];
      bool okx[2], oky[2];
      if (border == 1) {
        xs[0] = clipi(sx, 0, W); xs[1] = clipi(sx + 1, 0, W); ys[0] = clipi(sy, 0, H); ys[1] = clipi(sy + 1, 0, H);
        okx[0] = okx[1] = oky[0] = oky[1] = true;
      } else {
        xs[0] = sx; xs[1] = sx + 1; ys[0] = sy; ys[1] = sy + 1;
        okx[0] = (unsigned)sx < (unsigned)W; okx[1] = (unsigned)(sx + 1) < (unsigned)W;
        oky[0] = (unsigned)sy < (unsigned)H; oky[1] = (unsigned)(sy + 1) < (unsigned)H;
      }
#pragma unroll
      for (int c = 0; c < CN; ++c) {
        const int v0 = (okx[0] && oky[0]) ? S[((size_t)ys[0] * W + xs[0]) * CN + c] : cval;
        const int v1 = (okx[1] && oky[0]) ? S[((size_t)ys[0] * W + xs[1]) * CN + c] : cval;
        const int v2 = (okx[0] && oky[1]) ? S[((size_t)ys[1] * W + xs[0]) * CN + c] : cval;
        const int v3 = (okx[1] && oky[1]) ? S[((size_t)ys[1] * W + xs[1]) * CN + c] : cval;
        D[c] = (unsigned char)((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
      }
    }
  }
}

__global__ void warp_affine_nearest_u8_kernel(int n_img, int H, int W, const WarpMats mats, int border, int cval, const unsigned char* __restrict__ src,
                                              unsigned char* __restrict__ dst) {
  const size_t total = (size_t)n_img * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((size_t)W * H));
    const double* M = mats.a[n].m;
    const int X0 = sat_i32((M[1] * y + M[2]) * 1024.0) + 512, Y0 = sat_i32((M[4] * y + M[5]) * 1024.0) + 512;
    int sx = sat_i16((X0 + sat_i32(M[0] * x * 1024.0)) >> 10), sy = sat_i16((Y0 + sat_i32(M[3] * x * 1024.0)) >> 10);
    const unsigned char* S = src + (size_t)n * H * W;
    int v;
    if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H) v = S[(size_t)sy * W + sx];
    else if (border == 1) v = S[(size_t)clipi(sy, 0, H) * W + clipi(sx, 0, W)];
    else v = cval;
    dst[i] = (unsigned char)v;
  }
}

static int invert_affine(const double* m, double* o) {  // cv::warpAffine (no WARP_INVERSE_MAP): invert M in double
  double D = m[0] * m[4] - m[1] * m[3];
  D = D != 0 ? 1.0 / D : 0.0;
  const double A11 = m[4] * D, A22 = m[0] * D;
  o[0] = A11; o[1] = m[1] * (-D); o[3] = m[3] * (-D); o[4] = A22;
  const double b1 = -o[0] * m[2] - o[1] * m[5], b2 = -o[3] * m[2] - o[4] * m[5];
  o[2] = b1; o[5] = b2;
  return 0;
}

static int fill_mats(pp_ctx* ctx, int n_img, const double* mats_host, WarpMats* out, const char* who) {
  PP_CHECK_ARG(ctx, n_img > 0 && n_img <= PP_WARP_MAX_IMG && mats_host, PP_ERR_ARG, "%s: 1..%d images, matrices on the host", who, PP_WARP_MAX_IMG);
  for (int i = 0; i < n_img; ++i) invert_affine(mats_host + 6 * i, out->a[i].m);
  return PP_OK;
}

extern "C" int pp_warp_affine_u8(pp_ctx* ctx, int n_img, int H, int W, int channels, const double* mats_host, int interpolation, int border,
                                 int cval, const unsigned char* src, unsigned char* dst) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && dst && src != dst && H > 1 && W > 1 && H < 32768 && W < 32768, PP_ERR_ARG, "pp_warp_affine_u8: bad image");
  PP_CHECK_ARG(ctx, (interpolation == 0 && channels == 1) || (interpolation == 1 && (channels == 1 || channels == 3)), PP_ERR_ARG,
               "pp_warp_affine_u8: interpolation 0 (nearest, 1 channel) or 1 (linear, 1 or 3 channels)");
  PP_CHECK_ARG(ctx, (border == 0 || border == 1) && cval >= 0 && cval <= 255, PP_ERR_ARG, "pp_warp_affine_u8: border 0 (constant) / 1 (replicate)");
  WarpMats mats;
  int rc = fill_mats(ctx, n_img, mats_host, &mats, "pp_warp_affine_u8");
  if (rc) return rc;
  const size_t total = (size_t)n_img * H * W;
  if (interpolation == 0)
    hipLaunchKernelGGL(warp_affine_nearest_u8_kernel, dim3(img_grid(total, ctx)), dim3(256), 0, ctx->stream, n_img, H, W, mats, border, cval, src, dst);
  else if (channels == 3)
    hipLaunchKernelGGL((warp_affine_linear_u8_kernel<3>), dim3(img_grid(total, ctx)), dim3(256), 0, ctx->stream, n_img, H, W, mats, border, cval, src,
                       dst);
  else
    hipLaunchKernelGGL((warp_affine_linear_u8_kernel<1>), dim3(img_grid(total, ctx)), dim3(256), 0, ctx->stream, n_img, H, W, mats, border, cval, src,
                       dst);
  PP_CHECK_LAUNCH(ctx, "pp_warp_affine_u8");
  return PP_OK;
}

// ---- cv2.resize(img, None, fx = fy = scale), INTER_LINEAR, uint8 ----
template <int CN>
__global__ void resize_linear_u8_kernel(int n_img, int SH, int SW, int DH, int DW, double scale_x, double scale_y, const unsigned char* __restrict__ src,
                                        unsigned char* __restrict__ dst) {
  const size_t total = (size_t)n_img * DH * DW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int dx = (int)(i % DW), dy = (int)((i / DW) % DH), n = (int)(i / ((size_t)DW * DH));
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= SW - 1) { fx = 0.f; sx = SW - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0.f; sy = 0; }
    if (sy >= SH - 1) { fy = 0.f; sy = SH - 1; }
    // cv: coefficients saturate_cast<short>(w * 2048) (round half to even); right / bottom neighbour clamped to the last pixel
    const int a0 = (int)rintf((1.f - fx) * 2048.f), a1 = (int)rintf(fx * 2048.f);
    const int b0 = (int)rintf((1.f - fy) * 2048.f), b1 = (int)rintf(fy * 2048.f);
    const int sx1 = sx + 1 < SW ? sx + 1 : SW - 1, sy1 = sy + 1 < SH ? sy + 1 : SH - 1;
    const unsigned char* S = src + (size_t)n * SH * SW * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      const int r0 = S[((size_t)sy * SW + sx) * CN + c] * a0 + S[((size_t)sy * SW + sx1) * CN + c] * a1;   // horizontal pass, row sy
      const int r1 = S[((size_t)sy1 * SW + sx) * CN + c] * a0 + S[((size_t)sy1 * SW + sx1) * CN + c] * a1;  // row sy + 1
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;                         // cv::VResizeLinear, 8-bit
      dst[i * CN + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

extern "C" int pp_resize_scale(int rows, int cols, int min_side, int max_side, double* scale) {
  // utils/image.py:281-304 compute_resize_scale
  if (rows <= 0 || cols <= 0 || min_side <= 0 || max_side <= 0 || !scale) return PP_ERR_ARG;
  const int smallest = rows < cols ? rows : cols, largest = rows < cols ? cols : rows;
  double s = (double)min_side / (double)smallest;
  if ((double)largest * s > (double)max_side) s = (double)max_side / (double)largest;
  *scale = s;
  return PP_OK;
}

extern "C" int pp_resize_linear_u8(pp_ctx* ctx, int n_img, int SH, int SW, int channels, double scale, int DH, int DW, const unsigned char* src,
                                   unsigned char* dst) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && dst && n_img > 0 && SH > 0 && SW > 0 && scale > 0 && (channels == 1 || channels == 3), PP_ERR_ARG, "pp_resize_linear_u8: bad image");
  // cv::resize with dsize = None: dsize = (saturate_cast<int>(cols * fx), saturate_cast<int>(rows * fy)), source step 1 / f
  const int want_w = (int)rint((double)SW * scale), want_h = (int)rint((double)SH * scale);
  PP_CHECK_ARG(ctx, DH == want_h && DW == want_w && DH > 0 && DW > 0, PP_ERR_SHAPE, "pp_resize_linear_u8: output must be %d x %d (rows x cols)", want_h, want_w);
  const size_t total = (size_t)n_img * DH * DW;
  const double inv = 1.0 / scale;
  if (channels == 3)
    hipLaunchKernelGGL((resize_linear_u8_kernel<3>), dim3(img_grid(total, ctx)), dim3(256), 0, ctx->stream, n_img, SH, SW, DH, DW, inv, inv, src, dst);
  else
    hipLaunchKernelGGL((resize_linear_u8_kernel<1>), dim3(img_grid(total, ctx)), dim3(256), 0, ctx->stream, n_img, SH, SW, DH, DW, inv, inv, src, dst);
  PP_CHECK_LAUNCH(ctx, "pp_resize_linear_u8");
  return PP_OK;
}
