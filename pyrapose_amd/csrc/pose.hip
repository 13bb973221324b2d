// Pose-error metrics of the evaluation tail: ADD and ADD-S/ADI (utils/pose_error.py:210-246, called at
// utils/linemod_eval.py:525-531 with the decision err < 0.1 * diameter).  float64 like the reference's numpy.
//   ADD = mean_i || (R_est p_i + t_est) - (R_gt p_i + t_gt) ||
//   ADI = mean_i  min_j || (R_gt p_i + t_gt) - (R_est p_j + t_est) ||     (cKDTree(pts_est).query(pts_gt, k=1))
// Reductions are fixed-order (per-tile partial sums, then one pass over the tiles): results do not depend on timing.
// Compiled with -ffp-contract=off: x*x + y*y + z*z is evaluated as written.
#include "pp_internal.h"

#define POSE_TILE 256

__device__ __forceinline__ void rigid(const double* __restrict__ R, const double* __restrict__ t, double x, double y, double z,
                                      double* ox, double* oy, double* oz) {
  *ox = R[0] * x + R[1] * y + R[2] * z + t[0];
  *oy = R[3] * x + R[4] * y + R[5] * z + t[1];
  *oz = R[6] * x + R[7] * y + R[8] * z + t[2];
}

__device__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = POSE_TILE / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// grid (tiles, poses): partial[pose][tile] = sum over the tile's points of the per-point distance
__global__ void pose_add_kernel(int n_pts, const double* __restrict__ pts, const double* __restrict__ R_est,
                                const double* __restrict__ t_est, const double* __restrict__ R_gt, const double* __restrict__ t_gt,
                                double* __restrict__ partial) {
  __shared__ double red[POSE_TILE];
  const int pose = blockIdx.y, i = blockIdx.x * POSE_TILE + threadIdx.x;
  double d = 0.0;
  if (i < n_pts) {
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    double ax, ay, az, bx, by, bz;
    rigid(R_est + 9 * pose, t_est + 3 * pose, x, y, z, &ax, &ay, &az);
    rigid(R_gt + 9 * pose, t_gt + 3 * pose, x, y, z, &bx, &by, &bz);
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    d = sqrt(dx * dx + dy * dy + dz * dz);
  }
  const double s = block_sum(d, red);
  if (threadIdx.x == 0) partial[(size_t)pose * gridDim.x + blockIdx.x] = s;
}

__global__ void pose_adi_kernel(int n_pts, const double* __restrict__ pts, const double* __restrict__ R_est,
                                const double* __restrict__ t_est, const double* __restrict__ R_gt, const double* __restrict__ t_gt,
                                double* __restrict__ partial) {
  __shared__ double red[POSE_TILE];
  __shared__ double ex[POSE_TILE], ey[POSE_TILE], ez[POSE_TILE];
  const int pose = blockIdx.y, i = blockIdx.x * POSE_TILE + threadIdx.x;
  double gx = 0.0, gy = 0.0, gz = 0.0;
  if (i < n_pts) rigid(R_gt + 9 * pose, t_gt + 3 * pose, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], &gx, &gy, &gz);
  double best = 1.0e300;
  for (int j0 = 0; j0 < n_pts; j0 += POSE_TILE) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    if (j < n_pts) rigid(R_est + 9 * pose, t_est + 3 * pose, pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], &ex[threadIdx.x], &ey[threadIdx.x], &ez[threadIdx.x]);
    __syncthreads();
    const int lim = min(POSE_TILE, n_pts - j0);
    for (int k = 0; k < lim; ++k) {
      const double dx = gx - ex[k], dy = gy - ey[k], dz = gz - ez[k];
      const double q = dx * dx + dy * dy + dz * dz;
      best = q < best ? q : best;
    }
  }
  const double s = block_sum(i < n_pts ? sqrt(best) : 0.0, red);
  if (threadIdx.x == 0) partial[(size_t)pose * gridDim.x + blockIdx.x] = s;
}

__global__ void pose_mean_kernel(int n_pose, int n_tiles, int n_pts, const double* __restrict__ partial, double* __restrict__ out) {
  const int pose = blockIdx.x * blockDim.x + threadIdx.x;
  if (pose >= n_pose) return;
  double s = 0.0;
  for (int t = 0; t < n_tiles; ++t) s += partial[(size_t)pose * n_tiles + t];
  out[pose] = s / (double)n_pts;
}

extern "C" size_t pp_pose_error_workspace_bytes(int n_pose, int n_pts) {
  if (n_pose <= 0 || n_pts <= 0) return 0;
  return (size_t)n_pose * ((n_pts + POSE_TILE - 1) / POSE_TILE) * sizeof(double);
}

static int pose_error(pp_ctx* ctx, bool symmetric, int n_pose, int n_pts, const double* pts, const double* R_est, const double* t_est,
                      const double* R_gt, const double* t_gt, void* workspace, double* out, const char* who) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_pose > 0 && n_pose <= 65535 && n_pts > 0, PP_ERR_SHAPE, "%s: need 1..65535 poses and at least one model point", who);
  PP_CHECK_ARG(ctx, pts && R_est && t_est && R_gt && t_gt && workspace && out, PP_ERR_ARG, "%s: null argument", who);
  const int tiles = (n_pts + POSE_TILE - 1) / POSE_TILE;
  double* partial = (double*)workspace;
  if (symmetric)
    hipLaunchKernelGGL(pose_adi_kernel, dim3(tiles, n_pose), dim3(POSE_TILE), 0, ctx->stream, n_pts, pts, R_est, t_est, R_gt, t_gt, partial);
  else
    hipLaunchKernelGGL(pose_add_kernel, dim3(tiles, n_pose), dim3(POSE_TILE), 0, ctx->stream, n_pts, pts, R_est, t_est, R_gt, t_gt, partial);
  hipLaunchKernelGGL(pose_mean_kernel, dim3((n_pose + 63) / 64), dim3(64), 0, ctx->stream, n_pose, tiles, n_pts, (const double*)partial, out);
  PP_CHECK_LAUNCH(ctx, who);
  return PP_OK;
}

extern "C" int pp_pose_add_f64(pp_ctx* ctx, int n_pose, int n_pts, const double* pts, const double* R_est, const double* t_est,
                               const double* R_gt, const double* t_gt, void* workspace, double* out) {
  return pose_error(ctx, false, n_pose, n_pts, pts, R_est, t_est, R_gt, t_gt, workspace, out, "pp_pose_add_f64");
}

extern "C" int pp_pose_adi_f64(pp_ctx* ctx, int n_pose, int n_pts, const double* pts, const double* R_est, const double* t_est,
                               const double* R_gt, const double* t_gt, void* workspace, double* out) {
  return pose_error(ctx, true, n_pose, n_pts, pts, R_est, t_est, R_gt, t_gt, workspace, out, "pp_pose_adi_f64");
}
