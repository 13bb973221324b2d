// RANSAC-PnP of the pose-decode tail: the k votes x 8 projected cuboid corners of one class -> (R, t) and
// the inlier set.  Stands where the reference calls cv2.solvePnPRansac(obj_points, est_points, K, None, iterationsCount=300,
// reprojectionError=5.0, confidence=0.99, flags=cv2.SOLVEPNP_ITERATIVE) + cv2.Rodrigues (utils/linemod_eval.py:479-485).
// OpenCV's internals (RNG, minimal solver, refinement) are third-party and absent: PARITY UNPINNED -- same contract
// (5 px inlier rule, rotation / translation / inlier list out), own algorithm, restated in oracle/pnp_np.py:
//   1. `iterations` hypotheses, one thread each: a minimal sample -- the 8 corners of ONE vote (every vote once, then random
//      votes; counter-based splitmix64 draws: results do not depend on scheduling), or six correspondences when the points
//      carry no vote structure -- normalised DLT, null vector of the 12x12 normal matrix by cyclic Jacobi, [R|t] by polar
//      decomposition of the left 3x3 block, 5 damped Gauss-Newton steps on the sample itself;
//   2. every hypothesis scored on all points (one wave per hypothesis): count of points with squared reprojection error
//      below reproj_error^2 in front of the camera; the largest count wins, ties go to the lower iteration;
//   3. damped Gauss-Newton on the inliers (rotation increments multiplied from the left), 10 iterations, inliers
//      re-selected, 10 more; fixed-order block reductions.
// One workgroup of 256 threads per problem (class x image); float64; compiled with -ffp-contract=off.
#include "pp_internal.h"

#define PNP_THREADS 256
#define PNP_REFINE_ITERS 10
#define PNP_POLISH_ITERS 5

struct PnpArgs {
  int n_problems;
  const int* offsets;    // [n_problems + 1] into the point arrays
  const double* obj;     // [N][3]
  const double* img;     // [N][2]
  const double* K4;      // [n_problems][4] = fx, fy, cx, cy
  int iterations;
  double thr2;
  unsigned long long seed;
  int ppv;               // points per vote (8 for cuboid corner votes), 0 = unstructured
  double* hyp;           // workspace [n_problems][iterations][12] : R row-major, t ; R[0] = NaN when the sample was degenerate
  double* R_out;         // [n_problems][9]
  double* t_out;         // [n_problems][3]
  int* n_inliers;        // [n_problems]
  unsigned char* mask;   // [N]
  int* ok;               // [n_problems]
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned long long pnp_draw(unsigned long long seed, int problem, int it, int j) {
  const unsigned long long key = ((seed & 0xFFFFFFFFull) << 32) ^ ((unsigned long long)(problem & 0xFFF) << 20) ^
                                 ((unsigned long long)(it & 0xFFFF) << 4) ^ (unsigned long long)(j & 0xF);
  return splitmix64(key) >> 11;
}

#define PNP_MAX_SAMPLE 16

// one minimal sample: with vote structure the ppv points of ONE vote (the it-th vote for it < number of votes, a random one
// afterwards); without, six correspondences at a random start and stride.  Returns the sample size, 0 = impossible.
__device__ int pnp_sample(unsigned long long seed, int problem, int it, int n, int ppv, int* idx) {
  if (ppv > 0) {
    const int nv = n / ppv;
    if (ppv < 6 || ppv > PNP_MAX_SAMPLE || nv < 1) return 0;
    const int v = it < nv ? it : (int)(pnp_draw(seed, problem, it, 0) % (unsigned long long)nv);
    for (int j = 0; j < ppv; ++j) idx[j] = v * ppv + j;
    return ppv;
  }
  if (n < 6) return 0;
  const int start = (int)(pnp_draw(seed, problem, it, 0) % (unsigned long long)n);
  const int span = (n - 1) / 6 > 1 ? (n - 1) / 6 : 1;
  const int step = 1 + (int)(pnp_draw(seed, problem, it, 1) % (unsigned long long)span);
  for (int j = 0; j < 6; ++j) idx[j] = (start + j * step) % n;
  return 6;
}

__device__ bool inv3(const double* M, double* inv, double* det_out) {
  const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
  const double A_ = e * i - f * h, B_ = c * h - b * i, C_ = b * f - c * e;
  const double det = a * A_ + d * B_ + g * C_;
  *det_out = det;
  if (!(fabs(det) > 1e-300)) return false;
  inv[0] = A_ / det; inv[1] = B_ / det; inv[2] = C_ / det;
  inv[3] = (f * g - d * i) / det; inv[4] = (a * i - c * g) / det; inv[5] = (c * d - a * f) / det;
  inv[6] = (d * h - e * g) / det; inv[7] = (b * g - a * h) / det; inv[8] = (a * e - b * d) / det;
  return true;
}

// m correspondences (6 <= m <= 16) -> pose; private 12x12 matrices live in scratch memory (300 threads per problem, not a
// hot path)
__device__ bool dlt_pose(int m, const double (*X)[3], const double (*xn)[2], double* R, double* t) {
  double c[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < 3; ++k) {
    double s = 0.0;
    for (int i = 0; i < m; ++i) s += X[i][k];
    c[k] = s / (double)m;
  }
  double Xn[PNP_MAX_SAMPLE][3];
  double ss = 0.0;
  for (int i = 0; i < m; ++i)
    for (int k = 0; k < 3; ++k) {
      Xn[i][k] = X[i][k] - c[k];
      ss += Xn[i][k] * Xn[i][k];
    }
  const double rms = sqrt(ss / (double)m);
  if (!(rms > 0.0)) return false;
  const double s = 1.0 / rms;
  double A[12][12], V[12][12];
  for (int a = 0; a < 12; ++a)
    for (int b = 0; b < 12; ++b) {
      A[a][b] = 0.0;
      V[a][b] = a == b ? 1.0 : 0.0;
    }
  for (int i = 0; i < m; ++i) {
    const double px = Xn[i][0] * s, py = Xn[i][1] * s, pz = Xn[i][2] * s, x = xn[i][0], y = xn[i][1];
    const double r1[12] = {px, py, pz, 1.0, 0.0, 0.0, 0.0, 0.0, -x * px, -x * py, -x * pz, -x};
    const double r2[12] = {0.0, 0.0, 0.0, 0.0, px, py, pz, 1.0, -y * px, -y * py, -y * pz, -y};
    for (int a = 0; a < 12; ++a)
      for (int b = 0; b < 12; ++b) A[a][b] = A[a][b] + (r1[a] * r1[b] + r2[a] * r2[b]);
  }
  for (int sweep = 0; sweep < 8; ++sweep)
    for (int p = 0; p < 11; ++p)
      for (int q = p + 1; q < 12; ++q) {
        const double apq = A[p][q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cc = 1.0 / sqrt(tt * tt + 1.0);
        const double sn = tt * cc;
        for (int k = 0; k < 12; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = cc * akp - sn * akq;
          A[k][q] = sn * akp + cc * akq;
        }
        for (int k = 0; k < 12; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cc * apk - sn * aqk;
          A[q][k] = sn * apk + cc * aqk;
        }
        for (int k = 0; k < 12; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cc * vkp - sn * vkq;
          V[k][q] = sn * vkp + cc * vkq;
        }
      }
  int kmin = 0;
  for (int k = 1; k < 12; ++k)
    if (A[k][k] < A[kmin][kmin]) kmin = k;
  double M[9], p4[3];
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) M[3 * r + k] = V[4 * r + k][kmin] * s;
    p4[r] = V[4 * r + 3][kmin] - (M[3 * r] * c[0] + M[3 * r + 1] * c[1] + M[3 * r + 2] * c[2]);
  }
  double inv[9], det;
  if (!inv3(M, inv, &det)) return false;
  if (det < 0.0) {
    for (int k = 0; k < 9; ++k) M[k] = -M[k];
    for (int k = 0; k < 3; ++k) p4[k] = -p4[k];
    det = -det;
  }
  double lam = cbrt(det);
  for (int k = 0; k < 9; ++k) R[k] = M[k] / lam;
  for (int itp = 0; itp < 12; ++itp) {
    double dR;
    if (!inv3(R, inv, &dR)) return false;
    // R <- (R + R^-T) / 2
    const double n0 = 0.5 * (R[0] + inv[0]), n1 = 0.5 * (R[1] + inv[3]), n2 = 0.5 * (R[2] + inv[6]);
    const double n3 = 0.5 * (R[3] + inv[1]), n4 = 0.5 * (R[4] + inv[4]), n5 = 0.5 * (R[5] + inv[7]);
    const double n6 = 0.5 * (R[6] + inv[2]), n7 = 0.5 * (R[7] + inv[5]), n8 = 0.5 * (R[8] + inv[8]);
    R[0] = n0; R[1] = n1; R[2] = n2; R[3] = n3; R[4] = n4; R[5] = n5; R[6] = n6; R[7] = n7; R[8] = n8;
  }
  lam = 0.0;
  for (int k = 0; k < 9; ++k) lam += R[k] * M[k];
  lam /= 3.0;
  if (!(lam > 0.0)) return false;
  for (int k = 0; k < 3; ++k) t[k] = p4[k] / lam;
  return true;
}

// squared reprojection error of point i under (R, t); false when the point is behind the camera
__device__ __forceinline__ bool reproj_sq(const double* R, const double* t, const double* __restrict__ X, const double* __restrict__ uv,
                                          double fx, double fy, double cx, double cy, double* e) {
  const double x = (X[0] * R[0] + X[1] * R[1] + X[2] * R[2]) + t[0];
  const double y = (X[0] * R[3] + X[1] * R[4] + X[2] * R[5]) + t[1];
  const double z = (X[0] * R[6] + X[1] * R[7] + X[2] * R[8]) + t[2];
  if (!(z > 1e-9)) {
    *e = 1e12;
    return false;
  }
  const double du = fx * x / z + cx - uv[0], dv = fy * y / z + cy - uv[1];
  *e = du * du + dv * dv;
  return true;
}

// fixed-order sum of NV doubles per thread over the block; result in red[0 .. NV)
template <int NV>
__device__ void block_sum_vec(const double* v, double* red /* [PNP_THREADS / 64][NV] */, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if (lane == 0) red[wave * NV + k] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = red[threadIdx.x];
    for (int w = 1; w < PNP_THREADS / 64; ++w) s += red[w * NV + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

__device__ void so3_exp_mul(const double* w, const double* R, double* out) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double Kx[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  double E[9];
  double a, b;
  if (th < 1e-12) {
    a = 1.0;
    b = 0.0;
  } else {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / (th * th);
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double k2 = 0.0;
      for (int k = 0; k < 3; ++k) k2 += Kx[3 * r + k] * Kx[3 * k + c];
      E[3 * r + c] = (r == c ? 1.0 : 0.0) + a * Kx[3 * r + c] + b * k2;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += E[3 * r + k] * R[3 * k + c];
      out[3 * r + c] = s;
    }
}

__device__ bool solve6(const double (*H)[6], const double* g, double* d) {
  double L[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = 0; k < j; ++k) s += L[i][k] * L[j][k];
      s = H[i][j] - s;
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < i; ++k) s += L[i][k] * y[k];
    y[i] = (g[i] - s) / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = 0.0;
    for (int k = i + 1; k < 6; ++k) s += L[k][i] * d[k];
    d[i] = (y[i] - s) / L[i][i];
  }
  return true;
}

// normal-equation terms of one point: v[0..20] += lower triangle of J^T J, v[21..26] -= J^T r
__device__ __forceinline__ void accumulate_point(const double* R, const double* t, const double* X, const double* uv, double fx, double fy,
                                                 double cx, double cy, double* v) {
  const double ra = X[0] * R[0] + X[1] * R[1] + X[2] * R[2];
  const double rb = X[0] * R[3] + X[1] * R[4] + X[2] * R[5];
  const double rc = X[0] * R[6] + X[1] * R[7] + X[2] * R[8];
  const double x = ra + t[0], y = rb + t[1], z = rc + t[2];
  if (!(z > 1e-9)) return;
  const double ru = fx * x / z + cx - uv[0], rv = fy * y / z + cy - uv[1];
  const double ju0 = fx / z, ju2 = -fx * x / (z * z), jv1 = fy / z, jv2 = -fy * y / (z * z);
  // d Xc / d w = -[R X]x ; d Xc / d t = I
  const double Ju[6] = {ju2 * rb, ju0 * rc - ju2 * ra, -ju0 * rb, ju0, 0.0, ju2};
  const double Jv[6] = {jv2 * rb - jv1 * rc, -jv2 * ra, jv1 * ra, 0.0, jv1, jv2};
  int q = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c <= r; ++c, ++q) v[q] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
  for (int r = 0; r < 6; ++r) v[21 + r] -= Ju[r] * ru + Jv[r] * rv;
}

// sums -> damped 6x6 system -> candidate pose; false when the system is not positive definite
__device__ bool lm_candidate(const double* sums, double lam, const double* R, const double* t, double* R2, double* t2) {
  double H[6][6], g[6], d[6];
  int q = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c <= r; ++c, ++q) H[r][c] = H[c][r] = sums[q];
  for (int r = 0; r < 6; ++r) g[r] = sums[21 + r];
  for (int r = 0; r < 6; ++r) H[r][r] = H[r][r] + lam * H[r][r] + 1e-12;
  if (!solve6(H, g, d)) return false;
  so3_exp_mul(d, R, R2);
  for (int k = 0; k < 3; ++k) t2[k] = t[k] + d[3 + k];
  return true;
}

// the hypothesis polished on its own sample: the damped Gauss-Newton of step 3, sequential over m points
__device__ void polish_on_sample(int m, const double (*X)[3], const double (*uv)[2], double fx, double fy, double cx, double cy, int iters,
                                 double* R, double* t) {
  double lam = 1e-3, cur = 0.0;
  for (int i = 0; i < m; ++i) {
    double e;
    reproj_sq(R, t, X[i], uv[i], fx, fy, cx, cy, &e);
    cur += e;
  }
  for (int iter = 0; iter < iters; ++iter) {
    double v[27];
    for (int k = 0; k < 27; ++k) v[k] = 0.0;
    for (int i = 0; i < m; ++i) accumulate_point(R, t, X[i], uv[i], fx, fy, cx, cy, v);
    double R2[9], t2[3];
    if (!lm_candidate(v, lam, R, t, R2, t2)) break;
    double c2 = 0.0;
    for (int i = 0; i < m; ++i) {
      double e;
      reproj_sq(R2, t2, X[i], uv[i], fx, fy, cx, cy, &e);
      c2 += e;
    }
    if (c2 < cur) {
      for (int k = 0; k < 9; ++k) R[k] = R2[k];
      for (int k = 0; k < 3; ++k) t[k] = t2[k];
      cur = c2;
      lam = lam * 0.1 > 1e-9 ? lam * 0.1 : 1e-9;
    } else {
      lam = lam * 10.0 < 1e6 ? lam * 10.0 : 1e6;
    }
  }
}

__global__ __launch_bounds__(PNP_THREADS) void pnp_ransac_kernel(const PnpArgs a) {
  const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = a.offsets[prob], n = a.offsets[prob + 1] - p0;
  const double* obj = a.obj + 3 * (size_t)p0;
  const double* img = a.img + 2 * (size_t)p0;
  unsigned char* mask = a.mask + p0;
  const double fx = a.K4[4 * prob], fy = a.K4[4 * prob + 1], cx = a.K4[4 * prob + 2], cy = a.K4[4 * prob + 3];
  double* hyp = a.hyp + (size_t)prob * a.iterations * 12;

  __shared__ double red[(PNP_THREADS / 64) * 28];
  __shared__ double sums[28];
  __shared__ double sR[9], st[3], cR[9], ct[3];
  __shared__ int s_best_cnt[PNP_THREADS / 64], s_best_it[PNP_THREADS / 64];
  __shared__ int s_flag;

  // ---- 1. hypotheses ----
  for (int it = tid; it < a.iterations; it += PNP_THREADS) {
    int idx[PNP_MAX_SAMPLE];
    double R[9], t[3];
    const int m = pnp_sample(a.seed, prob, it, n, a.ppv, idx);
    bool good = m > 0;
    if (good) {
      double X[PNP_MAX_SAMPLE][3], xn[PNP_MAX_SAMPLE][2], uv[PNP_MAX_SAMPLE][2];
      for (int j = 0; j < m; ++j) {
        for (int k = 0; k < 3; ++k) X[j][k] = obj[3 * idx[j] + k];
        uv[j][0] = img[2 * idx[j]];
        uv[j][1] = img[2 * idx[j] + 1];
        xn[j][0] = (uv[j][0] - cx) / fx;
        xn[j][1] = (uv[j][1] - cy) / fy;
      }
      good = dlt_pose(m, X, xn, R, t);
      // the DLT ignores that [R|t] has 6 degrees of freedom: polish on the sample itself before scoring
      if (good) polish_on_sample(m, X, uv, fx, fy, cx, cy, PNP_POLISH_ITERS, R, t);
    }
    double* h = hyp + (size_t)it * 12;
    if (good) {
      for (int k = 0; k < 9; ++k) h[k] = R[k];
      for (int k = 0; k < 3; ++k) h[9 + k] = t[k];
    } else {
      h[0] = __longlong_as_double(0x7ff8000000000000ll);
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- 2. score: wave w takes hypotheses w, w + 4, ... ----
  int best_cnt = -1, best_it = -1;
  for (int it = wave; it < a.iterations; it += PNP_THREADS / 64) {
    const double* h = hyp + (size_t)it * 12;
    if (h[0] != h[0]) continue;  // wave-uniform
    double R[9], t[3];
    for (int k = 0; k < 9; ++k) R[k] = h[k];
    for (int k = 0; k < 3; ++k) t[k] = h[9 + k];
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
      double e;
      const bool front = reproj_sq(R, t, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, &e);
      cnt += (front && e < a.thr2) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    cnt = __shfl(cnt, 0, 64);
    if (cnt > best_cnt) {
      best_cnt = cnt;
      best_it = it;
    }
  }
  if (lane == 0) {
    s_best_cnt[wave] = best_cnt;
    s_best_it[wave] = best_it;
  }
  __syncthreads();
  if (tid == 0) {
    int bc = -1, bi = -1;
    for (int w = 0; w < PNP_THREADS / 64; ++w)
      if (s_best_cnt[w] > bc || (s_best_cnt[w] == bc && bc >= 0 && s_best_it[w] < bi)) {
        bc = s_best_cnt[w];
        bi = s_best_it[w];
      }
    s_flag = bc >= 4 ? 1 : 0;
    if (bc >= 4) {
      const double* h = hyp + (size_t)bi * 12;
      for (int k = 0; k < 9; ++k) sR[k] = h[k];
      for (int k = 0; k < 3; ++k) st[k] = h[9 + k];
    }
  }
  __syncthreads();
  if (!s_flag) {  // block-uniform
    for (int i = tid; i < n; i += PNP_THREADS) mask[i] = 0;
    if (tid == 0) {
      for (int k = 0; k < 9; ++k) a.R_out[9 * prob + k] = (k % 4 == 0) ? 1.0 : 0.0;
      for (int k = 0; k < 3; ++k) a.t_out[3 * prob + k] = 0.0;
      a.n_inliers[prob] = 0;
      a.ok[prob] = 0;
    }
    return;
  }

  // ---- 3. refine on the inliers, twice ----
  for (int round = 0; round < 2; ++round) {
    for (int i = tid; i < n; i += PNP_THREADS) {
      double e;
      const bool front = reproj_sq(sR, st, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, &e);
      mask[i] = (front && e < a.thr2) ? 1 : 0;
    }
    __threadfence_block();
    __syncthreads();
    double lam = 1e-3;  // every thread keeps the same copy
    double v[28];
    // cost of the starting pose
    {
      double c0 = 0.0;
      for (int i = tid; i < n; i += PNP_THREADS)
        if (mask[i]) {
          double e;
          reproj_sq(sR, st, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, &e);
          c0 += e;
        }
      v[0] = c0;
      block_sum_vec<1>(v, red, sums);
    }
    double cur = sums[0];
    __syncthreads();
    for (int iter = 0; iter < PNP_REFINE_ITERS; ++iter) {
      for (int k = 0; k < 28; ++k) v[k] = 0.0;
      for (int i = tid; i < n; i += PNP_THREADS)
        if (mask[i]) accumulate_point(sR, st, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, v);
      block_sum_vec<27>(v, red, sums);
      if (tid == 0) s_flag = lm_candidate(sums, lam, sR, st, cR, ct) ? 1 : 0;
      __syncthreads();
      if (!s_flag) break;  // block-uniform
      double c2 = 0.0;
      for (int i = tid; i < n; i += PNP_THREADS)
        if (mask[i]) {
          double e;
          reproj_sq(cR, ct, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, &e);
          c2 += e;
        }
      v[0] = c2;
      block_sum_vec<1>(v, red, sums);
      const double cand = sums[0];
      __syncthreads();
      if (cand < cur) {  // block-uniform
        if (tid == 0) {
          for (int k = 0; k < 9; ++k) sR[k] = cR[k];
          for (int k = 0; k < 3; ++k) st[k] = ct[k];
        }
        cur = cand;
        lam = lam * 0.1 > 1e-9 ? lam * 0.1 : 1e-9;
      } else {
        lam = lam * 10.0 < 1e6 ? lam * 10.0 : 1e6;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  // final inlier set
  int cnt = 0;
  for (int i = tid; i < n; i += PNP_THREADS) {
    double e;
    const bool front = reproj_sq(sR, st, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, &e);
    const int in = (front && e < a.thr2) ? 1 : 0;
    mask[i] = (unsigned char)in;
    cnt += in;
  }
  double vc[1] = {(double)cnt};
  block_sum_vec<1>(vc, red, sums);
  if (tid == 0) {
    const int total = (int)sums[0];
    for (int k = 0; k < 9; ++k) a.R_out[9 * prob + k] = sR[k];
    for (int k = 0; k < 3; ++k) a.t_out[3 * prob + k] = st[k];
    a.n_inliers[prob] = total;
    a.ok[prob] = total >= 4 ? 1 : 0;
  }
}

extern "C" size_t pp_pnp_ransac_workspace_bytes(int n_problems, int iterations) {
  if (n_problems <= 0 || iterations <= 0) return 0;
  return (size_t)n_problems * (size_t)iterations * 12 * sizeof(double);
}

extern "C" int pp_pnp_ransac_f64(pp_ctx* ctx, int n_problems, const int* offsets_dev, int n_points_total, const double* obj,
                                 const double* img, const double* K4, int iterations, double reproj_error, unsigned long long seed,
                                 int points_per_vote, void* workspace, double* R_out, double* t_out, int* n_inliers,
                                 unsigned char* inlier_mask, int* ok) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_problems >= 0 && n_points_total >= 0 && iterations > 0 && iterations <= 65536 && reproj_error > 0.0, PP_ERR_ARG,
               "pp_pnp_ransac_f64: bad counts (iterations 1..65536, reproj_error > 0)");
  PP_CHECK_ARG(ctx, n_problems <= 4096, PP_ERR_ARG, "pp_pnp_ransac_f64: at most 4096 problems per call");
  PP_CHECK_ARG(ctx, points_per_vote == 0 || (points_per_vote >= 6 && points_per_vote <= 16), PP_ERR_ARG,
               "pp_pnp_ransac_f64: points_per_vote must be 0 or 6..16");
  if (n_problems == 0) return PP_OK;
  PP_CHECK_ARG(ctx, offsets_dev && obj && img && K4 && workspace && R_out && t_out && n_inliers && inlier_mask && ok, PP_ERR_ARG,
               "pp_pnp_ransac_f64: null pointer");
  PnpArgs a;
  a.n_problems = n_problems; a.offsets = offsets_dev; a.obj = obj; a.img = img; a.K4 = K4;
  a.iterations = iterations; a.thr2 = reproj_error * reproj_error; a.seed = seed; a.ppv = points_per_vote;
  a.hyp = (double*)workspace; a.R_out = R_out; a.t_out = t_out; a.n_inliers = n_inliers; a.mask = inlier_mask; a.ok = ok;
  hipLaunchKernelGGL(pnp_ransac_kernel, dim3((unsigned)n_problems), dim3(PNP_THREADS), 0, ctx->stream, a);
  PP_CHECK_LAUNCH(ctx, "pp_pnp_ransac_f64");
  return PP_OK;
}
