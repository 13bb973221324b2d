/* Plain-C restatement of the reference's compiled op on the hot path and its callers' index logic.
 * TEST INFRASTRUCTURE (see oracle/__init__.py); pinned by tests/golden/anchors_targets.npz.
 *   oracle_compute_overlap      <- PyraPose/utils/compute_overlap.pyx:13-53
 *   oracle_gt_annotations       <- PyraPose/utils/anchors.py:290-318
 *   oracle_shift                <- PyraPose/utils/anchors.py:415-444
 */
#include <math.h>
#include <stddef.h>

void oracle_compute_overlap(const double* boxes, int n, const double* query, int k, double* out) {
  for (int q = 0; q < k; ++q) {
    const double* qb = query + 4 * q;
    double box_area = (qb[2] - qb[0] + 1) * (qb[3] - qb[1] + 1);
    for (int i = 0; i < n; ++i) {
      const double* b = boxes + 4 * i;
      double v = 0.0;
      double iw = fmin(b[2], qb[2]) - fmax(b[0], qb[0]) + 1;
      if (iw > 0) {
        double ih = fmin(b[3], qb[3]) - fmax(b[1], qb[1]) + 1;
        if (ih > 0) {
          double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
          v = iw * ih / ua;
        }
      }
      out[(size_t)i * k + q] = v;
    }
  }
}

void oracle_gt_annotations(const double* overlaps, int n, int k, double neg, double pos, int* argmax, signed char* state) {
  for (int i = 0; i < n; ++i) {
    int am = 0;
    double mx = overlaps[(size_t)i * k];
    for (int q = 1; q < k; ++q)
      if (overlaps[(size_t)i * k + q] > mx) { mx = overlaps[(size_t)i * k + q]; am = q; }
    argmax[i] = am;
    state[i] = (mx >= pos) ? 1 : ((mx > neg) ? -1 : 0);
  }
}

void oracle_shift(int fh, int fw, int stride, const double* base, int A, double* out) {
  for (int y = 0; y < fh; ++y)
    for (int x = 0; x < fw; ++x)
      for (int a = 0; a < A; ++a) {
        double sx = (x + 0.5) * stride, sy = (y + 0.5) * stride;
        double* o = out + ((size_t)(y * fw + x) * A + a) * 4;
        o[0] = base[4 * a + 0] + sx; o[1] = base[4 * a + 1] + sy;
        o[2] = base[4 * a + 2] + sx; o[3] = base[4 * a + 3] + sy;
      }
}
