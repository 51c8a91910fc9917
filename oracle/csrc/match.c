/*
 * CPU oracle: brute-force 2-nearest-neighbour descriptor matching with ratio test and the
 * reference's first-come uniqueness filter.
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product.
 *
 * Reference call sites: src/vo/features/harris.py:246-262 (ratio 0.85, raw 19x19 patches,
 * D = 361) and src/vo/features/sift.py:38-54 (ratio 0.8, D = 128):
 *     matches = cv2.BFMatcher().knnMatch(desc1, desc2, k=2)        # NORM_L2, no cross-check
 *     for m, n in matches:
 *         if m.distance < ratio * n.distance and used[m.trainIdx] == 0: keep (queryIdx, trainIdx)
 * PARITY UNPINNED against OpenCV (opencv-python==4.8.1.78 is absent; the reference's tests
 * pin only shapes, tests/test_harris.py:30-122).  Definition restated here: distance =
 * sqrtf((float) sum_k (a_k - b_k)^2) with the sum accumulated in float64 in index order
 * (exact for the integer-valued descriptors both call sites produce); nearest and second
 * nearest by (distance^2, train index); the ratio test is evaluated in float64 as Python
 * does with the two float32 distances; queries are visited in order.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* best[2*i], best[2*i+1] = indices of the two nearest train rows of query i (-1 if absent);
 * d2[2*i], d2[2*i+1] = their squared distances as float64 */
void oracle_knn2(const float* q, int nq, const float* t, int nt, int D, int32_t* best, double* d2) {
  for (int i = 0; i < nq; ++i) {
    int b0 = -1, b1 = -1;
    double e0 = 0, e1 = 0;
    for (int j = 0; j < nt; ++j) {
      double s = 0.0;
      for (int k = 0; k < D; ++k) {
        double d = (double)q[(size_t)i * D + k] - (double)t[(size_t)j * D + k];
        s += d * d;
      }
      if (b0 < 0 || s < e0) {
        b1 = b0;
        e1 = e0;
        b0 = j;
        e0 = s;
      } else if (b1 < 0 || s < e1) {
        b1 = j;
        e1 = s;
      }
    }
    best[2 * i] = b0;
    best[2 * i + 1] = b1;
    d2[2 * i] = e0;
    d2[2 * i + 1] = e1;
  }
}

/* returns the number of pairs written to pairs[2*k] = query, pairs[2*k+1] = train */
int oracle_match_knn2_ratio(const float* q, int nq, const float* t, int nt, int D, double ratio, int32_t* pairs,
                            int32_t* work_best, double* work_d2, uint8_t* used /* nt, zeroed by caller */) {
  oracle_knn2(q, nq, t, nt, D, work_best, work_d2);
  int n = 0;
  for (int i = 0; i < nq; ++i) {
    if (work_best[2 * i] < 0 || work_best[2 * i + 1] < 0) continue;   /* knnMatch returned fewer than 2 */
    float m = sqrtf((float)work_d2[2 * i]), s = sqrtf((float)work_d2[2 * i + 1]);
    if ((double)m < ratio * (double)s && !used[work_best[2 * i]]) {
      pairs[2 * n] = i;
      pairs[2 * n + 1] = work_best[2 * i];
      used[work_best[2 * i]] = 1;
      ++n;
    }
  }
  return n;
}
