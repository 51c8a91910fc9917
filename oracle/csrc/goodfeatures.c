/*
 * CPU oracle: Shi-Tomasi corner selection ("good features to track").
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product.
 *
 * Reference call site: src/vo/features/klt.py:98
 *     cv2.goodFeaturesToTrack(img, mask=mask, maxCorners=500, qualityLevel=0.01,
 *                             minDistance=8, blockSize=7)                 (klt.py:24-26)
 * PARITY UNPINNED against OpenCV (opencv-python==4.8.1.78 is absent; the reference has no
 * test for it).  Restated algorithm (cornerMinEigenVal + goodFeaturesToTrack):
 *   Dx, Dy   3x3 Sobel (correlation), reflect-101 border, scaled by 1 / (4 * blockSize * 255)
 *   cov      blockSize x blockSize box sums of Dx^2, DxDy, Dy^2 (anchor at the centre,
 *            reflect-101 border)
 *   eig      (a + c) - sqrt((a - c)^2 + b^2),  a = cov_xx / 2, b = cov_xy, c = cov_yy / 2
 *   keep     eig > quality * max(eig) [within the mask], equal to its 3x3 maximum, non-zero,
 *            rows 1..H-2 and columns 1..W-2, mask != 0
 *   order    eig descending, ties by higher pixel address first
 *   greedy   accept a corner unless an accepted one lies closer than minDistance (Euclidean);
 *            stop at maxCorners
 * One deliberate difference, as in klt.c: the box sums are exact integers (the Sobel outputs
 * are integers before scaling) converted to float32 once, instead of float32 accumulation
 * whose rounding depends on OpenCV's SIMD build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline int refl(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = c < 0 ? -c : 2 * (n - 1) - c;
  return c;
}

void oracle_min_eigen_map(const uint8_t* img, int H, int W, int block, float* eig) {
  int* gx = (int*)malloc(sizeof(int) * (size_t)H * W);
  int* gy = (int*)malloc(sizeof(int) * (size_t)H * W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int p[3][3];
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) p[j][i] = img[(size_t)refl(y + j - 1, H) * W + refl(x + i - 1, W)];
      gx[(size_t)y * W + x] = (p[0][2] - p[0][0]) + 2 * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0]);
      gy[(size_t)y * W + x] = (p[2][0] - p[0][0]) + 2 * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2]);
    }
  const double scale = 1.0 / (4.0 * block * 255.0);
  const float s2 = (float)(scale * scale);
  const int r0 = block / 2;   /* anchor = centre: window [-(block/2), block - 1 - block/2] */
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      long long sxx = 0, sxy = 0, syy = 0;
      for (int j = 0; j < block; ++j) {
        int yy = refl(y + j - r0, H);
        for (int i = 0; i < block; ++i) {
          int xx = refl(x + i - r0, W);
          long long a = gx[(size_t)yy * W + xx], b = gy[(size_t)yy * W + xx];
          sxx += a * a;
          sxy += a * b;
          syy += b * b;
        }
      }
      float a = (float)sxx * s2 * 0.5f, b = (float)sxy * s2, c = (float)syy * s2 * 0.5f;
      eig[(size_t)y * W + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    }
  free(gx);
  free(gy);
}

typedef struct {
  float v;
  int idx;
} cand_t;

static int cmp_cand(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa;
  const cand_t* b = (const cand_t*)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}

/* returns the number of corners; xy: max_corners * 2 float32 */
int oracle_good_features(const uint8_t* img, int H, int W, const uint8_t* mask, int max_corners, double quality,
                         double min_dist, int block, float* xy) {
  float* eig = (float*)malloc(sizeof(float) * (size_t)H * W);
  oracle_min_eigen_map(img, H, W, block, eig);
  float mx = 0.f;
  int have = 0;
  for (size_t i = 0; i < (size_t)H * W; ++i)
    if (!mask || mask[i]) {
      if (!have || eig[i] > mx) mx = eig[i];
      have = 1;
    }
  const float thr = (float)((double)mx * quality);
  cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)H * W);
  int nc = 0;
  for (int y = 1; y < H - 1; ++y)
    for (int x = 1; x < W - 1; ++x) {
      float v = eig[(size_t)y * W + x];
      if (!(v > thr)) continue;                      /* THRESH_TOZERO, then val != 0 */
      if (v == 0.f) continue;
      if (mask && !mask[(size_t)y * W + x]) continue;
      float m = 0.f;
      for (int j = -1; j <= 1; ++j)
        for (int i = -1; i <= 1; ++i) {
          float q = eig[(size_t)(y + j) * W + (x + i)];
          q = q > thr ? q : 0.f;
          if (q > m) m = q;
        }
      if (v == m) {
        c[nc].v = v;
        c[nc].idx = y * W + x;
        ++nc;
      }
    }
  qsort(c, nc, sizeof(cand_t), cmp_cand);
  int n = 0;
  const double md2 = min_dist * min_dist;
  for (int k = 0; k < nc && (max_corners <= 0 || n < max_corners); ++k) {
    int y = c[k].idx / W, x = c[k].idx % W;
    int ok = 1;
    if (min_dist >= 1)
      for (int j = 0; j < n; ++j) {
        double dx = x - xy[2 * j], dy = y - xy[2 * j + 1];
        if (dx * dx + dy * dy < md2) {
          ok = 0;
          break;
        }
      }
    if (ok) {
      xy[2 * n] = (float)x;
      xy[2 * n + 1] = (float)y;
      ++n;
    }
  }
  free(c);
  free(eig);
  return n;
}
