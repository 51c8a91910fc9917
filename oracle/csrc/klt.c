/*
 * CPU oracle: image pyramid + pyramidal Lucas-Kanade tracking.
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product.
 *
 * Reference call site: src/vo/features/klt.py:233-249
 *     cv2.calcOpticalFlowPyrLK(prev, next, prevPts, None, winSize=(17,17), maxLevel=2,
 *                              criteria=(EPS|COUNT, 10, 0.03))        (klt.py:29-33)
 *     keep = status & (err < 100)                                      (klt.py:244-249)
 *
 * PARITY UNPINNED against OpenCV: the arithmetic lives in opencv-python==4.8.1.78
 * (environment.yml:22), absent here, and the reference has no KLT test.  This is
 * a restatement of the published algorithm as OpenCV's lkpyramid implements it
 * (Bouguet's pyramidal LK): 5-tap [1 4 6 4 1]/16 pyrDown with reflect-101
 * borders and (sum+128)>>8 rounding; Scharr derivatives (3,10,3) as int16 with a
 * reflect-101 border inside the image and zero outside; 14-bit fixed-point
 * bilinear weights; template patch at 5 extra bits; 2x2 normal matrix in float32
 * scaled by 2^-20; min-eigenvalue test / (win*win); <= max_iter Newton steps with
 * |delta|^2 <= eps^2 and the ping-pong stop; err = mean |patch difference| / 32.
 * One deliberate difference: the window sums (A11, A12, A22, b1, b2) are
 * accumulated as exact integers and converted to float32 once, instead of
 * float32 accumulation whose rounding depends on OpenCV's SIMD build; this makes
 * the result independent of summation order (what the GPU needs for bit-parity).
 * Validated against the analytic flow of synthetic scenes (tests/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) {
    if (c < 0) c = -c;
    else c = 2 * (n - 1) - c;
  }
  return c;
}

/* dst: ((H+1)/2) x ((W+1)/2) */
void oracle_pyr_down(const uint8_t* src, int H, int W, uint8_t* dst) {
  int Hd = (H + 1) / 2, Wd = (W + 1) / 2;
  static const int w[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < Hd; ++y)
    for (int x = 0; x < Wd; ++x) {
      int sum = 0;
      for (int j = 0; j < 5; ++j) {
        int sy = reflect101(2 * y + j - 2, H);
        int row = 0;
        for (int i = 0; i < 5; ++i) row += w[i] * src[(size_t)sy * W + reflect101(2 * x + i - 2, W)];
        sum += w[j] * row;
      }
      dst[(size_t)y * Wd + x] = (uint8_t)((sum + 128) >> 8);
    }
}

typedef struct {
  const uint8_t* img;
  int H, W;
} level_t;

static inline int pix(const level_t* L, int y, int x) {
  return L->img[(size_t)reflect101(y, L->H) * L->W + reflect101(x, L->W)];
}

/* Scharr derivative at (y, x): zero outside the image, reflect-101 inside */
static inline void scharr(const level_t* L, int y, int x, int* dx, int* dy) {
  if (y < 0 || y >= L->H || x < 0 || x >= L->W) {
    *dx = 0;
    *dy = 0;
    return;
  }
  int t[3], d[3];
  for (int k = 0; k < 3; ++k) {
    int xx = x + k - 1;
    int a = pix(L, y - 1, xx), b = pix(L, y, xx), c = pix(L, y + 1, xx);
    t[k] = (a + c) * 3 + b * 10;
    d[k] = c - a;
  }
  *dx = t[2] - t[0];
  *dy = (d[2] + d[0]) * 3 + d[1] * 10;
}

#define W_BITS 14
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

static void weights(float a, float b, int* w00, int* w01, int* w10, int* w11) {
  *w00 = (int)rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
  *w01 = (int)rintf(a * (1.f - b) * (float)(1 << W_BITS));
  *w10 = (int)rintf((1.f - a) * b * (float)(1 << W_BITS));
  *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

/*
 * prev_pyr/next_pyr: arrays of level pointers (level 0 = full image), sizes Hs/Ws.
 * n_levels = max_level + 1 actually available.
 */
void oracle_klt_track_pyr(const uint8_t* const* prev_pyr, const uint8_t* const* next_pyr, const int* Hs,
                          const int* Ws, int n_levels, const float* prev_xy, int N, int win, int max_iter,
                          double eps, double min_eig_thr, float* next_xy, uint8_t* status, float* err) {
  const int ww = win * win;
  short* Ibuf = (short*)malloc(sizeof(short) * ww * 3);
  if (max_iter < 0) max_iter = 0;
  if (max_iter > 100) max_iter = 100;
  if (eps < 0) eps = 0;
  if (eps > 10) eps = 10;
  const double eps2 = eps * eps;
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);

  for (int i = 0; i < N; ++i) {
    status[i] = 1;
    err[i] = 0.f;
    float nx = 0.f, ny = 0.f; /* nextPts[i] carried across levels */
    for (int level = n_levels - 1; level >= 0; --level) {
      level_t I = {prev_pyr[level], Hs[level], Ws[level]};
      level_t J = {next_pyr[level], Hs[level], Ws[level]};
      const float sc = (float)(1. / (double)(1 << level));
      float px = prev_xy[2 * i] * sc, py = prev_xy[2 * i + 1] * sc;
      float qx, qy;
      if (level == n_levels - 1) {
        qx = px;
        qy = py;
      } else {
        qx = nx * 2.f;
        qy = ny * 2.f;
      }
      nx = qx;
      ny = qy;
      px -= half;
      py -= half;
      int ipx = (int)floorf(px), ipy = (int)floorf(py);
      if (ipx < -win || ipx >= I.W || ipy < -win || ipy >= I.H) {
        if (level == 0) {
          status[i] = 0;
          err[i] = 0.f;
        }
        continue;
      }
      float a = px - (float)ipx, b = py - (float)ipy;
      int w00, w01, w10, w11;
      weights(a, b, &w00, &w01, &w10, &w11);
      long long sA11 = 0, sA12 = 0, sA22 = 0;
      for (int y = 0; y < win; ++y)
        for (int x = 0; x < win; ++x) {
          int X = ipx + x, Y = ipy + y;
          int ival = descale(pix(&I, Y, X) * w00 + pix(&I, Y, X + 1) * w01 + pix(&I, Y + 1, X) * w10 +
                                 pix(&I, Y + 1, X + 1) * w11,
                             W_BITS - 5);
          int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
          scharr(&I, Y, X, &dx00, &dy00);
          scharr(&I, Y, X + 1, &dx01, &dy01);
          scharr(&I, Y + 1, X, &dx10, &dy10);
          scharr(&I, Y + 1, X + 1, &dx11, &dy11);
          int ix = descale(dx00 * w00 + dx01 * w01 + dx10 * w10 + dx11 * w11, W_BITS);
          int iy = descale(dy00 * w00 + dy01 * w01 + dy10 * w10 + dy11 * w11, W_BITS);
          short* o = Ibuf + 3 * (y * win + x);
          o[0] = (short)ival;
          o[1] = (short)ix;
          o[2] = (short)iy;
          sA11 += (long long)ix * ix;
          sA12 += (long long)ix * iy;
          sA22 += (long long)iy * iy;
        }
      float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww);
      if (minEig < (float)min_eig_thr || D < 1.1920929e-07f) {
        if (level == 0) status[i] = 0;
        continue;
      }
      D = 1.f / D;
      qx -= half;
      qy -= half;
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < max_iter; ++j) {
        int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
        if (iqx < -win || iqx >= J.W || iqy < -win || iqy >= J.H) {
          if (level == 0) status[i] = 0;
          break;
        }
        a = qx - (float)iqx;
        b = qy - (float)iqy;
        weights(a, b, &w00, &w01, &w10, &w11);
        long long sb1 = 0, sb2 = 0;
        for (int y = 0; y < win; ++y)
          for (int x = 0; x < win; ++x) {
            int X = iqx + x, Y = iqy + y;
            int jv = descale(pix(&J, Y, X) * w00 + pix(&J, Y, X + 1) * w01 + pix(&J, Y + 1, X) * w10 +
                                 pix(&J, Y + 1, X + 1) * w11,
                             W_BITS - 5);
            const short* o = Ibuf + 3 * (y * win + x);
            int diff = jv - o[0];
            sb1 += (long long)diff * o[1];
            sb2 += (long long)diff * o[2];
          }
        float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        float ddx = (A12 * b2 - A22 * b1) * D;
        float ddy = (A12 * b1 - A11 * b2) * D;
        qx += ddx;
        qy += ddy;
        nx = qx + half;
        ny = qy + half;
        if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= eps2) break;
        if (j > 0 && fabsf(ddx + pdx) < 0.01f && fabsf(ddy + pdy) < 0.01f) {
          nx -= ddx * 0.5f;
          ny -= ddy * 0.5f;
          break;
        }
        pdx = ddx;
        pdy = ddy;
      }
      if (status[i] && level == 0) {
        float ex = nx - half, ey = ny - half;
        int iex = (int)floorf(ex), iey = (int)floorf(ey);
        if (iex < -win || iex >= J.W || iey < -win || iey >= J.H) {
          status[i] = 0;
          continue;
        }
        a = ex - (float)iex;
        b = ey - (float)iey;
        weights(a, b, &w00, &w01, &w10, &w11);
        long long s = 0;
        for (int y = 0; y < win; ++y)
          for (int x = 0; x < win; ++x) {
            int X = iex + x, Y = iey + y;
            int jv = descale(pix(&J, Y, X) * w00 + pix(&J, Y, X + 1) * w01 + pix(&J, Y + 1, X) * w10 +
                                 pix(&J, Y + 1, X + 1) * w11,
                             W_BITS - 5);
            int diff = jv - Ibuf[3 * (y * win + x)];
            s += diff < 0 ? -diff : diff;
          }
        err[i] = (float)s / (float)(32 * ww);
      }
    }
    next_xy[2 * i] = nx;
    next_xy[2 * i + 1] = ny;
  }
  free(Ibuf);
}

/* number of levels OpenCV would build: stops when the next level is not larger than the window */
int oracle_klt_num_levels(int H, int W, int win, int max_level) {
  int levels = 1;
  int h = H, w = W;
  for (int l = 0; l < max_level; ++l) {
    h = (h + 1) / 2;
    w = (w + 1) / 2;
    if (w <= win || h <= win) break;
    ++levels;
  }
  return levels;
}

/* convenience: builds both pyramids, tracks, frees */
void oracle_klt_track(const uint8_t* prev, const uint8_t* next, int H, int W, const float* prev_xy, int N,
                      int win, int max_level, int max_iter, double eps, double min_eig_thr, float* next_xy,
                      uint8_t* status, float* err) {
  int nl = oracle_klt_num_levels(H, W, win, max_level);
  const uint8_t* pp[16];
  const uint8_t* np_[16];
  uint8_t* own[32];
  int Hs[16], Ws[16], n_own = 0;
  pp[0] = prev;
  np_[0] = next;
  Hs[0] = H;
  Ws[0] = W;
  for (int l = 1; l < nl; ++l) {
    Hs[l] = (Hs[l - 1] + 1) / 2;
    Ws[l] = (Ws[l - 1] + 1) / 2;
    uint8_t* a = (uint8_t*)malloc((size_t)Hs[l] * Ws[l]);
    uint8_t* b = (uint8_t*)malloc((size_t)Hs[l] * Ws[l]);
    oracle_pyr_down(pp[l - 1], Hs[l - 1], Ws[l - 1], a);
    oracle_pyr_down(np_[l - 1], Hs[l - 1], Ws[l - 1], b);
    pp[l] = a;
    np_[l] = b;
    own[n_own++] = a;
    own[n_own++] = b;
  }
  oracle_klt_track_pyr(pp, np_, Hs, Ws, nl, prev_xy, N, win, max_iter, eps, min_eig_thr, next_xy, status, err);
  for (int k = 0; k < n_own; ++k) free(own[k]);
}
