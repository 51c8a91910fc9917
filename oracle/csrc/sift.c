/*
 * CPU oracle: SIFT keypoints + 128-D descriptors.
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product.
 *
 * Reference call site: src/vo/features/sift.py:10,17
 *     self.sift = cv2.SIFT_create();  kp, desc = self.sift.detectAndCompute(image, None)
 * (only kp.pt and the (n, 128) float32 descriptors are used, sift.py:18-19).
 * PARITY UNPINNED against OpenCV (opencv-python==4.8.1.78 is absent, the reference has no
 * SIFT test).  This restates Lowe's algorithm with cv2.SIFT_create()'s defaults as OpenCV
 * structures it: image doubled (bilinear) and blurred to sigma 1.6; octaves of 3 layers
 * (6 Gaussian images, separable kernels of cvRound(8 sigma + 1) | 1 taps, reflect-101);
 * DoG; 26-neighbour extrema beyond |D| > floor(0.5 * 0.04 / 3 * 255) with a 5-pixel border;
 * up to 5 quadratic refinement steps; contrast 0.04 and edge-ratio 10 tests; 36-bin
 * orientation histogram (radius 4.5 s, Gaussian 1.5 s, [1 4 6 4 1]/16 smoothing, peaks >= 0.8
 * max, parabolic interpolation); 4x4x8 descriptor (trilinear binning, Gaussian window,
 * 0.2 clamp, x512, saturate to 0..255, stored as float32).  Keypoints are returned in the
 * original image's coordinates, sorted by (x, y, size desc, angle, response desc) with exact
 * duplicates removed (KeyPointsFilter::removeDuplicatedSorted).
 * exp and atan2 are small polynomial routines built from + - * / only (atan2 follows the
 * polynomial of cv::fastAtan2), so that a second implementation with the same operation
 * order reproduces every value bit for bit.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NOL 3            /* layers per octave */
#define NG (NOL + 3)     /* Gaussian images per octave */
#define BORDER 5
#define MAX_OCT 12

static inline int refl(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = c < 0 ? -c : 2 * (n - 1) - c;
  return c;
}

/* exp(x) for x <= 0: 2^(x log2 e) = 2^n 2^f, f in [-0.5, 0.5], degree-6 Taylor of 2^f */
static inline float sift_exp(float x) {
  if (x < -87.0f) return 0.0f;
  const float t = x * 1.4426950408889634f;
  const float n = rintf(t);
  const float f = (t - n) * 0.6931471805599453f;
  float p = 1.0f / 720.0f;
  p = p * f + 1.0f / 120.0f;
  p = p * f + 1.0f / 24.0f;
  p = p * f + 1.0f / 6.0f;
  p = p * f + 0.5f;
  p = p * f + 1.0f;
  p = p * f + 1.0f;
  return ldexpf(p, (int)n);
}

/* angle of (x, y) in degrees, [0, 360) -- the polynomial of cv::fastAtan2 */
static inline float sift_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + 2.220446049250313e-16f);
    c2 = c * c;
    a = (((-2.5397272f * c2 + 8.9140005f) * c2 - 18.667446f) * c2 + 57.283627f) * c;
  } else {
    c = ax / (ay + 2.220446049250313e-16f);
    c2 = c * c;
    a = 90.0f - (((-2.5397272f * c2 + 8.9140005f) * c2 - 18.667446f) * c2 + 57.283627f) * c;
  }
  if (x < 0) a = 180.0f - a;
  if (y < 0) a = 360.0f - a;
  return a;
}

typedef struct {
  float x, y, size, angle, response;
  int octave, layer;    /* octave index in the (doubled) pyramid, layer 1..NOL */
  float oct_x, oct_y;   /* position in the octave image */
} kp_t;

static int make_kernel(double sigma, float* w) {
  int ks = (int)lrint(sigma * 8 + 1) | 1;
  int r = ks / 2;
  double sum = 0, tmp[64];
  for (int i = 0; i < ks; ++i) {
    double d = i - r;
    tmp[i] = exp(-d * d / (2 * sigma * sigma));
    sum += tmp[i];
  }
  for (int i = 0; i < ks; ++i) w[i] = (float)(tmp[i] / sum);
  return r;
}

static void blur(const float* src, int H, int W, const float* w, int r, float* tmp, float* dst) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float s = 0.f;
      for (int k = -r; k <= r; ++k) s += w[k + r] * src[(size_t)y * W + refl(x + k, W)];
      tmp[(size_t)y * W + x] = s;
    }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float s = 0.f;
      for (int k = -r; k <= r; ++k) s += w[k + r] * tmp[(size_t)refl(y + k, H) * W + x];
      dst[(size_t)y * W + x] = s;
    }
}

/* 3x3 solve by Gaussian elimination with partial pivoting; returns 0 if singular */
static int solve3(float A[3][3], float b[3], float x[3]) {
  float M[3][4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i][j] = A[i][j];
    M[i][3] = b[i];
  }
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r2 = c + 1; r2 < 3; ++r2)
      if (fabsf(M[r2][c]) > fabsf(M[p][c])) p = r2;
    if (fabsf(M[p][c]) < 1.1920929e-07f) return 0;
    if (p != c)
      for (int j = 0; j < 4; ++j) {
        float t = M[p][j];
        M[p][j] = M[c][j];
        M[c][j] = t;
      }
    for (int r2 = c + 1; r2 < 3; ++r2) {
      float f = M[r2][c] / M[c][c];
      for (int j = c; j < 4; ++j) M[r2][j] -= f * M[c][j];
    }
  }
  x[2] = M[2][3] / M[2][2];
  x[1] = (M[1][3] - M[1][2] * x[2]) / M[1][1];
  x[0] = (M[0][3] - M[0][1] * x[1] - M[0][2] * x[2]) / M[0][0];
  return 1;
}

typedef struct {
  int H, W;
  float* g[NG];
  float* d[NG - 1];
} octave_t;

static int refine(const octave_t* o, int octv, int* pl, int* pr, int* pc, kp_t* kp, float contrast_thr, float edge_thr,
                  float sigma) {
  const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
  int layer = *pl, r = *pr, c = *pc, i;
  float xi = 0, xr = 0, xc = 0;
  const int W = o->W, H = o->H;
  for (i = 0; i < 5; ++i) {
    const float* img = o->d[layer];
    const float* prv = o->d[layer - 1];
    const float* nxt = o->d[layer + 1];
    size_t p = (size_t)r * W + c;
    float dD[3] = {(img[p + 1] - img[p - 1]) * deriv_scale, (img[p + W] - img[p - W]) * deriv_scale,
                   (nxt[p] - prv[p]) * deriv_scale};
    float v2 = img[p] * 2;
    float dxx = (img[p + 1] + img[p - 1] - v2) * second, dyy = (img[p + W] + img[p - W] - v2) * second,
          dss = (nxt[p] + prv[p] - v2) * second;
    float dxy = (img[p + W + 1] - img[p + W - 1] - img[p - W + 1] + img[p - W - 1]) * cross;
    float dxs = (nxt[p + 1] - nxt[p - 1] - prv[p + 1] + prv[p - 1]) * cross;
    float dys = (nxt[p + W] - nxt[p - W] - prv[p + W] + prv[p - W]) * cross;
    float Hm[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    float X[3];
    if (!solve3(Hm, dD, X)) return 0;
    xi = -X[2];
    xr = -X[1];
    xc = -X[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    if (fabsf(xi) > 7e8f || fabsf(xr) > 7e8f || fabsf(xc) > 7e8f) return 0;
    c += (int)rintf(xc);
    r += (int)rintf(xr);
    layer += (int)rintf(xi);
    if (layer < 1 || layer > NOL || c < BORDER || c >= W - BORDER || r < BORDER || r >= H - BORDER) return 0;
  }
  if (i >= 5) return 0;
  {
    const float* img = o->d[layer];
    const float* prv = o->d[layer - 1];
    const float* nxt = o->d[layer + 1];
    size_t p = (size_t)r * W + c;
    float dD[3] = {(img[p + 1] - img[p - 1]) * deriv_scale, (img[p + W] - img[p - W]) * deriv_scale,
                   (nxt[p] - prv[p]) * deriv_scale};
    float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
    float contr = img[p] * img_scale + t * 0.5f;
    if (fabsf(contr) * NOL < contrast_thr) return 0;
    float v2 = img[p] * 2.f;
    float dxx = (img[p + 1] + img[p - 1] - v2) * second, dyy = (img[p + W] + img[p - W] - v2) * second;
    float dxy = (img[p + W + 1] - img[p + W - 1] - img[p - W + 1] + img[p - W - 1]) * cross;
    float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return 0;
    kp->oct_x = c + xc;
    kp->oct_y = r + xr;
    kp->x = (c + xc) * (float)(1 << octv);
    kp->y = (r + xr) * (float)(1 << octv);
    kp->octave = octv;
    kp->layer = layer;
    kp->size = sigma * sift_exp(((layer + xi) / NOL) * 0.6931471805599453f) * (float)(1 << octv) * 2;
    kp->response = fabsf(contr);
  }
  *pl = layer;
  *pr = r;
  *pc = c;
  return 1;
}

/* orientation histogram around (r, c) of Gaussian image g; returns smoothed hist and its max */
static float ori_hist(const float* g, int H, int W, int c, int r, int radius, float sigma, float* hist) {
  float tmp[36];
  for (int i = 0; i < 36; ++i) tmp[i] = 0.f;
  const float expf_scale = -1.f / (2.f * sigma * sigma);
  for (int i = -radius; i <= radius; ++i) {
    int y = r + i;
    if (y <= 0 || y >= H - 1) continue;
    for (int j = -radius; j <= radius; ++j) {
      int x = c + j;
      if (x <= 0 || x >= W - 1) continue;
      float dx = g[(size_t)y * W + x + 1] - g[(size_t)y * W + x - 1];
      float dy = g[(size_t)(y - 1) * W + x] - g[(size_t)(y + 1) * W + x];
      float w = sift_exp((float)(i * i + j * j) * expf_scale);
      float ori = sift_atan2(dy, dx);
      float mag = sqrtf(dx * dx + dy * dy);
      int bin = (int)rintf(0.1f * ori);
      if (bin >= 36) bin -= 36;
      if (bin < 0) bin += 36;
      tmp[bin] += w * mag;
    }
  }
  float mx = 0.f;
  for (int i = 0; i < 36; ++i) {
    float h = (tmp[(i + 34) % 36] + tmp[(i + 2) % 36]) * (1.f / 16.f) + (tmp[(i + 35) % 36] + tmp[(i + 1) % 36]) * (4.f / 16.f) +
              tmp[i] * (6.f / 16.f);
    hist[i] = h;
    if (h > mx) mx = h;
  }
  return mx;
}

static void descriptor(const float* g, int H, int W, float px, float py, float ori_deg, float scl, float* dst) {
  const int d = 4, n = 8;
  const int pxi = (int)rintf(px), pyi = (int)rintf(py);
  const float rad = ori_deg * 0.017453292519943295f;
  /* sin/cos through the same exp-free route on both sides: use sinf/cosf replacement by polynomial of degrees */
  float cos_t, sin_t;
  {
    /* range-reduce to [-pi, pi] then to [-pi/2, pi/2]; odd/even Taylor, deterministic */
    float a = rad;
    while (a > 3.14159265358979f) a -= 6.28318530717959f;
    while (a < -3.14159265358979f) a += 6.28318530717959f;
    float sgn = 1.f;
    if (a > 1.5707963267949f) {
      a = 3.14159265358979f - a;
      sgn = -1.f;
    } else if (a < -1.5707963267949f) {
      a = -3.14159265358979f - a;
      sgn = -1.f;
    }
    float a2 = a * a;
    sin_t = a * (1.f + a2 * (-1.f / 6 + a2 * (1.f / 120 + a2 * (-1.f / 5040 + a2 * (1.f / 362880 + a2 * (-1.f / 39916800))))));
    cos_t = sgn * (1.f + a2 * (-0.5f + a2 * (1.f / 24 + a2 * (-1.f / 720 + a2 * (1.f / 40320 + a2 * (-1.f / 3628800 + a2 * (1.f / 479001600)))))));
  }
  const float bins_per_deg = n / 360.f;
  const float exp_scale = -1.f / (d * d * 0.5f);
  const float hist_width = 3.f * scl;
  int radius = (int)rintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
  const int maxr = (int)sqrt((double)H * H + (double)W * W);
  if (radius > maxr) radius = maxr;
  cos_t /= hist_width;
  sin_t /= hist_width;
  float hist[6 * 6 * 10];
  for (int i = 0; i < 360; ++i) hist[i] = 0.f;
  for (int i = -radius; i <= radius; ++i)
    for (int j = -radius; j <= radius; ++j) {
      float c_rot = j * cos_t - i * sin_t;
      float r_rot = j * sin_t + i * cos_t;
      float rbin = r_rot + d / 2 - 0.5f;
      float cbin = c_rot + d / 2 - 0.5f;
      int r = pyi + i, c = pxi + j;
      if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < H - 1 && c > 0 && c < W - 1)) continue;
      float dx = g[(size_t)r * W + c + 1] - g[(size_t)r * W + c - 1];
      float dy = g[(size_t)(r - 1) * W + c] - g[(size_t)(r + 1) * W + c];
      float wgt = sift_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
      float ang = sift_atan2(dy, dx);
      float mag = sqrtf(dx * dx + dy * dy) * wgt;
      float obin = (ang - ori_deg) * bins_per_deg;
      int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin), o0 = (int)floorf(obin);
      rbin -= r0;
      cbin -= c0;
      obin -= o0;
      if (o0 < 0) o0 += n;
      if (o0 >= n) o0 -= n;
      float v_r1 = mag * rbin, v_r0 = mag - v_r1;
      float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
      float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
      float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
      float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
      float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
      float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
      int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
      hist[idx] += v_rco000;
      hist[idx + 1] += v_rco001;
      hist[idx + (n + 2)] += v_rco010;
      hist[idx + (n + 3)] += v_rco011;
      hist[idx + (d + 2) * (n + 2)] += v_rco100;
      hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
      hist[idx + (d + 3) * (n + 2)] += v_rco110;
      hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
    }
  float raw[128];
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
      hist[idx] += hist[idx + n];
      hist[idx + 1] += hist[idx + n + 1];
      for (int k = 0; k < n; ++k) raw[(i * d + j) * n + k] = hist[idx + k];
    }
  float nrm2 = 0;
  for (int k = 0; k < 128; ++k) nrm2 += raw[k] * raw[k];
  float thr = sqrtf(nrm2) * 0.2f;
  nrm2 = 0;
  for (int k = 0; k < 128; ++k) {
    float v = raw[k] < thr ? raw[k] : thr;
    raw[k] = v;
    nrm2 += v * v;
  }
  float s = 512.f / fmaxf(sqrtf(nrm2), 1.1920929e-07f);
  for (int k = 0; k < 128; ++k) {
    float v = rintf(raw[k] * s);
    dst[k] = v < 0 ? 0.f : (v > 255.f ? 255.f : v);
  }
}

static int kp_less(const void* pa, const void* pb) {
  const float* a = (const float*)pa;
  const float* b = (const float*)pb;   /* x, y, size, angle, response, octave */
  if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
  if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
  if (a[2] != b[2]) return a[2] > b[2] ? -1 : 1;
  if (a[3] != b[3]) return a[3] < b[3] ? -1 : 1;
  if (a[4] != b[4]) return a[4] > b[4] ? -1 : 1;
  if (a[5] != b[5]) return a[5] > b[5] ? -1 : 1;
  return 0;
}

/*
 * img: H x W uint8.  kp_out: cap x 6 float32 (x, y, size, angle, response, octave), desc_out: cap x 128 float32.
 * Returns the number of keypoints (<= cap; when more are found the `cap` strongest by response are kept).
 */
int oracle_sift(const uint8_t* img, int H, int W, int cap, float* kp_out, float* desc_out) {
  const float sigma = 1.6f, contrast_thr = 0.04f, edge_thr = 10.f;
  const int W0 = W * 2, H0 = H * 2;
  float* base = (float*)malloc(sizeof(float) * (size_t)W0 * H0);
  for (int y = 0; y < H0; ++y) {
    float sy = (y + 0.5f) * 0.5f - 0.5f;
    int y0 = (int)floorf(sy);
    float fy = sy - y0;
    int ya = y0 < 0 ? 0 : (y0 >= H ? H - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 >= H ? H - 1 : y0 + 1);
    for (int x = 0; x < W0; ++x) {
      float sx = (x + 0.5f) * 0.5f - 0.5f;
      int x0 = (int)floorf(sx);
      float fx = sx - x0;
      int xa = x0 < 0 ? 0 : (x0 >= W ? W - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 >= W ? W - 1 : x0 + 1);
      float top = (float)img[(size_t)ya * W + xa] * (1.f - fx) + (float)img[(size_t)ya * W + xb] * fx;
      float bot = (float)img[(size_t)yb * W + xa] * (1.f - fx) + (float)img[(size_t)yb * W + xb] * fx;
      base[(size_t)y * W0 + x] = top * (1.f - fy) + bot * fy;
    }
  }
  int n_oct = (int)lrint(log((double)(W0 < H0 ? W0 : H0)) / log(2.0) - 2);
  if (n_oct > MAX_OCT) n_oct = MAX_OCT;
  {
    int w = W0, h = H0, k = 0;
    while (k < n_oct && w >= 2 * BORDER + 3 && h >= 2 * BORDER + 3) {
      ++k;
      w /= 2;
      h /= 2;
    }
    n_oct = k;
  }
  double sig[NG];
  sig[0] = sigma;
  const double kf = pow(2.0, 1.0 / NOL);
  for (int i = 1; i < NG; ++i) {
    double sp = pow(kf, i - 1) * sigma, st = sp * kf;
    sig[i] = sqrt(st * st - sp * sp);
  }
  float wk[NG][64];
  int rk[NG];
  {
    double sd = sqrt(fmax((double)sigma * sigma - 1.0, 0.01));
    rk[0] = make_kernel(sd, wk[0]);
    for (int i = 1; i < NG; ++i) rk[i] = make_kernel(sig[i], wk[i]);
  }
  octave_t* oct = (octave_t*)calloc(n_oct, sizeof(octave_t));
  int w = W0, h = H0;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)W0 * H0);
  for (int o = 0; o < n_oct; ++o) {
    oct[o].W = w;
    oct[o].H = h;
    for (int i = 0; i < NG; ++i) oct[o].g[i] = (float*)malloc(sizeof(float) * (size_t)w * h);
    for (int i = 0; i < NG - 1; ++i) oct[o].d[i] = (float*)malloc(sizeof(float) * (size_t)w * h);
    if (o == 0) {
      blur(base, h, w, wk[0], rk[0], tmp, oct[o].g[0]);
    } else {
      const float* src = oct[o - 1].g[NOL];
      int pw = oct[o - 1].W;
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) oct[o].g[0][(size_t)y * w + x] = src[(size_t)(2 * y) * pw + 2 * x];
    }
    for (int i = 1; i < NG; ++i) blur(oct[o].g[i - 1], h, w, wk[i], rk[i], tmp, oct[o].g[i]);
    for (int i = 0; i < NG - 1; ++i)
      for (size_t p = 0; p < (size_t)w * h; ++p) oct[o].d[i][p] = oct[o].g[i + 1][p] - oct[o].g[i][p];
    w /= 2;
    h /= 2;
  }
  free(tmp);
  free(base);

  const float threshold = floorf(0.5f * contrast_thr / NOL * 255.f);
  int cap_all = 1 << 16, n_all = 0;
  float* all = (float*)malloc(sizeof(float) * 8 * (size_t)cap_all);   /* x,y,size,angle,response,octave,oct_x/y packed later */
  kp_t* kps = (kp_t*)malloc(sizeof(kp_t) * (size_t)cap_all);
  for (int o = 0; o < n_oct; ++o) {
    const int Wo = oct[o].W, Ho = oct[o].H;
    for (int layer = 1; layer <= NOL; ++layer) {
      const float* img2 = oct[o].d[layer];
      const float* prv = oct[o].d[layer - 1];
      const float* nxt = oct[o].d[layer + 1];
      for (int r = BORDER; r < Ho - BORDER; ++r)
        for (int c = BORDER; c < Wo - BORDER; ++c) {
          size_t p = (size_t)r * Wo + c;
          float val = img2[p];
          if (!(fabsf(val) > threshold)) continue;
          int is_ext = 1;
          if (val > 0) {
            for (int dy = -1; dy <= 1 && is_ext; ++dy)
              for (int dx = -1; dx <= 1; ++dx) {
                size_t q = p + dy * Wo + dx;
                if (val < img2[q] || val < prv[q] || val < nxt[q]) {
                  is_ext = 0;
                  break;
                }
              }
          } else {
            for (int dy = -1; dy <= 1 && is_ext; ++dy)
              for (int dx = -1; dx <= 1; ++dx) {
                size_t q = p + dy * Wo + dx;
                if (val > img2[q] || val > prv[q] || val > nxt[q]) {
                  is_ext = 0;
                  break;
                }
              }
          }
          if (!is_ext) continue;
          int l2 = layer, r2 = r, c2 = c;
          kp_t kp;
          if (!refine(&oct[o], o, &l2, &r2, &c2, &kp, contrast_thr, edge_thr, sigma)) continue;
          float scl_octv = kp.size * 0.5f / (float)(1 << o);
          float hist[36];
          float mx = ori_hist(oct[o].g[l2], Ho, Wo, c2, r2, (int)rintf(4.5f * scl_octv), 1.5f * scl_octv, hist);
          float mag_thr = mx * 0.8f;
          for (int j = 0; j < 36; ++j) {
            int l = j > 0 ? j - 1 : 35, rr = j < 35 ? j + 1 : 0;
            if (hist[j] > hist[l] && hist[j] > hist[rr] && hist[j] >= mag_thr) {
              float bin = j + 0.5f * (hist[l] - hist[rr]) / (hist[l] - 2 * hist[j] + hist[rr]);
              bin = bin < 0 ? 36 + bin : (bin >= 36 ? bin - 36 : bin);
              float angle = 360.f - (360.f / 36) * bin;
              if (fabsf(angle - 360.f) < 1.1920929e-07f) angle = 0.f;
              if (n_all < cap_all) {
                kps[n_all] = kp;
                kps[n_all].angle = angle;
                ++n_all;
              }
            }
          }
        }
    }
  }
  /* descriptors in the octave image, then map to the original image (first octave = -1: halve) */
  float* rows = (float*)malloc(sizeof(float) * (size_t)(6 + 128 + 1) * (n_all > 0 ? n_all : 1));
  for (int k = 0; k < n_all; ++k) {
    kp_t* q = &kps[k];
    float* row = rows + (size_t)k * 135;
    row[0] = q->x * 0.5f;
    row[1] = q->y * 0.5f;
    row[2] = q->size * 0.5f;
    row[3] = q->angle;
    row[4] = q->response;
    row[5] = (float)(q->octave - 1);
    float scl = q->size * 0.5f / (float)(1 << q->octave);
    float ang = 360.f - q->angle;
    if (fabsf(ang - 360.f) < 1.1920929e-07f) ang = 0.f;
    descriptor(oct[q->octave].g[q->layer], oct[q->octave].H, oct[q->octave].W, q->oct_x, q->oct_y, ang, scl, row + 6);
    row[134] = 0;
  }
  qsort(rows, n_all, sizeof(float) * 135, kp_less);
  /* exact duplicates (same x, y, size, angle) removed */
  int n = 0;
  for (int k = 0; k < n_all; ++k) {
    float* row = rows + (size_t)k * 135;
    if (n > 0) {
      float* prev = rows + (size_t)(n - 1) * 135;
      if (prev[0] == row[0] && prev[1] == row[1] && prev[2] == row[2] && prev[3] == row[3]) continue;
    }
    if (n != k) memmove(rows + (size_t)n * 135, row, sizeof(float) * 135);
    ++n;
  }
  if (n > cap) {
    /* keep the `cap` strongest by response (ties by sorted position), preserving the sorted order */
    float* resp = (float*)malloc(sizeof(float) * n);
    for (int k = 0; k < n; ++k) resp[k] = rows[(size_t)k * 135 + 4];
    /* threshold = cap-th largest response */
    float* cp = (float*)malloc(sizeof(float) * n);
    memcpy(cp, resp, sizeof(float) * n);
    for (int i = 0; i < cap; ++i) {   /* partial selection sort is fine for test sizes */
      int m = i;
      for (int j = i + 1; j < n; ++j)
        if (cp[j] > cp[m]) m = j;
      float t = cp[i];
      cp[i] = cp[m];
      cp[m] = t;
    }
    float thr = cp[cap - 1];
    int above = 0;
    for (int k = 0; k < n; ++k) above += resp[k] > thr;
    int ties = cap - above, m = 0;
    for (int k = 0; k < n; ++k) {
      int keep = resp[k] > thr || (resp[k] == thr && ties-- > 0);
      if (keep) {
        if (m != k) memmove(rows + (size_t)m * 135, rows + (size_t)k * 135, sizeof(float) * 135);
        ++m;
      }
    }
    n = m;
    free(resp);
    free(cp);
  }
  for (int k = 0; k < n; ++k) {
    memcpy(kp_out + (size_t)k * 6, rows + (size_t)k * 135, sizeof(float) * 6);
    memcpy(desc_out + (size_t)k * 128, rows + (size_t)k * 135 + 6, sizeof(float) * 128);
  }
  free(rows);
  free(kps);
  free(all);
  for (int o = 0; o < n_oct; ++o) {
    for (int i = 0; i < NG; ++i) free(oct[o].g[i]);
    for (int i = 0; i < NG - 1; ++i) free(oct[o].d[i]);
  }
  free(oct);
  return n;
}
