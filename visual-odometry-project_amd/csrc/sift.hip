// SIFT keypoints + 128-D descriptors for gfx950.
//
// Reference call site: src/vo/features/sift.py:10,17 (cv2.SIFT_create().detectAndCompute).
// The definition (Lowe's algorithm with cv2.SIFT_create()'s defaults, restated in
// oracle/csrc/sift.c) is followed operation for operation so that results agree with the
// oracle bit for bit: separable Gaussian taps summed in tap order without FMA, polynomial
// exp / atan2 / sin / cos built from + - * /, sequential accumulation of each orientation
// histogram and descriptor in pixel raster order.
//   scale space   image-wide kernels: bilinear 2x up-sampling, row / column blur passes,
//                 2:1 decimation, DoG                                     (HBM-bound)
//   detection     one lane per DoG pixel, compacting 26-neighbour extrema into a list
//   refinement    one lane per candidate: <= 5 quadratic-fit steps, contrast / edge tests,
//                 36-bin orientation histogram, one keypoint per accepted peak
//   description   one lane per keypoint, 4x4x8 trilinear histogram
// Ordering (by x, y, ...), duplicate removal and the optional strongest-N cap are applied on
// the host to the few thousand resulting rows.
#include <algorithm>
#include <cmath>

#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int NOL = 3, NG = NOL + 3, BORDER = 5, MAX_OCT = 12, MAX_TAPS = 64;

struct taps_t {
  float w[MAX_TAPS];
  int r;
};

struct skp_t {   // one keypoint row on the device
  float x, y, size, angle, response, octave;   // original-image coordinates
  float oct_x, oct_y;                          // octave-image coordinates
  int oct, layer;
};

__device__ __forceinline__ int refl(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = c < 0 ? -c : 2 * (n - 1) - c;
  return c;
}

__device__ __forceinline__ float sift_exp(float x) {
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

__device__ __forceinline__ float sift_atan2(float y, float x) {
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

// ---------------- scale space ----------------
__global__ __launch_bounds__(256) void upsample2_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        float* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int W0 = 2 * W, H0 = 2 * H;
  if (x >= W0 || y >= H0) return;
  const float sy = (y + 0.5f) * 0.5f - 0.5f, sx = (x + 0.5f) * 0.5f - 0.5f;
  const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
  const float fy = sy - y0, fx = sx - x0;
  const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
  const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
  const float top = (float)img[(size_t)ya * W + xa] * (1.f - fx) + (float)img[(size_t)ya * W + xb] * fx;
  const float bot = (float)img[(size_t)yb * W + xa] * (1.f - fx) + (float)img[(size_t)yb * W + xb] * fx;
  out[(size_t)y * W0 + x] = top * (1.f - fy) + bot * fy;
}

template <bool ROWS>
__global__ __launch_bounds__(256) void blur_kernel(const float* __restrict__ src, int H, int W, taps_t k,
                                                   float* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  float s = 0.f;
  if (ROWS) {
    const float* row = src + (size_t)y * W;
    for (int j = -k.r; j <= k.r; ++j) s += k.w[j + k.r] * row[refl(x + j, W)];
  } else {
    for (int j = -k.r; j <= k.r; ++j) s += k.w[j + k.r] * src[(size_t)refl(y + j, H) * W + x];
  }
  dst[(size_t)y * W + x] = s;
}

__global__ __launch_bounds__(256) void decimate_kernel(const float* __restrict__ src, int pw, int H, int W,
                                                       float* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < W && y < H) dst[(size_t)y * W + x] = src[(size_t)(2 * y) * pw + 2 * x];
}

__global__ __launch_bounds__(256) void dog_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                  float* __restrict__ d) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = b[i] - a[i];
}

// ---------------- detection ----------------
struct oct_t {
  const float* g[NG];
  const float* d[NG - 1];
  int H, W, o;
};

__global__ __launch_bounds__(256) void extrema_kernel(oct_t O, int layer, float threshold, int* __restrict__ cand,
                                                      unsigned* __restrict__ n_cand, unsigned cap) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c < BORDER || c >= O.W - BORDER || r < BORDER || r >= O.H - BORDER) return;
  const float* img = O.d[layer];
  const float* prv = O.d[layer - 1];
  const float* nxt = O.d[layer + 1];
  const size_t p = (size_t)r * O.W + c;
  const float val = img[p];
  if (!(fabsf(val) > threshold)) return;
  bool ext = true;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const size_t q = p + dy * O.W + dx;
      if (val > 0 ? (val < img[q] || val < prv[q] || val < nxt[q]) : (val > img[q] || val > prv[q] || val > nxt[q]))
        ext = false;
    }
  if (!ext) return;
  const unsigned pos = atomicAdd(n_cand, 1u);
  if (pos < cap) {
    cand[3 * pos] = layer;
    cand[3 * pos + 1] = r;
    cand[3 * pos + 2] = c;
  }
}

__device__ bool solve3(float A[3][3], float b[3], float x[3]) {
  float M[3][4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i][j] = A[i][j];
    M[i][3] = b[i];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int p = c;
#pragma unroll
    for (int r2 = c + 1; r2 < 3; ++r2)
      if (fabsf(M[r2][c]) > fabsf(M[p][c])) p = r2;
    if (fabsf(M[p][c]) < 1.1920929e-07f) return false;
    if (p != c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = M[c][j];
        // static indexing only: select the pivot row by comparisons
        float pv = (p == 1) ? M[1][j] : M[2][j];
        M[c][j] = pv;
        if (p == 1) M[1][j] = t;
        else M[2][j] = t;
      }
    }
#pragma unroll
    for (int r2 = c + 1; r2 < 3; ++r2) {
      const float f = M[r2][c] / M[c][c];
#pragma unroll
      for (int j = c; j < 4; ++j) M[r2][j] -= f * M[c][j];
    }
  }
  x[2] = M[2][3] / M[2][2];
  x[1] = (M[1][3] - M[1][2] * x[2]) / M[1][1];
  x[0] = (M[0][3] - M[0][1] * x[1] - M[0][2] * x[2]) / M[0][0];
  return true;
}

__global__ __launch_bounds__(64) void refine_orient_kernel(oct_t O, const int* __restrict__ cand,
                                                           const unsigned* __restrict__ n_cand, unsigned cap,
                                                           float contrast_thr, float edge_thr, float sigma,
                                                           skp_t* __restrict__ out, unsigned* __restrict__ n_out,
                                                           unsigned cap_out) {
  const unsigned k = blockIdx.x * 64 + threadIdx.x;
  if (k >= min(*n_cand, cap)) return;
  int layer = cand[3 * k], r = cand[3 * k + 1], c = cand[3 * k + 2];
  const int W = O.W, H = O.H;
  const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
  float xi = 0, xr = 0, xc = 0;
  int i;
  for (i = 0; i < 5; ++i) {
    const float* img = O.d[layer];
    const float* prv = O.d[layer - 1];
    const float* nxt = O.d[layer + 1];
    const size_t p = (size_t)r * W + c;
    float dD[3] = {(img[p + 1] - img[p - 1]) * deriv_scale, (img[p + W] - img[p - W]) * deriv_scale,
                   (nxt[p] - prv[p]) * deriv_scale};
    const float v2 = img[p] * 2;
    const float dxx = (img[p + 1] + img[p - 1] - v2) * second, dyy = (img[p + W] + img[p - W] - v2) * second,
                dss = (nxt[p] + prv[p] - v2) * second;
    const float dxy = (img[p + W + 1] - img[p + W - 1] - img[p - W + 1] + img[p - W - 1]) * cross;
    const float dxs = (nxt[p + 1] - nxt[p - 1] - prv[p + 1] + prv[p - 1]) * cross;
    const float dys = (nxt[p + W] - nxt[p - W] - prv[p + W] + prv[p - W]) * cross;
    float Hm[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    float X[3];
    if (!solve3(Hm, dD, X)) return;
    xi = -X[2];
    xr = -X[1];
    xc = -X[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    if (fabsf(xi) > 7e8f || fabsf(xr) > 7e8f || fabsf(xc) > 7e8f) return;
    c += (int)rintf(xc);
    r += (int)rintf(xr);
    layer += (int)rintf(xi);
    if (layer < 1 || layer > NOL || c < BORDER || c >= W - BORDER || r < BORDER || r >= H - BORDER) return;
  }
  if (i >= 5) return;
  skp_t kp;
  {
    const float* img = O.d[layer];
    const float* prv = O.d[layer - 1];
    const float* nxt = O.d[layer + 1];
    const size_t p = (size_t)r * W + c;
    const float d0 = (img[p + 1] - img[p - 1]) * deriv_scale, d1 = (img[p + W] - img[p - W]) * deriv_scale,
                d2 = (nxt[p] - prv[p]) * deriv_scale;
    const float t = d0 * xc + d1 * xr + d2 * xi;
    const float contr = img[p] * img_scale + t * 0.5f;
    if (fabsf(contr) * NOL < contrast_thr) return;
    const float v2 = img[p] * 2.f;
    const float dxx = (img[p + 1] + img[p - 1] - v2) * second, dyy = (img[p + W] + img[p - W] - v2) * second;
    const float dxy = (img[p + W + 1] - img[p + W - 1] - img[p - W + 1] + img[p - W - 1]) * cross;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return;
    const float po = (float)(1 << O.o);
    kp.oct_x = c + xc;
    kp.oct_y = r + xr;
    kp.x = (c + xc) * po;
    kp.y = (r + xr) * po;
    kp.oct = O.o;
    kp.layer = layer;
    kp.size = sigma * sift_exp(((layer + xi) / NOL) * 0.6931471805599453f) * po * 2;
    kp.response = fabsf(contr);
  }
  // ---- orientation histogram (raster order, sequential) ----
  const float scl_octv = kp.size * 0.5f / (float)(1 << O.o);
  const int radius = (int)rintf(4.5f * scl_octv);
  const float osig = 1.5f * scl_octv;
  const float* g = O.g[layer];
  float tmp[36];
#pragma unroll
  for (int b = 0; b < 36; ++b) tmp[b] = 0.f;
  const float expf_scale = -1.f / (2.f * osig * osig);
  for (int ii = -radius; ii <= radius; ++ii) {
    const int y = r + ii;
    if (y <= 0 || y >= H - 1) continue;
    for (int jj = -radius; jj <= radius; ++jj) {
      const int x = c + jj;
      if (x <= 0 || x >= W - 1) continue;
      const float dx = g[(size_t)y * W + x + 1] - g[(size_t)y * W + x - 1];
      const float dy = g[(size_t)(y - 1) * W + x] - g[(size_t)(y + 1) * W + x];
      const float w = sift_exp((float)(ii * ii + jj * jj) * expf_scale);
      const float ori = sift_atan2(dy, dx);
      const float mag = sqrtf(dx * dx + dy * dy);
      int bin = (int)rintf(0.1f * ori);
      if (bin >= 36) bin -= 36;
      if (bin < 0) bin += 36;
      const float add = w * mag;
      // static-index accumulate keeps the histogram in registers
#pragma unroll
      for (int b = 0; b < 36; ++b)
        if (b == bin) tmp[b] += add;
    }
  }
  float hist[36];
  float mx = 0.f;
#pragma unroll
  for (int b = 0; b < 36; ++b) {
    const float h = (tmp[(b + 34) % 36] + tmp[(b + 2) % 36]) * (1.f / 16.f) +
                    (tmp[(b + 35) % 36] + tmp[(b + 1) % 36]) * (4.f / 16.f) + tmp[b] * (6.f / 16.f);
    hist[b] = h;
    if (h > mx) mx = h;
  }
  const float mag_thr = mx * 0.8f;
#pragma unroll
  for (int j = 0; j < 36; ++j) {
    const int l = j > 0 ? j - 1 : 35, rr = j < 35 ? j + 1 : 0;
    if (hist[j] > hist[l] && hist[j] > hist[rr] && hist[j] >= mag_thr) {
      float bin = j + 0.5f * (hist[l] - hist[rr]) / (hist[l] - 2 * hist[j] + hist[rr]);
      bin = bin < 0 ? 36 + bin : (bin >= 36 ? bin - 36 : bin);
      float angle = 360.f - (360.f / 36) * bin;
      if (fabsf(angle - 360.f) < 1.1920929e-07f) angle = 0.f;
      const unsigned pos = atomicAdd(n_out, 1u);
      if (pos < cap_out) {
        skp_t q = kp;
        q.angle = angle;
        out[pos] = q;
      }
    }
  }
}

// ---------------- description ----------------
struct pyr_ptrs {
  const float* g[MAX_OCT][NG];
  int H[MAX_OCT], W[MAX_OCT];
};

__global__ __launch_bounds__(64) void descriptor_kernel(pyr_ptrs P, const skp_t* __restrict__ kps, unsigned n,
                                                        float* __restrict__ rows /* n x 134 */) {
  const unsigned k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const skp_t q = kps[k];
  const float* g = P.g[q.oct][q.layer];
  const int H = P.H[q.oct], W = P.W[q.oct];
  float* row = rows + (size_t)k * 134;
  row[0] = q.x * 0.5f;
  row[1] = q.y * 0.5f;
  row[2] = q.size * 0.5f;
  row[3] = q.angle;
  row[4] = q.response;
  row[5] = (float)(q.oct - 1);
  const float scl = q.size * 0.5f / (float)(1 << q.oct);
  float ori_deg = 360.f - q.angle;
  if (fabsf(ori_deg - 360.f) < 1.1920929e-07f) ori_deg = 0.f;
  const int d = 4, n8 = 8;
  const int pxi = (int)rintf(q.oct_x), pyi = (int)rintf(q.oct_y);
  float cos_t, sin_t;
  {
    float a = ori_deg * 0.017453292519943295f;
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
    const float a2 = a * a;
    sin_t = a * (1.f + a2 * (-1.f / 6 + a2 * (1.f / 120 + a2 * (-1.f / 5040 + a2 * (1.f / 362880 + a2 * (-1.f / 39916800))))));
    cos_t = sgn * (1.f + a2 * (-0.5f + a2 * (1.f / 24 + a2 * (-1.f / 720 + a2 * (1.f / 40320 + a2 * (-1.f / 3628800 + a2 * (1.f / 479001600)))))));
  }
  const float bins_per_deg = n8 / 360.f;
  const float exp_scale = -1.f / (d * d * 0.5f);
  const float hist_width = 3.f * scl;
  int radius = (int)rintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
  const int maxr = (int)sqrt((double)H * H + (double)W * W);
  if (radius > maxr) radius = maxr;
  cos_t /= hist_width;
  sin_t /= hist_width;
  // the (d+2) x (d+2) x (n+2) histogram lives in the output row's scratch tail (global memory,
  // private to this lane): 360 floats
  float* hist = rows + (size_t)n * 134 + (size_t)k * 360;
  for (int i = 0; i < 360; ++i) hist[i] = 0.f;
  for (int i = -radius; i <= radius; ++i)
    for (int j = -radius; j <= radius; ++j) {
      const float c_rot = j * cos_t - i * sin_t;
      const float r_rot = j * sin_t + i * cos_t;
      float rbin = r_rot + d / 2 - 0.5f;
      float cbin = c_rot + d / 2 - 0.5f;
      const int r = pyi + i, c = pxi + j;
      if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < H - 1 && c > 0 && c < W - 1)) continue;
      const float dx = g[(size_t)r * W + c + 1] - g[(size_t)r * W + c - 1];
      const float dy = g[(size_t)(r - 1) * W + c] - g[(size_t)(r + 1) * W + c];
      const float wgt = sift_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
      const float ang = sift_atan2(dy, dx);
      const float mag = sqrtf(dx * dx + dy * dy) * wgt;
      float obin = (ang - ori_deg) * bins_per_deg;
      const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
      int o0 = (int)floorf(obin);
      rbin -= r0;
      cbin -= c0;
      obin -= o0;
      if (o0 < 0) o0 += n8;
      if (o0 >= n8) o0 -= n8;
      const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
      const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
      const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
      const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
      const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
      const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
      const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
      const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n8 + 2) + o0;
      hist[idx] += v_rco000;
      hist[idx + 1] += v_rco001;
      hist[idx + (n8 + 2)] += v_rco010;
      hist[idx + (n8 + 3)] += v_rco011;
      hist[idx + (d + 2) * (n8 + 2)] += v_rco100;
      hist[idx + (d + 2) * (n8 + 2) + 1] += v_rco101;
      hist[idx + (d + 3) * (n8 + 2)] += v_rco110;
      hist[idx + (d + 3) * (n8 + 2) + 1] += v_rco111;
    }
  float* raw = row + 6;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n8 + 2);
      hist[idx] += hist[idx + n8];
      hist[idx + 1] += hist[idx + n8 + 1];
      for (int b = 0; b < n8; ++b) raw[(i * d + j) * n8 + b] = hist[idx + b];
    }
  float nrm2 = 0;
  for (int b = 0; b < 128; ++b) nrm2 += raw[b] * raw[b];
  const float thr = sqrtf(nrm2) * 0.2f;
  nrm2 = 0;
  for (int b = 0; b < 128; ++b) {
    const float v = raw[b] < thr ? raw[b] : thr;
    raw[b] = v;
    nrm2 += v * v;
  }
  const float s = 512.f / fmaxf(sqrtf(nrm2), 1.1920929e-07f);
  for (int b = 0; b < 128; ++b) {
    const float v = rintf(raw[b] * s);
    raw[b] = v < 0 ? 0.f : (v > 255.f ? 255.f : v);
  }
}

__global__ void overflow_kernel(const unsigned* n, unsigned cap, unsigned* flag) {
  if (*n > cap) *flag = 1u;
}

taps_t make_taps(double sigma) {
  taps_t t;
  const int ks = (int)std::lrint(sigma * 8 + 1) | 1;
  t.r = ks / 2;
  double tmp[MAX_TAPS], sum = 0;
  for (int i = 0; i < ks; ++i) {
    const double d = i - t.r;
    tmp[i] = std::exp(-d * d / (2 * sigma * sigma));
    sum += tmp[i];
  }
  for (int i = 0; i < MAX_TAPS; ++i) t.w[i] = i < ks ? (float)(tmp[i] / sum) : 0.f;
  return t;
}

bool row_less(const float* a, const float* b) {
  if (a[0] != b[0]) return a[0] < b[0];
  if (a[1] != b[1]) return a[1] < b[1];
  if (a[2] != b[2]) return a[2] > b[2];
  if (a[3] != b[3]) return a[3] < b[3];
  if (a[4] != b[4]) return a[4] > b[4];
  if (a[5] != b[5]) return a[5] > b[5];
  return false;
}

}  // namespace

extern "C" {

// internal keypoint list capacity for an H x W image: grows with the image (the full 1376x1241 frame of the
// synthetic stream yields ~9.3k keypoints, one per ~180 pixels); also the row count a caller must provide
// when it asks for every keypoint (cap <= 0)
int vo_sift_capacity(int H, int W) {
  const long long px = (long long)H * W;
  const long long c = px / 32;
  return (int)(c < 65536 ? 65536 : (c > (1 << 20) ? (1 << 20) : c));
}

int vo_sift(vo_ctx* ctx, const uint8_t* img, int H, int W, int cap, float* kp_out, float* desc_out, int32_t* n_out) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && kp_out && desc_out && n_out, "sift: null pointer");
  VO_REQUIRE(ctx, H >= 16 && W >= 16, "sift: bad arguments");
  if (cap <= 0) cap = vo_sift_capacity(H, W);   // keep every keypoint, as cv2.SIFT_create() (nfeatures = 0) does
  *n_out = 0;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const float sigma = 1.6f, contrast_thr = 0.04f, edge_thr = 10.f;
  const int W0 = 2 * W, H0 = 2 * H;
  int n_oct = (int)std::lrint(std::log((double)std::min(W0, H0)) / std::log(2.0) - 2);
  n_oct = std::min(n_oct, MAX_OCT);
  {
    int w = W0, h = H0, k = 0;
    while (k < n_oct && w >= 2 * BORDER + 3 && h >= 2 * BORDER + 3) {
      ++k;
      w /= 2;
      h /= 2;
    }
    n_oct = k;
  }
  // one arena for the whole scale space: per octave NG Gaussian + NG-1 DoG images, plus one temp
  size_t total = (size_t)W0 * H0;   // temp
  {
    int w = W0, h = H0;
    for (int o = 0; o < n_oct; ++o) {
      total += (size_t)(2 * NG - 1) * w * h;
      w /= 2;
      h /= 2;
    }
  }
  const unsigned cap_kp = (unsigned)vo_sift_capacity(H, W), cap_cand = 4u * cap_kp;
  VO_TRY(vo_ensure(ctx, ctx->img, (size_t)H * W));
  VO_TRY(vo_ensure(ctx, ctx->sift_arena, total * 4));
  VO_TRY(vo_ensure(ctx, ctx->scratch[0], (size_t)cap_cand * 12));
  VO_TRY(vo_ensure(ctx, ctx->scratch[1], (size_t)cap_kp * sizeof(skp_t)));
  VO_TRY(vo_ensure(ctx, ctx->scratch[2], 64));
  VO_TRY(vo_ensure(ctx, ctx->scratch[3], (size_t)cap_kp * (134 + 360) * 4));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, (size_t)H * W, hipMemcpyHostToDevice, st));
  unsigned* d_cnt = (unsigned*)ctx->scratch[2].p;   // [0] candidates of the current octave, [1] keypoints
  VO_HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 64, st));

  float* arena = (float*)ctx->sift_arena.p;
  float* tmp = arena;
  float* cur = arena + (size_t)W0 * H0;
  oct_t oct[MAX_OCT];
  pyr_ptrs P;
  memset(&P, 0, sizeof(P));
  {
    int w = W0, h = H0;
    for (int o = 0; o < n_oct; ++o) {
      oct[o].H = h;
      oct[o].W = w;
      oct[o].o = o;
      P.H[o] = h;
      P.W[o] = w;
      for (int i = 0; i < NG; ++i) {
        oct[o].g[i] = cur;
        P.g[o][i] = cur;
        cur += (size_t)w * h;
      }
      for (int i = 0; i < NG - 1; ++i) {
        oct[o].d[i] = cur;
        cur += (size_t)w * h;
      }
      w /= 2;
      h /= 2;
    }
  }
  taps_t taps[NG];
  {
    const double kf = std::pow(2.0, 1.0 / NOL);
    taps[0] = make_taps(std::sqrt(std::max((double)sigma * sigma - 1.0, 0.01)));
    for (int i = 1; i < NG; ++i) {
      const double sp = std::pow(kf, i - 1) * sigma, s2 = sp * kf;
      taps[i] = make_taps(std::sqrt(s2 * s2 - sp * sp));
    }
  }
  auto grid2 = [](int w, int h) { return dim3(vo_cdiv(w, 64), vo_cdiv(h, 4)); };
  {
    vo_prof_scope ps(ctx, VO_K_SIFT_SCALESPACE);
    // base image: doubled, blurred to sigma
    float* up = const_cast<float*>(oct[0].g[1]);   // scratch until g[1] is produced
    hipLaunchKernelGGL(upsample2_kernel, grid2(W0, H0), dim3(256), 0, st, (const uint8_t*)ctx->img.p, H, W, up);
    hipLaunchKernelGGL(blur_kernel<true>, grid2(W0, H0), dim3(256), 0, st, up, H0, W0, taps[0], tmp);
    hipLaunchKernelGGL(blur_kernel<false>, grid2(W0, H0), dim3(256), 0, st, tmp, H0, W0, taps[0],
                       const_cast<float*>(oct[0].g[0]));
    for (int o = 0; o < n_oct; ++o) {
      const int w = oct[o].W, h = oct[o].H;
      if (o > 0)
        hipLaunchKernelGGL(decimate_kernel, grid2(w, h), dim3(256), 0, st, oct[o - 1].g[NOL], oct[o - 1].W, h, w,
                           const_cast<float*>(oct[o].g[0]));
      for (int i = 1; i < NG; ++i) {
        hipLaunchKernelGGL(blur_kernel<true>, grid2(w, h), dim3(256), 0, st, oct[o].g[i - 1], h, w, taps[i], tmp);
        hipLaunchKernelGGL(blur_kernel<false>, grid2(w, h), dim3(256), 0, st, tmp, h, w, taps[i],
                           const_cast<float*>(oct[o].g[i]));
      }
      const size_t n = (size_t)w * h;
      for (int i = 0; i < NG - 1; ++i)
        hipLaunchKernelGGL(dog_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, oct[o].g[i],
                           oct[o].g[i + 1], n, const_cast<float*>(oct[o].d[i]));
    }
  }
  VO_TRY(vo_check_launch(ctx, "sift scale space"));
  const float threshold = std::floor(0.5f * contrast_thr / NOL * 255.f);
  skp_t* d_kps = (skp_t*)ctx->scratch[1].p;
  for (int o = 0; o < n_oct; ++o) {
    VO_HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 4, st));
    {
      vo_prof_scope ps(ctx, VO_K_SIFT_DETECT);
      for (int layer = 1; layer <= NOL; ++layer)
        hipLaunchKernelGGL(extrema_kernel, grid2(oct[o].W, oct[o].H), dim3(256), 0, st, oct[o], layer, threshold,
                           (int*)ctx->scratch[0].p, d_cnt, cap_cand);
      // the candidate count stays on the device: launch for the capacity, surplus lanes exit
      hipLaunchKernelGGL(refine_orient_kernel, dim3(cap_cand / 64), dim3(64), 0, st, oct[o],
                         (const int*)ctx->scratch[0].p, d_cnt, cap_cand, contrast_thr, edge_thr, sigma, d_kps,
                         d_cnt + 1, cap_kp);
      hipLaunchKernelGGL(overflow_kernel, dim3(1), dim3(1), 0, st, d_cnt, cap_cand, d_cnt + 2);
    }
    VO_TRY(vo_check_launch(ctx, "sift detection"));
  }
  unsigned cnt[3] = {0, 0, 0};
  VO_HIP_TRY(ctx, hipMemcpyAsync(cnt, d_cnt, 12, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  if (cnt[2] || cnt[1] > cap_kp) return vo_set_error(ctx, VO_ECAPACITY, "sift: candidate / keypoint list overflow");
  const unsigned n_all = cnt[1];
  if (n_all == 0) return VO_OK;
  float* d_rows = (float*)ctx->scratch[3].p;
  {
    vo_prof_scope ps(ctx, VO_K_SIFT_DESCRIBE);
    hipLaunchKernelGGL(descriptor_kernel, dim3(vo_cdiv((int)n_all, 64)), dim3(64), 0, st, P, d_kps, n_all, d_rows);
  }
  VO_TRY(vo_check_launch(ctx, "sift descriptor_kernel"));
  std::vector<float> rows((size_t)n_all * 134);
  VO_HIP_TRY(ctx, hipMemcpyAsync(rows.data(), d_rows, rows.size() * 4, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  // order, duplicates, optional cap (KeyPointsFilter::removeDuplicatedSorted / retainBest)
  std::vector<unsigned> order(n_all);
  for (unsigned i = 0; i < n_all; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
    const float* ra = &rows[(size_t)a * 134];
    const float* rb = &rows[(size_t)b * 134];
    if (row_less(ra, rb)) return true;
    if (row_less(rb, ra)) return false;
    return memcmp(ra + 6, rb + 6, 128 * 4) < 0;   // full tie: any fixed order (rows are then duplicates)
  });
  std::vector<unsigned> keep;
  keep.reserve(n_all);
  for (unsigned i = 0; i < n_all; ++i) {
    const float* r = &rows[(size_t)order[i] * 134];
    if (!keep.empty()) {
      const float* p = &rows[(size_t)keep.back() * 134];
      if (p[0] == r[0] && p[1] == r[1] && p[2] == r[2] && p[3] == r[3]) continue;
    }
    keep.push_back(order[i]);
  }
  if ((int)keep.size() > cap) {
    std::vector<float> resp(keep.size());
    for (size_t i = 0; i < keep.size(); ++i) resp[i] = rows[(size_t)keep[i] * 134 + 4];
    std::vector<float> srt = resp;
    std::nth_element(srt.begin(), srt.begin() + (cap - 1), srt.end(), std::greater<float>());
    const float thr = srt[cap - 1];
    int above = 0;
    for (float v : resp) above += v > thr;
    int ties = cap - above;
    std::vector<unsigned> kept;
    for (size_t i = 0; i < keep.size(); ++i)
      if (resp[i] > thr || (resp[i] == thr && ties-- > 0)) kept.push_back(keep[i]);
    keep.swap(kept);
  }
  for (size_t i = 0; i < keep.size(); ++i) {
    const float* r = &rows[(size_t)keep[i] * 134];
    memcpy(kp_out + i * 6, r, 24);
    memcpy(desc_out + i * 128, r + 6, 512);
  }
  *n_out = (int32_t)keep.size();
  return VO_OK;
}

}  // extern "C"
