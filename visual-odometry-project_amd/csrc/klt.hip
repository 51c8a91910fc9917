// Image pyramid + pyramidal Lucas-Kanade tracking for gfx950.
//
// Reference call site: src/vo/features/klt.py:233-249
//   cv2.calcOpticalFlowPyrLK(prev, next, prevPts, None, winSize=(17,17), maxLevel=2,
//                            criteria=(EPS|COUNT, 10, 0.03))          (klt.py:29-33)
// The arithmetic (Bouguet's pyramidal LK as OpenCV implements it) is restated in
// oracle/csrc/klt.c; this file is the device version of the same definition:
//   pyrDown   5-tap [1 4 6 4 1] separable, reflect-101, (sum + 128) >> 8
//   Scharr    (3,10,3) int16 derivatives, reflect-101 inside the image, 0 outside
//   bilinear  14-bit fixed-point weights, template kept at 5 extra bits
//   solve     2x2 normal equations in float32 (sums are exact integers converted
//             once), <= max_iter steps, |d|^2 <= eps^2 and ping-pong stops
// One wavefront tracks one keypoint through all levels: the 64 lanes share the
// window pixels, patch data lives in LDS, the window sums are reduced across the
// wave with shuffles, and every lane carries the (uniform) 2x2 solve.
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int MAX_WIN = 31;
constexpr int W_BITS = 14;

struct pyr_t {
  const uint8_t* prev[MAX_LEVELS];
  const uint8_t* next[MAX_LEVELS];
  int H[MAX_LEVELS], W[MAX_LEVELS];
  int n_levels;
};

__device__ __forceinline__ int reflect101(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = (c < 0) ? -c : 2 * (n - 1) - c;
  return c;
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                       uint8_t* __restrict__ dst, int Hd, int Wd) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wd || y >= Hd) return;
  int xs[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xs[i] = reflect101(2 * x + i - 2, W);
  int sum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, H) * W;
    const int r = row[xs[0]] + 4 * row[xs[1]] + 6 * row[xs[2]] + 4 * row[xs[3]] + row[xs[4]];
    const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
    sum += wj * r;
  }
  dst[(size_t)y * Wd + x] = (uint8_t)((sum + 128) >> 8);
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  w00 = (int)rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
  w01 = (int)rintf(a * (1.f - b) * (float)(1 << W_BITS));
  w10 = (int)rintf((1.f - a) * b * (float)(1 << W_BITS));
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

// stage the (n x n) block of image pixels whose top-left is (x0, y0) into LDS, reflect-101
__device__ __forceinline__ void stage_region(const uint8_t* __restrict__ img, int H, int W, int x0, int y0, int n,
                                             uint8_t* s, int lane) {
  for (int i = lane; i < n * n; i += 64) {
    const int ly = i / n, lx = i - ly * n;
    s[i] = img[(size_t)reflect101(y0 + ly, H) * W + reflect101(x0 + lx, W)];
  }
}

__global__ __launch_bounds__(64) void klt_track_kernel(pyr_t P, const float* __restrict__ prev_xy, int N, int win,
                                                       int max_iter, double eps2, float min_eig_thr,
                                                       float* __restrict__ next_xy, uint8_t* __restrict__ status,
                                                       float* __restrict__ err) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const int ww = win * win;
  const int n1 = win + 1, n3 = win + 3;
  // LDS: region bytes (n3*n3) | derivatives int (n1*n1, dx | dy << 16) | template shorts (3 * ww)
  uint8_t* s_reg = smem;
  int* s_der = reinterpret_cast<int*>(smem + ((n3 * n3 + 15) & ~15));
  short* s_tpl = reinterpret_cast<short*>(s_der + n1 * n1);

  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const float p0x = prev_xy[2 * i], p0y = prev_xy[2 * i + 1];
  bool ok = true;
  float e_out = 0.f;
  float nx = 0.f, ny = 0.f;

  for (int level = P.n_levels - 1; level >= 0; --level) {
    const uint8_t* I = P.prev[level];
    const uint8_t* J = P.next[level];
    const int H = P.H[level], W = P.W[level];
    const float sc = (float)(1. / (double)(1 << level));
    float px = p0x * sc, py = p0y * sc;
    float qx, qy;
    if (level == P.n_levels - 1) {
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
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -win || ipx >= W || ipy < -win || ipy >= H) {
      if (level == 0) {
        ok = false;
        e_out = 0.f;
      }
      continue;
    }
    int w00, w01, w10, w11;
    bilinear_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);

    // ---- template: image block, Scharr derivatives, interpolated patch ----
    __syncthreads();
    stage_region(I, H, W, ipx - 1, ipy - 1, n3, s_reg, lane);
    __syncthreads();
    for (int k = lane; k < n1 * n1; k += 64) {
      const int ly = k / n1, lx = k - ly * n1;
      const int gy = ipy + ly, gx = ipx + lx;
      int dx = 0, dy = 0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const uint8_t* r0 = s_reg + ly * n3 + lx;
        const uint8_t* r1 = r0 + n3;
        const uint8_t* r2 = r1 + n3;
        const int t0 = (r0[0] + r2[0]) * 3 + r1[0] * 10;
        const int t2 = (r0[2] + r2[2]) * 3 + r1[2] * 10;
        const int d0 = r2[0] - r0[0], d1 = r2[1] - r0[1], d2 = r2[2] - r0[2];
        dx = t2 - t0;
        dy = (d2 + d0) * 3 + d1 * 10;
      }
      s_der[k] = (dx & 0xffff) | (dy << 16);
    }
    __syncthreads();
    int a11 = 0, a12 = 0, a22 = 0;
    long long A11l = 0, A12l = 0, A22l = 0;
    for (int k = lane; k < ww; k += 64) {
      const int y = k / win, x = k - y * win;
      const uint8_t* r = s_reg + (y + 1) * n3 + (x + 1);
      const int ival = descale(r[0] * w00 + r[1] * w01 + r[n3] * w10 + r[n3 + 1] * w11, W_BITS - 5);
      const int* d = s_der + y * n1 + x;
      const int v00 = d[0], v01 = d[1], v10 = d[n1], v11 = d[n1 + 1];
      const int ix = descale((int)(short)(v00 & 0xffff) * w00 + (int)(short)(v01 & 0xffff) * w01 +
                                 (int)(short)(v10 & 0xffff) * w10 + (int)(short)(v11 & 0xffff) * w11,
                             W_BITS);
      const int iy = descale((v00 >> 16) * w00 + (v01 >> 16) * w01 + (v10 >> 16) * w10 + (v11 >> 16) * w11, W_BITS);
      s_tpl[3 * k] = (short)ival;
      s_tpl[3 * k + 1] = (short)ix;
      s_tpl[3 * k + 2] = (short)iy;
      A11l += (long long)ix * ix;
      A12l += (long long)ix * iy;
      A22l += (long long)iy * iy;
    }
    (void)a11; (void)a12; (void)a22;
    const float A11 = (float)wave_sum(A11l) * FLT_SCALE;
    const float A12 = (float)wave_sum(A12l) * FLT_SCALE;
    const float A22 = (float)wave_sum(A22l) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww);
    if (minEig < min_eig_thr || D < 1.1920929e-07f) {
      if (level == 0) ok = false;
      continue;
    }
    D = 1.f / D;
    qx -= half;
    qy -= half;
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < max_iter; ++j) {
      const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
      if (iqx < -win || iqx >= W || iqy < -win || iqy >= H) {
        if (level == 0) ok = false;
        break;
      }
      bilinear_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
      __syncthreads();
      stage_region(J, H, W, iqx, iqy, n1, s_reg, lane);
      __syncthreads();
      long long b1l = 0, b2l = 0;
      for (int k = lane; k < ww; k += 64) {
        const int y = k / win, x = k - y * win;
        const uint8_t* r = s_reg + y * n1 + x;
        const int jv = descale(r[0] * w00 + r[1] * w01 + r[n1] * w10 + r[n1 + 1] * w11, W_BITS - 5);
        const int diff = jv - s_tpl[3 * k];
        b1l += (long long)diff * s_tpl[3 * k + 1];
        b2l += (long long)diff * s_tpl[3 * k + 2];
      }
      const float b1 = (float)wave_sum(b1l) * FLT_SCALE;
      const float b2 = (float)wave_sum(b2l) * FLT_SCALE;
      const float ddx = (A12 * b2 - A22 * b1) * D;
      const float ddy = (A12 * b1 - A11 * b2) * D;
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
    if (ok && level == 0) {
      const float ex = nx - half, ey = ny - half;
      const int iex = (int)floorf(ex), iey = (int)floorf(ey);
      if (iex < -win || iex >= W || iey < -win || iey >= H) {
        ok = false;
        continue;
      }
      bilinear_weights(ex - (float)iex, ey - (float)iey, w00, w01, w10, w11);
      __syncthreads();
      stage_region(J, H, W, iex, iey, n1, s_reg, lane);
      __syncthreads();
      long long sl = 0;
      for (int k = lane; k < ww; k += 64) {
        const int y = k / win, x = k - y * win;
        const uint8_t* r = s_reg + y * n1 + x;
        const int jv = descale(r[0] * w00 + r[1] * w01 + r[n1] * w10 + r[n1 + 1] * w11, W_BITS - 5);
        const int diff = jv - s_tpl[3 * k];
        sl += diff < 0 ? -diff : diff;
      }
      e_out = (float)wave_sum(sl) / (float)(32 * ww);
    }
  }
  if (lane == 0) {
    next_xy[2 * i] = nx;
    next_xy[2 * i + 1] = ny;
    status[i] = ok ? 1 : 0;
    err[i] = e_out;
  }
}

size_t klt_lds_bytes(int win) {
  const int n1 = win + 1, n3 = win + 3;
  return (size_t)((n3 * n3 + 15) & ~15) + (size_t)n1 * n1 * 4 + (size_t)3 * win * win * 2;
}

}  // namespace

extern "C" {

int vo_klt_num_levels(int H, int W, int win, int max_level) {
  // levels OpenCV's pyramid builder keeps: stop when the next level is not larger than the window
  int levels = 1, h = H, w = W;
  for (int l = 0; l < max_level && levels < MAX_LEVELS; ++l) {
    h = (h + 1) / 2;
    w = (w + 1) / 2;
    if (w <= win || h <= win) break;
    ++levels;
  }
  return levels;
}

size_t vo_pyramid_bytes(int H, int W, int n_levels) {
  size_t total = 0;
  int h = H, w = W;
  for (int l = 1; l < n_levels; ++l) {
    h = (h + 1) / 2;
    w = (w + 1) / 2;
    total += ((size_t)h * w + 255) & ~size_t(255);
  }
  return total ? total : 256;
}

// d_pyr receives levels 1 .. n_levels-1 back to back (each rounded up to 256 bytes)
int vo_pyramid_build_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int n_levels, uint8_t* d_pyr) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_img && d_pyr, "pyramid_build: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && n_levels >= 1 && n_levels <= MAX_LEVELS, "pyramid_build: bad arguments");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint8_t* src = d_img;
  int h = H, w = W;
  uint8_t* dst = d_pyr;
  for (int l = 1; l < n_levels; ++l) {
    const int hd = (h + 1) / 2, wd = (w + 1) / 2;
    {
      vo_prof_scope ps(ctx, VO_K_PYR_DOWN);
      hipLaunchKernelGGL(pyr_down_kernel, dim3(vo_cdiv(wd, 64), vo_cdiv(hd, 4)), dim3(256), 0, ctx->stream, src, h, w,
                         dst, hd, wd);
    }
    VO_TRY(vo_check_launch(ctx, "pyr_down_kernel"));
    src = dst;
    dst += ((size_t)hd * wd + 255) & ~size_t(255);
    h = hd;
    w = wd;
  }
  return VO_OK;
}

int vo_klt_track_dev(vo_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_prev_pyr, const uint8_t* d_next,
                     const uint8_t* d_next_pyr, int H, int W, int n_levels, const float* d_prev_xy, int N, int win,
                     int max_iter, double eps, double min_eig, float* d_next_xy, uint8_t* d_status, float* d_err) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, N >= 0, "klt_track: bad N");
  if (N == 0) return VO_OK;
  VO_REQUIRE(ctx, d_prev && d_next && d_prev_xy && d_next_xy && d_status && d_err, "klt_track: null pointer");
  VO_REQUIRE(ctx, n_levels >= 1 && n_levels <= MAX_LEVELS, "klt_track: n_levels must be in 1..%d", MAX_LEVELS);
  VO_REQUIRE(ctx, n_levels == 1 || (d_prev_pyr && d_next_pyr), "klt_track: pyramid buffers missing");
  VO_REQUIRE(ctx, win >= 3 && win <= MAX_WIN, "klt_track: window must be in 3..%d", MAX_WIN);
  VO_REQUIRE(ctx, H > 0 && W > 0, "klt_track: bad image size");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  pyr_t P;
  memset(&P, 0, sizeof(P));
  P.n_levels = n_levels;
  P.prev[0] = d_prev;
  P.next[0] = d_next;
  P.H[0] = H;
  P.W[0] = W;
  size_t off = 0;
  for (int l = 1; l < n_levels; ++l) {
    P.H[l] = (P.H[l - 1] + 1) / 2;
    P.W[l] = (P.W[l - 1] + 1) / 2;
    P.prev[l] = d_prev_pyr + off;
    P.next[l] = d_next_pyr + off;
    off += ((size_t)P.H[l] * P.W[l] + 255) & ~size_t(255);
  }
  if (max_iter < 0) max_iter = 0;
  if (max_iter > 100) max_iter = 100;
  if (eps < 0) eps = 0;
  if (eps > 10) eps = 10;
  {
    vo_prof_scope ps(ctx, VO_K_KLT_TRACK);
    hipLaunchKernelGGL(klt_track_kernel, dim3(N), dim3(64), klt_lds_bytes(win), ctx->stream, P, d_prev_xy, N, win,
                       max_iter, eps * eps, (float)min_eig, d_next_xy, d_status, d_err);
  }
  return vo_check_launch(ctx, "klt_track_kernel");
}

int vo_klt_track(vo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int H, int W, const float* prev_xy, int N,
                 int win, int max_level, int max_iter, double eps, double min_eig, float* next_xy, uint8_t* status,
                 float* err) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, N >= 0, "klt_track: bad N");
  if (N == 0) return VO_OK;
  VO_REQUIRE(ctx, prev && next && prev_xy && next_xy && status && err, "klt_track: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && max_level >= 0, "klt_track: bad arguments");
  VO_REQUIRE(ctx, win >= 3 && win <= MAX_WIN, "klt_track: window must be in 3..%d", MAX_WIN);
  const int nl = vo_klt_num_levels(H, W, win, max_level);
  const size_t px = (size_t)H * W, pb = vo_pyramid_bytes(H, W, nl);
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, ctx->img, px));
  VO_TRY(vo_ensure(ctx, ctx->img2, px));
  VO_TRY(vo_ensure(ctx, s[8], pb));
  VO_TRY(vo_ensure(ctx, s[9], pb));
  VO_TRY(vo_ensure(ctx, s[10], (size_t)N * 8));
  VO_TRY(vo_ensure(ctx, s[11], (size_t)N * 8));
  VO_TRY(vo_ensure(ctx, s[12], (size_t)N));
  VO_TRY(vo_ensure(ctx, s[13], (size_t)N * 4));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, prev, px, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img2.p, next, px, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[10].p, prev_xy, (size_t)N * 8, hipMemcpyHostToDevice, st));
  VO_TRY(vo_pyramid_build_dev(ctx, (const uint8_t*)ctx->img.p, H, W, nl, (uint8_t*)s[8].p));
  VO_TRY(vo_pyramid_build_dev(ctx, (const uint8_t*)ctx->img2.p, H, W, nl, (uint8_t*)s[9].p));
  VO_TRY(vo_klt_track_dev(ctx, (const uint8_t*)ctx->img.p, (const uint8_t*)s[8].p, (const uint8_t*)ctx->img2.p,
                          (const uint8_t*)s[9].p, H, W, nl, (const float*)s[10].p, N, win, max_iter, eps, min_eig,
                          (float*)s[11].p, (uint8_t*)s[12].p, (float*)s[13].p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(next_xy, s[11].p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(status, s[12].p, (size_t)N, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(err, s[13].p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_pyr_down(vo_ctx* ctx, const uint8_t* img, int H, int W, uint8_t* out) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && out && H > 0 && W > 0, "pyr_down: bad arguments");
  const size_t px = (size_t)H * W;
  const int hd = (H + 1) / 2, wd = (W + 1) / 2;
  VO_TRY(vo_ensure(ctx, ctx->img, px));
  VO_TRY(vo_ensure(ctx, ctx->scratch[8], vo_pyramid_bytes(H, W, 2)));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, px, hipMemcpyHostToDevice, ctx->stream));
  VO_TRY(vo_pyramid_build_dev(ctx, (const uint8_t*)ctx->img.p, H, W, 2, (uint8_t*)ctx->scratch[8].p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch[8].p, (size_t)hd * wd, hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

}  // extern "C"
