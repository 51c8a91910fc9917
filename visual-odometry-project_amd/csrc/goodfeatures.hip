// Shi-Tomasi corner detection ("good features to track") for gfx950.
//
// Reference call site: src/vo/features/klt.py:98
//   cv2.goodFeaturesToTrack(img, mask=mask, maxCorners=500, qualityLevel=0.01,
//                           minDistance=8, blockSize=7)                  (klt.py:24-26)
// The definition (restated in oracle/csrc/goodfeatures.c): min-eigenvalue map of the
// block x block structure tensor of 3x3 Sobel gradients (reflect-101 borders, exact
// integer sums scaled once), quality threshold against the masked maximum, 3x3 local
// maxima, descending order, greedy minimum-distance selection.
// Device: the image-wide work (eigenvalue map, maximum, thresholded local maxima ->
// compact candidate list).  Host: ordering and the greedy distance filter over the few
// thousand candidates, with a cell grid as OpenCV does -- it runs only when the tracker
// (re)detects, klt.py:207-230.
#include <algorithm>
#include <cmath>

#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int GX = 64, GY = 16, GT = 256;

__device__ __forceinline__ int refl(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = c < 0 ? -c : 2 * (n - 1) - c;
  return c;
}

__device__ __forceinline__ unsigned float_key(float f) {   // monotone float -> unsigned
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float key_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(GT) void min_eig_kernel(const uint8_t* __restrict__ img, int H, int W, int block,
                                                     float s2, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ eig, unsigned* __restrict__ max_key) {
  extern __shared__ __align__(16) int s_g[];                 // gradient region, packed (gx | gy << 16)
  __shared__ unsigned s_max;
  const int r0 = block / 2;
  const int RW = GX + block - 1, RH = GY + block - 1;
  int* s_hxx = s_g + RW * RH;                                // horizontal sums: RH x GX
  int* s_hxy = s_hxx + RH * GX;
  int* s_hyy = s_hxy + RH * GX;
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * GX, y0 = blockIdx.y * GY;
  if (tid == 0) s_max = 0;
  for (int i = tid; i < RW * RH; i += GT) {
    const int ly = i / RW, lx = i - ly * RW;
    const int y = refl(y0 + ly - r0, H), x = refl(x0 + lx - r0, W);   // box border: reflect the product image
    const int ym = refl(y - 1, H), yp = refl(y + 1, H), xm = refl(x - 1, W), xp = refl(x + 1, W);
    const uint8_t* rm = img + (size_t)ym * W;
    const uint8_t* rc = img + (size_t)y * W;
    const uint8_t* rp = img + (size_t)yp * W;
    const int p00 = rm[xm], p01 = rm[x], p02 = rm[xp], p10 = rc[xm], p12 = rc[xp], p20 = rp[xm], p21 = rp[x],
              p22 = rp[xp];
    const int gx = (p02 - p00) + 2 * (p12 - p10) + (p22 - p20);
    const int gy = (p20 - p00) + 2 * (p21 - p01) + (p22 - p02);
    s_g[i] = (gx & 0xffff) | (gy << 16);
  }
  __syncthreads();
  for (int i = tid; i < RH * GX; i += GT) {
    const int ly = i / GX, lx = i - ly * GX;
    const int* g = s_g + ly * RW + lx;
    int sxx = 0, sxy = 0, syy = 0;
    for (int k = 0; k < block; ++k) {
      const int v = g[k];
      const int a = (int)(short)(v & 0xffff), b = v >> 16;
      sxx += a * a;
      sxy += a * b;
      syy += b * b;
    }
    s_hxx[i] = sxx;
    s_hxy[i] = sxy;
    s_hyy[i] = syy;
  }
  __syncthreads();
  const int lx = tid & (GX - 1);
  unsigned local = 0;
  for (int ly = tid / GX; ly < GY; ly += GT / GX) {
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    long long sxx = 0, sxy = 0, syy = 0;
    for (int k = 0; k < block; ++k) {
      const int j = (ly + k) * GX + lx;
      sxx += s_hxx[j];
      sxy += s_hxy[j];
      syy += s_hyy[j];
    }
    const float a = (float)sxx * s2 * 0.5f, b = (float)sxy * s2, c = (float)syy * s2 * 0.5f;
    const float e = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    eig[(size_t)y * W + x] = e;
    if (!mask || mask[(size_t)y * W + x]) local = max(local, float_key(e));
  }
  if (local) atomicMax(&s_max, local);
  __syncthreads();
  if (tid == 0 && s_max) atomicMax(max_key, s_max);
}

__global__ __launch_bounds__(GT) void corner_candidates_kernel(const float* __restrict__ eig, int H, int W,
                                                               const uint8_t* __restrict__ mask,
                                                               const unsigned* __restrict__ max_key, double quality,
                                                               float* __restrict__ val, int* __restrict__ idx,
                                                               unsigned* __restrict__ count, unsigned cap) {
  const int x = blockIdx.x * GX + (threadIdx.x & (GX - 1));
  const int y = blockIdx.y * (GT / GX) + threadIdx.x / GX;
  if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) return;
  const unsigned mk = *max_key;
  if (mk == 0) return;                                           // empty mask
  const float thr = (float)((double)key_float(mk) * quality);
  const float v = eig[(size_t)y * W + x];
  if (!(v > thr) || v == 0.f) return;
  if (mask && !mask[(size_t)y * W + x]) return;
  float m = 0.f;
#pragma unroll
  for (int j = -1; j <= 1; ++j)
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      float q = eig[(size_t)(y + j) * W + (x + i)];
      q = q > thr ? q : 0.f;
      m = q > m ? q : m;
    }
  if (v != m) return;
  const unsigned pos = atomicAdd(count, 1u);
  if (pos < cap) {
    val[pos] = v;
    idx[pos] = y * W + x;
  }
}

}  // namespace

extern "C" {

int vo_min_eigen_map(vo_ctx* ctx, const uint8_t* img, int H, int W, int block, float* eig) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && eig && H > 0 && W > 0, "min_eigen_map: bad arguments");
  VO_REQUIRE(ctx, block >= 1 && block <= 31, "min_eigen_map: blockSize must be in 1..31");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t px = (size_t)H * W;
  hipStream_t st = ctx->stream;
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, ctx->img, px));
  VO_TRY(vo_ensure(ctx, s[0], px * 4));
  VO_TRY(vo_ensure(ctx, s[1], 16));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, px, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(s[1].p, 0, 16, st));
  const double scale = 1.0 / (4.0 * block * 255.0);
  const int RW = GX + block - 1, RH = GY + block - 1;
  const size_t lds = ((size_t)RW * RH + (size_t)3 * RH * GX) * 4;
  hipLaunchKernelGGL(min_eig_kernel, dim3(vo_cdiv(W, GX), vo_cdiv(H, GY)), dim3(GT), lds, st, (const uint8_t*)ctx->img.p,
                     H, W, block, (float)(scale * scale), (const uint8_t*)nullptr, (float*)s[0].p, (unsigned*)s[1].p);
  VO_TRY(vo_check_launch(ctx, "min_eig_kernel"));
  VO_HIP_TRY(ctx, hipMemcpyAsync(eig, s[0].p, px * 4, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_good_features(vo_ctx* ctx, const uint8_t* img, int H, int W, const uint8_t* mask, int max_corners,
                     double quality, double min_dist, int block, float* xy, int32_t* n_out) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && xy && n_out && H > 0 && W > 0, "good_features: bad arguments");
  VO_REQUIRE(ctx, block >= 1 && block <= 31, "good_features: blockSize must be in 1..31");
  VO_REQUIRE(ctx, quality > 0 && min_dist >= 0, "good_features: bad quality / minDistance");
  *n_out = 0;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t px = (size_t)H * W;
  hipStream_t st = ctx->stream;
  vo_buf* s = ctx->scratch;
  const unsigned cap = (unsigned)((px + 3) / 4 + 64);             // 3x3 maxima: at most one per 2x2 block
  VO_TRY(vo_ensure(ctx, ctx->img, px));
  VO_TRY(vo_ensure(ctx, s[0], px * 4));
  VO_TRY(vo_ensure(ctx, s[1], 16));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)cap * 4));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)cap * 4));
  if (mask) VO_TRY(vo_ensure(ctx, ctx->img2, px));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, px, hipMemcpyHostToDevice, st));
  if (mask) VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img2.p, mask, px, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(s[1].p, 0, 16, st));
  const uint8_t* d_mask = mask ? (const uint8_t*)ctx->img2.p : nullptr;
  unsigned* d_ctl = (unsigned*)s[1].p;                            // [0] max key, [1] candidate count
  const double scale = 1.0 / (4.0 * block * 255.0);
  const int RW = GX + block - 1, RH = GY + block - 1;
  const size_t lds = ((size_t)RW * RH + (size_t)3 * RH * GX) * 4;
  hipLaunchKernelGGL(min_eig_kernel, dim3(vo_cdiv(W, GX), vo_cdiv(H, GY)), dim3(GT), lds, st, (const uint8_t*)ctx->img.p,
                     H, W, block, (float)(scale * scale), d_mask, (float*)s[0].p, d_ctl);
  VO_TRY(vo_check_launch(ctx, "min_eig_kernel"));
  hipLaunchKernelGGL(corner_candidates_kernel, dim3(vo_cdiv(W, GX), vo_cdiv(H, GT / GX)), dim3(GT), 0, st,
                     (const float*)s[0].p, H, W, d_mask, d_ctl, quality, (float*)s[2].p, (int*)s[3].p, d_ctl + 1, cap);
  VO_TRY(vo_check_launch(ctx, "corner_candidates_kernel"));
  unsigned ctl[2] = {0, 0};
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctl, d_ctl, 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  const unsigned nc = std::min(ctl[1], cap);
  if (nc == 0) return VO_OK;
  std::vector<float> val(nc);
  std::vector<int> idx(nc);
  VO_HIP_TRY(ctx, hipMemcpy(val.data(), s[2].p, (size_t)nc * 4, hipMemcpyDeviceToHost));
  VO_HIP_TRY(ctx, hipMemcpy(idx.data(), s[3].p, (size_t)nc * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned> order(nc);
  for (unsigned i = 0; i < nc; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
    if (val[a] != val[b]) return val[a] > val[b];
    return idx[a] > idx[b];                                       // ties: higher address first
  });
  // greedy minimum distance with a cell grid (cell side = minDistance)
  int n = 0;
  if (min_dist >= 1) {
    const int cell = std::max(1, (int)std::lround(min_dist));
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    std::vector<std::vector<int>> grid((size_t)gw * gh);
    const double md2 = min_dist * min_dist;
    for (unsigned k = 0; k < nc && (max_corners <= 0 || n < max_corners); ++k) {
      const int id = idx[order[k]], y = id / W, x = id % W;
      const int cx = x / cell, cy = y / cell;
      bool ok = true;
      for (int yy = std::max(0, cy - 1); ok && yy <= std::min(gh - 1, cy + 1); ++yy)
        for (int xx = std::max(0, cx - 1); ok && xx <= std::min(gw - 1, cx + 1); ++xx)
          for (int j : grid[(size_t)yy * gw + xx]) {
            const double dx = x - xy[2 * j], dy = y - xy[2 * j + 1];
            if (dx * dx + dy * dy < md2) {
              ok = false;
              break;
            }
          }
      if (ok) {
        grid[(size_t)cy * gw + cx].push_back(n);
        xy[2 * n] = (float)x;
        xy[2 * n + 1] = (float)y;
        ++n;
      }
    }
  } else {
    for (unsigned k = 0; k < nc && (max_corners <= 0 || n < max_corners); ++k) {
      const int id = idx[order[k]];
      xy[2 * n] = (float)(id % W);
      xy[2 * n + 1] = (float)(id / W);
      ++n;
    }
  }
  *n_out = n;
  return VO_OK;
}

}  // extern "C"
