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

// Every level (level 0 included) is kept with PYR_PAD pixels of reflect-101 border on all
// sides, so the row tracker's blocks -- which reach at most WIN + 4 pixels past an edge, WIN <= 21
// -- are plain in-bounds reads.  prev/next point at the interior origin of each level.
constexpr int PYR_PAD = 32;

struct pyr_t {
  const uint8_t* prev[MAX_LEVELS];
  const uint8_t* next[MAX_LEVELS];
  int H[MAX_LEVELS], W[MAX_LEVELS], pitch[MAX_LEVELS];
  int n_levels;
};

__host__ __device__ inline int pyr_pitch(int W) { return (W + 2 * PYR_PAD + 3) & ~3; }
inline size_t pyr_level_bytes(int H, int W) {
  return ((size_t)(H + 2 * PYR_PAD) * pyr_pitch(W) + 255) & ~size_t(255);
}

__device__ __forceinline__ int reflect101(int c, int n) {
  if (n == 1) return 0;
  while (c < 0 || c >= n) c = (c < 0) ? -c : 2 * (n - 1) - c;
  return c;
}

// One thread per pixel of the padded destination level: the border pixels take the value of
// the interior pixel they mirror.  src points at the interior origin of the level above.
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ src, int spitch, int H, int W,
                                                       uint8_t* __restrict__ dst, int dpitch, int Hd, int Wd,
                                                       size_t seq_stride) {
  src += (size_t)blockIdx.z * seq_stride;          // several sequences per launch: both levels live in the
  dst += (size_t)blockIdx.z * seq_stride;          // sequence's pyramid buffer
  const int xp = blockIdx.x * 64 + (threadIdx.x & 63);
  const int yp = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (xp >= Wd + 2 * PYR_PAD || yp >= Hd + 2 * PYR_PAD) return;
  const int x = reflect101(xp - PYR_PAD, Wd), y = reflect101(yp - PYR_PAD, Hd);
  int xs[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xs[i] = reflect101(2 * x + i - 2, W);
  int sum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, H) * spitch;
    const int r = row[xs[0]] + 4 * row[xs[1]] + 6 * row[xs[2]] + 4 * row[xs[3]] + row[xs[4]];
    const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
    sum += wj * r;
  }
  dst[(size_t)yp * dpitch + xp] = (uint8_t)((sum + 128) >> 8);
}

// level 0: the frame itself with its reflected border
__global__ __launch_bounds__(256) void pad_reflect_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                          uint8_t* __restrict__ dst, int dpitch, size_t img_stride,
                                                          size_t pyr_stride) {
  src += (size_t)blockIdx.z * img_stride;
  dst += (size_t)blockIdx.z * pyr_stride;
  const int xp = blockIdx.x * 64 + (threadIdx.x & 63);
  const int yp = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (xp >= W + 2 * PYR_PAD || yp >= H + 2 * PYR_PAD) return;
  dst[(size_t)yp * dpitch + xp] = src[(size_t)reflect101(yp - PYR_PAD, H) * W + reflect101(xp - PYR_PAD, W)];
}

// v = interior pixel (x, y) of a W x H level -> its place in the bordered buffer and every border
// place that mirrors onto it (one reflection: needs min(W, H) > PYR_PAD)
__device__ __forceinline__ void store_with_mirrors(uint8_t* __restrict__ dst, int pitch, int H, int W, int x, int y,
                                                   uint8_t v, bool skip_own = false) {
  int xs[3], ys[3];
  int nx = 1, ny = 1;
  xs[0] = x + PYR_PAD;
  ys[0] = y + PYR_PAD;
  if (x >= 1 && x <= PYR_PAD) xs[nx++] = PYR_PAD - x;
  if (x >= W - 1 - PYR_PAD && x <= W - 2) xs[nx++] = PYR_PAD + 2 * (W - 1) - x;
  if (y >= 1 && y <= PYR_PAD) ys[ny++] = PYR_PAD - y;
  if (y >= H - 1 - PYR_PAD && y <= H - 2) ys[ny++] = PYR_PAD + 2 * (H - 1) - y;
  for (int j = 0; j < ny; ++j)
    for (int i = (j == 0 && skip_own) ? 1 : 0; i < nx; ++i) dst[(size_t)ys[j] * pitch + xs[i]] = v;
}

// 5x5 [1 4 6 4 1]^2 tap at (2x, 2y) of an unbordered W x H image, reflect-101
__device__ __forceinline__ int pyr_tap(const uint8_t* __restrict__ src, int pitch, int H, int W, int x, int y) {
  int xs[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xs[i] = reflect101(2 * x + i - 2, W);
  int sum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, H) * pitch;
    const int r = row[xs[0]] + 4 * row[xs[1]] + 6 * row[xs[2]] + 4 * row[xs[3]] + row[xs[4]];
    const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
    sum += wj * r;
  }
  return (sum + 128) >> 8;
}

// Levels 0, 1 and 2 of the bordered pyramid in ONE launch (three dependent launches were
// 22 us of the step's critical path).  No workgroup waits for another:
//   * the first nB workgroups each produce a 16x16 tile of level 2 from a 35x35 patch of level 1
//     that they compute for themselves in LDS (straight from the frame);
//   * the others produce level 1 (one pixel per thread) and copy the 2x2 frame pixels under it
//     into the bordered level 0.
// Border pixels are written by the thread that owns the interior pixel they mirror.
__global__ __launch_bounds__(256) void pyramid3_kernel(const uint8_t* __restrict__ src, int H0, int W0,
                                                       uint8_t* __restrict__ d0, int p0, uint8_t* __restrict__ d1,
                                                       int p1, int H1, int W1, uint8_t* __restrict__ d2, int p2,
                                                       int H2, int W2, int nB, int tiles2_x, int blocks1_x,
                                                       size_t img_stride, size_t pyr_stride) {
  src += (size_t)blockIdx.y * img_stride;            // several sequences per launch: grid.y = sequence
  d0 += (size_t)blockIdx.y * pyr_stride;
  d1 += (size_t)blockIdx.y * pyr_stride;
  d2 += (size_t)blockIdx.y * pyr_stride;
  __shared__ uint8_t s_l0[73 * 76];   // frame patch under the level-1 patch
  __shared__ uint8_t s_l1[35 * 36];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nB) {
    const int tx = (int)blockIdx.x % tiles2_x, ty = (int)blockIdx.x / tiles2_x;
    const int x2a = tx * 16, y2a = ty * 16;
    // Level-1 patch entry (ly, lx) stands for level-1 coordinate (2 y2a - 2 + ly, 2 x2a - 2 + lx),
    // reflected into level 1 where that falls outside.  For every level-2 pixel this tile stores
    // the reflected coordinates stay inside [lo, hi], so the frame pixels under them are one
    // contiguous patch (reflected at the frame's own edges), staged once.
    const int ax = 2 * x2a - 2, ay = 2 * y2a - 2;
    const int lox = max(ax, 0), hix = min(ax + 34, W1 - 1);
    const int loy = max(ay, 0), hiy = min(ay + 34, H1 - 1);
    const int fx0 = 2 * lox - 2, fy0 = 2 * loy - 2;             // frame coordinate of s_l0[0][0]
    const int fw = 2 * (hix - lox) + 5, fh = 2 * (hiy - loy) + 5;
    {
      // all loads of the patch go out before the first byte is stored (one round trip, not 21)
      constexpr int PER = (73 * 73 + 255) / 256;
      uint8_t v[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * 256;
        const int ly = i / fw, lx = i - ly * fw;
        int gx = fx0 + lx, gy = fy0 + ly;                       // at most 2 pixels outside: one reflection
        gx = gx < 0 ? -gx : (gx >= W0 ? 2 * (W0 - 1) - gx : gx);
        gy = gy < 0 ? -gy : (gy >= H0 ? 2 * (H0 - 1) - gy : gy);
        gx = min(max(gx, 0), W0 - 1);
        gy = min(max(gy, 0), H0 - 1);                           // (entries past fw * fh: any valid address)
        v[k] = src[(size_t)gy * W0 + gx];
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * 256;
        const int ly = i / fw, lx = i - ly * fw;
        if (i < fw * fh) s_l0[ly * 76 + lx] = v[k];
      }
    }
    __syncthreads();
    for (int i = tid; i < 35 * 35; i += 256) {
      const int ly = i / 35, lx = i - ly * 35;
      int x1 = reflect101(ax + lx, W1), y1 = reflect101(ay + ly, H1);
      x1 = min(max(x1, lox), hix);                              // (only entries no stored pixel uses are clamped)
      y1 = min(max(y1, loy), hiy);
      const uint8_t* q = s_l0 + (2 * (y1 - loy)) * 76 + 2 * (x1 - lox);
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = q + j * 76;
        const int r = row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4];
        const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
        sum += wj * r;
      }
      s_l1[ly * 36 + lx] = (uint8_t)((sum + 128) >> 8);
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int x2 = x2a + lx, y2 = y2a + ly;
    if (x2 < W2 && y2 < H2) {
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = s_l1 + (2 * ly + j) * 36 + 2 * lx;
        const int r = row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4];
        const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
        sum += wj * r;
      }
      store_with_mirrors(d2, p2, H2, W2, x2, y2, (uint8_t)((sum + 128) >> 8));
    }
    return;
  }
  // level 1: a 32x8 tile per workgroup from a 67x19 frame patch staged once; the patch's centre
  // is also what goes into the bordered level 0
  const int blk = (int)blockIdx.x - nB;
  const int x1a = (blk % blocks1_x) * 32, y1a = (blk / blocks1_x) * 8;
  const int fx0 = 2 * x1a - 2, fy0 = 2 * y1a - 2;
  {
    constexpr int PER = (67 * 19 + 255) / 256;
    uint8_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * 256;
      const int ly = i / 67, lx = i - ly * 67;
      int gx = fx0 + lx, gy = fy0 + ly;
      gx = gx < 0 ? -gx : (gx >= W0 ? 2 * (W0 - 1) - gx : gx);     // at most 2 pixels outside (tiles that stick out
      gy = gy < 0 ? -gy : (gy >= H0 ? 2 * (H0 - 1) - gy : gy);     //  further only feed pixels that are not stored)
      gx = min(max(gx, 0), W0 - 1);
      gy = min(max(gy, 0), H0 - 1);
      v[k] = src[(size_t)gy * W0 + gx];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * 256;
      const int ly = i / 67, lx = i - ly * 67;
      if (i < 67 * 19) s_l0[ly * 76 + lx] = v[k];
    }
  }
  __syncthreads();
  const int lx = tid & 31, ly = tid >> 5;
  const int x1 = x1a + lx, y1 = y1a + ly;
  if (x1 >= W1 || y1 >= H1) return;
  {
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint8_t* row = s_l0 + (2 * ly + j) * 76 + 2 * lx;
      const int r = row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4];
      const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
      sum += wj * r;
    }
    store_with_mirrors(d1, p1, H1, W1, x1, y1, (uint8_t)((sum + 128) >> 8));
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int x0 = 2 * x1, y0 = 2 * y1 + dy;
    if (y0 >= H0) continue;
    const uint8_t v0 = s_l0[(2 * ly + 2 + dy) * 76 + 2 * lx + 2], v1 = s_l0[(2 * ly + 2 + dy) * 76 + 2 * lx + 3];
    uint8_t* own = d0 + (size_t)(y0 + PYR_PAD) * p0 + (x0 + PYR_PAD);       // even offset: one 16-bit store for the pair
    if (x0 + 1 < W0) *reinterpret_cast<unsigned short*>(own) = (unsigned short)(v0 | (v1 << 8));
    else *own = v0;
    store_with_mirrors(d0, p0, H0, W0, x0, y0, v0, true);                    // border copies only
    if (x0 + 1 < W0) store_with_mirrors(d0, p0, H0, W0, x0 + 1, y0, v1, true);
  }
}

// ---- the same three levels, word-wide: one workgroup per 64x32 tile of level 1 ----
// pyramid3_kernel moves single bytes (global loads, LDS, stores) and spends a workgroup per 16x16 tile of level
// 2 on a 35x35 level-1 patch of its own: 20 us per frame however many frames a launch holds.  Here a workgroup
//   A  stages the frame patch under its level-1 tile + the 2-pixel ring level 2 needs (144 x 75 bytes) as aligned
//      dwords, every load issued before the first LDS store; out-of-frame bytes are reflected while staging;
//   B  copies the patch's centre into the bordered level 0 (dword stores);
//   C  makes the 68x36 level-1 region, two neighbours per work item from three dwords per patch row, into LDS;
//   D  stores its 64x32 centre to level 1 (dwords) and E makes the 32x16 tile of level 2 from the region.
// Border pixels are written by the owner of the interior pixel they mirror (edge tiles only).
// Needs W % 4 == 0 and a 4-byte aligned frame (the host picks the byte kernel otherwise).
// Tile size: 64x32 when a launch holds several frames (less halo work per pixel), 32x16 for a single frame (four
// times the workgroups, each a third of the latency: the launch is on the tracker's critical path then).

// seven bytes starting at byte 2 of dword a: two neighbouring 5-tap row sums
__device__ __forceinline__ void tap_pair(unsigned a, unsigned b, unsigned c, int& ra, int& rb) {
  // (1, 4, 6, 4) . four bytes is one v_dot4_u32_u8, the fifth tap its addend: five instructions for the pair
  ra = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 2), 0x04060401u, (b >> 16) & 255u, false);
  rb = (int)__builtin_amdgcn_udot4(b, 0x04060401u, c & 255u, false);
}

__device__ __forceinline__ int reflect_once(int c, int n) {
  c = c < 0 ? -c : (c >= n ? 2 * (n - 1) - c : c);
  return min(max(c, 0), n - 1);
}

template <int PT_X, int PT_Y>
__global__ __launch_bounds__(256) void pyramid3_tiled_kernel(const uint8_t* __restrict__ src, int H0, int W0,
                                                             uint8_t* __restrict__ d0, int p0, uint8_t* __restrict__ d1,
                                                             int p1, int H1, int W1, uint8_t* __restrict__ d2, int p2,
                                                             int H2, int W2, int tiles_x, size_t img_stride,
                                                             size_t pyr_stride) {
  constexpr int PT_RX = PT_X + 4, PT_RY = PT_Y + 4;   // level-1 region (ring of 2)
  constexpr int PT_FD = (2 * PT_X + 16) / 4;          // frame patch: dwords per row (virtual columns 2 x1a - 8 ..)
  constexpr int PT_FR = 2 * PT_Y + 11;                // frame patch rows (virtual rows 2 y1a - 6 ..)
  constexpr int PT_L1P = PT_RX + 4;                   // level-1 region pitch in bytes; entry lx sits at byte lx + 2
  static_assert(PT_X % 4 == 0 && PT_Y % 2 == 0, "tile: whole dwords per row, whole level-2 rows");
  src += (size_t)blockIdx.y * img_stride;
  d0 += (size_t)blockIdx.y * pyr_stride;
  d1 += (size_t)blockIdx.y * pyr_stride;
  d2 += (size_t)blockIdx.y * pyr_stride;
  __shared__ unsigned s_f[PT_FR * PT_FD];
  __shared__ __attribute__((aligned(4))) uint8_t s_l1[PT_RY * PT_L1P];
  const int tid = threadIdx.x;
  const int tile = (int)vo_xcd_tile(blockIdx.x, gridDim.x);
  const int x1a = (tile % tiles_x) * PT_X, y1a = (tile / tiles_x) * PT_Y;
  const int gx0 = 2 * x1a - 8, gy0 = 2 * y1a - 6;
  // ---- A: frame patch ----
  {
    constexpr int PER = (PT_FR * PT_FD + 255) / 256;
    unsigned v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * 256;
      const int r = i / PT_FD, d = i - r * PT_FD;
      const int gy = reflect_once(gy0 + r, H0);
      const int gx = gx0 + 4 * d;
      const uint8_t* row = src + (size_t)gy * W0;
      if (gx >= 0 && gx + 3 < W0) {
        v[k] = *reinterpret_cast<const unsigned*>(row + gx);
      } else {
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) w |= (unsigned)row[reflect_once(gx + b, W0)] << (8 * b);
        v[k] = w;
      }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * 256;
      if (i < PT_FR * PT_FD) s_f[i] = v[k];
    }
  }
  __syncthreads();
  // ---- B: level 0 = the patch's centre (rows 6 .., dwords 2 ..) ----
  {
    const bool edge0 = 2 * x1a <= PYR_PAD || 2 * x1a + 2 * PT_X >= W0 - 1 - PYR_PAD || 2 * y1a <= PYR_PAD ||
                       2 * y1a + 2 * PT_Y >= H0 - 1 - PYR_PAD;
    for (int i = tid; i < 2 * PT_Y * (2 * PT_X / 4); i += 256) {
      const int r = i / (2 * PT_X / 4), d = i - r * (2 * PT_X / 4);
      const int y0 = 2 * y1a + r, x0 = 2 * x1a + 4 * d;
      if (y0 >= H0 || x0 >= W0) continue;                       // (W0 % 4 == 0: a dword is inside or outside)
      const unsigned w = s_f[(r + 6) * PT_FD + d + 2];
      *reinterpret_cast<unsigned*>(d0 + (size_t)(y0 + PYR_PAD) * p0 + x0 + PYR_PAD) = w;
      if (edge0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) store_with_mirrors(d0, p0, H0, W0, x0 + b, y0, (uint8_t)(w >> (8 * b)), true);
      }
    }
  }
  // ---- C: level-1 region ----
  {
    const int lox = max(x1a - 2, 0), hix = min(x1a + PT_X + 1, W1 - 1);
    const int loy = max(y1a - 2, 0), hiy = min(y1a + PT_Y + 1, H1 - 1);
    const bool plain_x = x1a - 2 >= 0 && x1a + PT_X + 1 < W1;   // no reflection along x: neighbours are neighbours
    for (int i = tid; i < PT_RY * (PT_RX / 2); i += 256) {
      const int ly = i / (PT_RX / 2), j = i - ly * (PT_RX / 2);
      int yr = reflect101(y1a - 2 + ly, H1);
      yr = min(max(yr, loy), hiy);                              // (only entries no stored pixel uses are clamped)
      const unsigned* prow = s_f + (2 * (yr - y1a) + 4) * PT_FD;
      int sa = 0, sb = 0;
      if (plain_x) {
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          const unsigned* q = prow + jj * PT_FD + j;
          int ra, rb;
          tap_pair(q[0], q[1], q[2], ra, rb);
          const int wj = (jj == 0 || jj == 4) ? 1 : ((jj == 2) ? 6 : 4);
          sa += wj * ra;
          sb += wj * rb;
        }
      } else {
        const uint8_t* pb = reinterpret_cast<const uint8_t*>(prow);
        int xa = reflect101(x1a - 2 + 2 * j, W1), xb = reflect101(x1a - 1 + 2 * j, W1);
        xa = min(max(xa, lox), hix);
        xb = min(max(xb, lox), hix);
        const uint8_t* qa = pb + 2 * (xa - x1a) + 6;
        const uint8_t* qb = pb + 2 * (xb - x1a) + 6;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          const uint8_t* ra = qa + jj * PT_FD * 4;
          const uint8_t* rb = qb + jj * PT_FD * 4;
          const int wj = (jj == 0 || jj == 4) ? 1 : ((jj == 2) ? 6 : 4);
          sa += wj * (ra[0] + 4 * ra[1] + 6 * ra[2] + 4 * ra[3] + ra[4]);
          sb += wj * (rb[0] + 4 * rb[1] + 6 * rb[2] + 4 * rb[3] + rb[4]);
        }
      }
      const unsigned va = (unsigned)((sa + 128) >> 8), vb = (unsigned)((sb + 128) >> 8);
      *reinterpret_cast<unsigned short*>(s_l1 + ly * PT_L1P + 2 + 2 * j) = (unsigned short)(va | (vb << 8));
    }
  }
  __syncthreads();
  // ---- D: level 1 = the region's centre (entries 2 .., bytes 4 ..) ----
  {
    const bool edge1 = x1a <= PYR_PAD || x1a + PT_X >= W1 - 1 - PYR_PAD || y1a <= PYR_PAD || y1a + PT_Y >= H1 - 1 - PYR_PAD;
    for (int i = tid; i < PT_Y * (PT_X / 4); i += 256) {
      const int r = i / (PT_X / 4), d = i - r * (PT_X / 4);
      const int y1 = y1a + r, x1 = x1a + 4 * d;
      if (y1 >= H1 || x1 >= W1) continue;
      const unsigned w = *reinterpret_cast<const unsigned*>(s_l1 + (r + 2) * PT_L1P + 4 + 4 * d);
      uint8_t* own = d1 + (size_t)(y1 + PYR_PAD) * p1 + x1 + PYR_PAD;
      if (x1 + 3 < W1) {
        *reinterpret_cast<unsigned*>(own) = w;
      } else {
        for (int b = 0; x1 + b < W1; ++b) own[b] = (uint8_t)(w >> (8 * b));
      }
      if (edge1) {
        for (int b = 0; b < 4 && x1 + b < W1; ++b) store_with_mirrors(d1, p1, H1, W1, x1 + b, y1, (uint8_t)(w >> (8 * b)), true);
      }
    }
  }
  // ---- E: level 2 from the region: pixel (x1a / 2 + lx, y1a / 2 + ly) taps region entries 2 lx .. 2 lx + 4 ----
  for (int i = tid; i < (PT_Y / 2) * (PT_X / 4); i += 256) {
    const int ly = i / (PT_X / 4), j = i - ly * (PT_X / 4);
    const int x2 = x1a / 2 + 2 * j, y2 = y1a / 2 + ly;
    if (x2 < W2 && y2 < H2) {
      int sa = 0, sb = 0;
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const unsigned* q = reinterpret_cast<const unsigned*>(s_l1 + (2 * ly + jj) * PT_L1P) + j;
        int ra, rb;
        tap_pair(q[0], q[1], q[2], ra, rb);
        const int wj = (jj == 0 || jj == 4) ? 1 : ((jj == 2) ? 6 : 4);
        sa += wj * ra;
        sb += wj * rb;
      }
      store_with_mirrors(d2, p2, H2, W2, x2, y2, (uint8_t)((sa + 128) >> 8));
      if (x2 + 1 < W2) store_with_mirrors(d2, p2, H2, W2, x2 + 1, y2, (uint8_t)((sb + 128) >> 8));
    }
  }
}

// Sum over the 64 lanes of a wave with DPP row operations (no LDS traffic), result in
// every lane.  Steps: within quads, within half rows, within rows of 16, then the two
// row broadcasts that gfx9 provides for wave64.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

// exact 64-bit sum of per-lane 32-bit partials: two 16-bit limbs, each summed in 32 bits
__device__ __forceinline__ long long wave_sum(int v) {
  const int lo = wave_sum_i32(v & 0xffff);
  const int hi = wave_sum_i32(v >> 16);
  return (long long)hi * 65536LL + (long long)lo;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  w00 = (int)rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
  w01 = (int)rintf(a * (1.f - b) * (float)(1 << W_BITS));
  w10 = (int)rintf((1.f - a) * b * (float)(1 << W_BITS));
  w11 = (1 << W_BITS) - w00 - w01 - w10;
}

// stage the (n x n) block of image pixels whose top-left is (x0, y0) into LDS, reflect-101.
// Loads go out in batches of 8 per lane before any is written to LDS (one memory round
// trip per batch instead of one per byte).
__device__ __forceinline__ void stage_region(const uint8_t* __restrict__ img, int pitch, int H, int W, int x0, int y0,
                                             int n, uint8_t* s, int lane) {
  const bool inside = x0 >= 0 && y0 >= 0 && x0 + n <= W && y0 + n <= H;
  const int total = n * n;
  for (int b0 = 0; b0 < total; b0 += 8 * 64) {
    uint8_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = b0 + k * 64 + lane;
      const int ii = i < total ? i : total - 1;
      const int ly = ii / n, lx = ii - ly * n;
      const int gy = inside ? y0 + ly : reflect101(y0 + ly, H);
      const int gx = inside ? x0 + lx : reflect101(x0 + lx, W);
      v[k] = img[(size_t)gy * pitch + gx];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = b0 + k * 64 + lane;
      if (i < total) s[i] = v[k];
    }
  }
}

constexpr int KLT_MARGIN = 4;   // pixels of slack staged around the search window
constexpr int KLT_WAVES = 4;    // keypoints (waves) per workgroup

// Orders LDS traffic between the lanes of ONE wave (each wave owns a private LDS slice, and
// waves of a workgroup run different trip counts, so a workgroup barrier must not be used).
// The LDS executes a wave's instructions in order; only the compiler has to be held back.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// WIN_T > 0: window side known at compile time (index arithmetic folds, loops unroll)
template <int WIN_T>
__global__ __launch_bounds__(64 * KLT_WAVES) void klt_track_kernel(pyr_t P, const float* __restrict__ prev_xy, int N, const int* __restrict__ d_n, vo_klt_source src, vo_klt_batch B, int win_arg,
                                                       int max_iter, double eps2, float min_eig_thr,
                                                       float* __restrict__ next_xy, uint8_t* __restrict__ status,
                                                       float* __restrict__ err, int lds_per_wave) {
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int i = blockIdx.x * KLT_WAVES + (threadIdx.x >> 6);
  // (the kernel argument P is left untouched: written to, it would be copied to scratch -- one copy per lane -- for the
  //  run-time level index; measured as 14 MB of scratch traffic per launch)
  size_t pyr_off = 0;
  if (blockIdx.y != 0) {                               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    pyr_off = q * B.pyr;
    prev_xy += q * B.xy;
    next_xy += q * B.xy;
    status += q * B.out;
    err += q * B.out;
    if (src.n) {
      src.n = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(src.n) + q * B.ctl);
      src.num_features = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(src.num_features) + q * B.ctl);
      src.det_kp += q * B.det;
      if (src.ts) src.ts = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(src.ts) + q * B.ctl);
      if (src.det_go) src.det_go += q;
    }
  }
  if (d_n) N = min(N, *d_n);                           // keypoint count read on the device (frame pipeline)
  int n_own = N;                                       // points 0 .. n_own-1 are prev_xy's, the rest the detector's
  if (src.ts && blockIdx.x == 0 && threadIdx.x == 0) *src.ts = wall_clock64();
  if (src.n) {
    n_own = *src.n;
    const bool redetect = (double)n_own < (double)*src.num_features * src.frac && (!src.det_go || *src.det_go != 0);
    N = min(N, n_own + (redetect ? src.n_det : 0));
  }
  if (i >= N) return;                                  // whole wave leaves together
  unsigned char* smem = smem_all + (size_t)(threadIdx.x >> 6) * lds_per_wave;
  const int lane = threadIdx.x & 63;
  const int win = WIN_T > 0 ? WIN_T : win_arg;
  const int ww = win * win;
  const int n1 = win + 1, n3 = win + 3;
  const int RS = n1 + 2 * KLT_MARGIN;             // side of the staged search region
  // LDS: region bytes (max(n3*n3, RS*RS)) | derivatives int (n1*n1, dx | dy << 16) | template shorts (3 * ww)
  uint8_t* s_reg = smem;
  int* s_der = reinterpret_cast<int*>(smem + ((max(n3 * n3, RS * RS) + 15) & ~15));
  short4* s_tpl = reinterpret_cast<short4*>(s_der + n1 * n1 + ((n1 * n1) & 1));   // 8-byte aligned

  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const float p0x = i < n_own ? prev_xy[2 * i] : (float)src.det_kp[2 * (i - n_own)];
  const float p0y = i < n_own ? prev_xy[2 * i + 1] : (float)src.det_kp[2 * (i - n_own) + 1];
  bool ok = true;
  float e_out = 0.f;
  float nx = 0.f, ny = 0.f;

  for (int level = P.n_levels - 1; level >= 0; --level) {
    const uint8_t* I = P.prev[level] + pyr_off;
    const uint8_t* J = P.next[level] + pyr_off;
    const int H = P.H[level], W = P.W[level], pitch = P.pitch[level];
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
    wave_sync();
    stage_region(I, pitch, H, W, ipx - 1, ipy - 1, n3, s_reg, lane);
    wave_sync();
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
    wave_sync();
    int a11 = 0, a12 = 0, a22 = 0;   // per-lane partial sums stay below 2^31 for win <= 31
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
      s_tpl[k] = make_short4((short)ival, (short)ix, (short)iy, 0);
      a11 += ix * ix;
      a12 += ix * iy;
      a22 += iy * iy;
    }
    const float A11 = (float)wave_sum(a11) * FLT_SCALE;
    const float A12 = (float)wave_sum(a12) * FLT_SCALE;
    const float A22 = (float)wave_sum(a22) * FLT_SCALE;
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
    // search region of `next`: staged once with KLT_MARGIN pixels of slack, re-staged
    // only when the window walks out of it
    int rx0 = 0, ry0 = 0;
    bool staged = false;
    for (int j = 0; j < max_iter; ++j) {
      const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
      if (iqx < -win || iqx >= W || iqy < -win || iqy >= H) {
        if (level == 0) ok = false;
        break;
      }
      bilinear_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
      if (!staged || iqx < rx0 || iqy < ry0 || iqx + n1 > rx0 + RS || iqy + n1 > ry0 + RS) {
        rx0 = iqx - KLT_MARGIN;
        ry0 = iqy - KLT_MARGIN;
        wave_sync();
        stage_region(J, pitch, H, W, rx0, ry0, RS, s_reg, lane);
        wave_sync();
        staged = true;
      }
      const uint8_t* base = s_reg + (iqy - ry0) * RS + (iqx - rx0);
      int b1 = 0, b2 = 0;
#pragma unroll
      for (int k = lane; k < ww; k += 64) {
        const int y = k / win, x = k - y * win;
        const uint8_t* r = base + y * RS + x;
        const int jv = descale(r[0] * w00 + r[1] * w01 + r[RS] * w10 + r[RS + 1] * w11, W_BITS - 5);
        const short4 t = s_tpl[k];
        const int diff = jv - t.x;
        b1 += diff * t.y;
        b2 += diff * t.z;
      }
      const float fb1 = (float)wave_sum(b1) * FLT_SCALE;
      const float fb2 = (float)wave_sum(b2) * FLT_SCALE;
      const float ddx = (A12 * fb2 - A22 * fb1) * D;
      const float ddy = (A12 * fb1 - A11 * fb2) * D;
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
      if (!staged || iex < rx0 || iey < ry0 || iex + n1 > rx0 + RS || iey + n1 > ry0 + RS) {
        rx0 = iex - KLT_MARGIN;
        ry0 = iey - KLT_MARGIN;
        wave_sync();
        stage_region(J, pitch, H, W, rx0, ry0, RS, s_reg, lane);
        wave_sync();
        staged = true;
      }
      const uint8_t* base = s_reg + (iey - ry0) * RS + (iex - rx0);
      int sabs = 0;
      for (int k = lane; k < ww; k += 64) {
        const int y = k / win, x = k - y * win;
        const uint8_t* r = base + y * RS + x;
        const int jv = descale(r[0] * w00 + r[1] * w01 + r[RS] * w10 + r[RS + 1] * w11, W_BITS - 5);
        const int diff = jv - s_tpl[k].x;
        sabs += diff < 0 ? -diff : diff;
      }
      e_out = (float)wave_sum(sabs) / (float)(32 * ww);
    }
  }
  if (lane == 0) {
    next_xy[2 * i] = nx;
    next_xy[2 * i + 1] = ny;
    status[i] = ok ? 1 : 0;
    err[i] = e_out;
  }
}

// ---------------------------------------------------------------------------------
// Windows of 15, 17 and 21: one lane per window row -- 16 lanes (a DPP row, four keypoints per
// wave) for 15x15, 32 lanes (two keypoints per wave) for the larger ones
// ---------------------------------------------------------------------------------
// The kernel above spends most of its issue slots on work that is uniform per keypoint
// (window position, bilinear weights, the 2x2 solve, the reduction tree), replicated over the
// 64 lanes of a wave and paid once per keypoint.  Here a keypoint owns 16 lanes: lane r holds
// window row r -- the 16 source bytes of the row come from LDS in one read, the template row
// (patch value, Ix, Iy for 15 pixels) stays in registers for all iterations, the derivative
// row below is fetched from the neighbouring lane with a DPP row shift, and the window sums
// are four DPP adds inside the row.  The uniform work is now shared by four keypoints.
// Same arithmetic, in the same order per keypoint, as klt_track_kernel (integer sums are
// exact, so their order is free).
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef unsigned u32_any __attribute__((aligned(1)));
typedef unsigned short u16_any __attribute__((aligned(1)));

// geometry of the row kernel for a WIN x WIN window
template <int WIN>
struct klt_rows {
  static constexpr int n1 = WIN + 1, n3 = WIN + 3;          // derivative rows / rows of the template block
  static constexpr int RS = n1 + 2 * KLT_MARGIN;            // side of the staged search region
  static constexpr int PITCH = (RS + 7) & ~7;               // LDS bytes per staged row
  static constexpr int SLICE = PITCH * RS;                  // LDS bytes per keypoint
  static constexpr int WA = (n3 + 3) / 4, WB = (n1 + 3) / 4;   // words per template-block row / search row
  static_assert(WIN + 5 <= PYR_PAD, "the staged blocks must stay inside the pyramid's border");
};

__device__ __forceinline__ int row_sum_i32(int v) {   // sum over the 16 lanes of a row, in all of them
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
  return v;
}
// sum over the LPK (16 or 32) lanes of a keypoint, in all of them
template <int LPK>
__device__ __forceinline__ int kp_sum_i32(int v) {
  v = row_sum_i32(v);
  if (LPK == 32) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast15: rows 1, 3 += rows 0, 2
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    v = (threadIdx.x & 32) ? hi : lo;
  }
  return v;
}
// exact sum of per-lane 32-bit partials (two 16-bit limbs), as a double (|sum| < 2^53)
template <int LPK>
__device__ __forceinline__ double kp_sum_exact(int v) {
  const int lo = kp_sum_i32<LPK>(v & 0xffff);
  const int hi = kp_sum_i32<LPK>(v >> 16);
  return (double)hi * 65536.0 + (double)lo;
}

__device__ __forceinline__ int byte_at(const unsigned* w, int c) { return (int)((w[c >> 2] >> ((c & 3) * 8)) & 0xffu); }
// (byte c) | (byte c+1) << 16 of a little-endian word array
__device__ __forceinline__ unsigned pair_at(const unsigned* w, int c) {
  const int k = c >> 2, b = c & 3;
  const unsigned sel = (unsigned)b | 0x0c00u | ((unsigned)(b + 1) << 16) | 0x0c000000u;
  return __builtin_amdgcn_perm(b == 3 ? w[k + 1] : 0u, w[k], sel);   // byte index 4 = first byte of the next word
}
__device__ __forceinline__ int udot2(unsigned a, unsigned b, int c) {
  return (int)__builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b), (unsigned)c, false);
}
__device__ __forceinline__ int sdot2(unsigned a, unsigned b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2i16, a), __builtin_bit_cast(v2i16, b), c, false);
}

// NB (16, 18 or 22) bytes from a byte-aligned LDS address into words.  The pieces behind the first
// 16 bytes are volatile on purpose: left to itself the compiler pairs the tails of two rows into one
// ds_read2_b32, which ignores the low address bits and returns the wrong bytes for a misaligned base.
template <int NB>
__device__ __forceinline__ void load_row_bytes(const uint8_t* p, unsigned* w) {
  __builtin_memcpy(w, p, 16);
  if (NB >= 20) {
    w[4] = *reinterpret_cast<const volatile u32_any*>(p + 16);
    w[5] = NB >= 22 ? (unsigned)*reinterpret_cast<const volatile u16_any*>(p + 20) : 0u;
  } else {
    w[4] = NB >= 18 ? (unsigned)*reinterpret_cast<const volatile u16_any*>(p + 16) : 0u;
    w[5] = 0u;
  }
}

// rows [0, NR) x bytes [0, NC) of the image block whose top-left pixel is (x0, y0) -> s, pitch
// PITCH; lane r takes rows r, r + LPK, ...  The block may hang over the image by up to WIN + 4
// pixels: the level is stored with PYR_PAD pixels of reflected border.
// The two halves are separate so that a block can be requested long before it is needed (the template
// block of the next level while this level iterates, the search region while the template is built).
template <int NC, int NR, int LPK>
struct staged_block {
  static constexpr int PASSES = (NR + LPK - 1) / LPK, WORDS = (NC + 3) / 4;
  unsigned v[PASSES][WORDS];
};
template <int NC, int NR, int LPK>
__device__ __forceinline__ void stage_load(const uint8_t* __restrict__ img, int pitch, int x0, int y0, int r,
                                           staged_block<NC, NR, LPK>& b) {
#pragma unroll
  for (int pass = 0; pass < b.PASSES; ++pass) {
    const int row = r + LPK * pass;
    const bool on = row < NR;
    const uint8_t* src = img + (ptrdiff_t)(y0 + (on ? row : 0)) * pitch + x0;
#pragma unroll
    for (int k = 0; k < NC / 4; ++k) b.v[pass][k] = *(const u32_any*)(src + 4 * k);
    if (NC & 2) b.v[pass][NC / 4] = *(const u16_any*)(src + (NC & ~3));
  }
}
template <int NC, int NR, int LPK, int PITCH>
__device__ __forceinline__ void stage_store(const staged_block<NC, NR, LPK>& b, uint8_t* s, int r) {
#pragma unroll
  for (int pass = 0; pass < b.PASSES; ++pass) {
    const int row = r + LPK * pass;
    if (row < NR) {
      unsigned* d = reinterpret_cast<unsigned*>(s + row * PITCH);
#pragma unroll
      for (int k = 0; k < b.WORDS; ++k) d[k] = b.v[pass][k];
    }
  }
}
template <int NC, int NR, int LPK, int PITCH>
__device__ __forceinline__ void stage16(const uint8_t* __restrict__ img, int pitch, int x0, int y0, uint8_t* s, int r) {
  staged_block<NC, NR, LPK> b;
  stage_load<NC, NR, LPK>(img, pitch, x0, y0, r, b);
  stage_store<NC, NR, LPK, PITCH>(b, s, r);
}

template <int WIN, int LPK>
__device__ __forceinline__ void klt_track16_body(pyr_t P, const float* __restrict__ prev_xy, int N, const int* __restrict__ d_n, vo_klt_source src, vo_klt_batch B, int max_iter,
                                                 double eps2, float min_eig_thr, float* __restrict__ next_xy,
                                                 uint8_t* __restrict__ status, float* __restrict__ err) {
  typedef klt_rows<WIN> G;
  constexpr int win = WIN, ww = WIN * WIN, RS = G::RS, n1 = G::n1, n3 = G::n3, K16_PITCH = G::PITCH;
  constexpr int KPW = 64 / LPK;                        // keypoints per wave
  static_assert(n1 <= LPK, "one lane per derivative row");
  __shared__ __align__(16) uint8_t smem[KPW * G::SLICE + 16];
  const int lane = threadIdx.x;
  const int i = blockIdx.x * KPW + lane / LPK;
  // (the kernel argument P is left untouched: written to, it would be copied to scratch -- one copy per lane -- for the
  //  run-time level index; measured as 14 MB of scratch traffic per launch)
  size_t pyr_off = 0;
  if (blockIdx.y != 0) {                               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    pyr_off = q * B.pyr;
    prev_xy += q * B.xy;
    next_xy += q * B.xy;
    status += q * B.out;
    err += q * B.out;
    if (src.n) {
      src.n = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(src.n) + q * B.ctl);
      src.num_features = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(src.num_features) + q * B.ctl);
      src.det_kp += q * B.det;
      if (src.ts) src.ts = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(src.ts) + q * B.ctl);
      if (src.det_go) src.det_go += q;
    }
  }
  if (d_n) N = min(N, *d_n);                           // keypoint count read on the device (frame pipeline)
  int n_own = N;                                       // points 0 .. n_own-1 are prev_xy's, the rest the detector's
  if (src.ts && blockIdx.x == 0 && threadIdx.x == 0) *src.ts = wall_clock64();
  const bool byp = src.gate_mode == 2 && src.gate_wait != nullptr;   // the regroup's outputs / this kernel's: agent-scope accesses
  if (src.n) {
    n_own = byp ? vo_ld_agent(src.n) : *src.n;
    const bool redetect = (double)n_own < (double)*src.num_features * src.frac && (!src.det_go || *src.det_go != 0);
    N = min(N, n_own + (redetect ? src.n_det : 0));
  }
  if (i >= N) return;                                  // all lanes of a keypoint leave together
  const int r = lane & (LPK - 1);                      // window row of this lane (row WIN only feeds row WIN-1's derivatives)
  uint8_t* s_reg = smem + (lane / LPK) * G::SLICE;
  const int live = r < win ? 1 : 0;
  const int rr = r < win ? r : win - 1;                // row whose pixels this lane reads in the search loop
  const int rd = r < n1 ? r : n1 - 1;                  // derivative row of this lane (lanes past it idle along)

  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  float p0x, p0y;
  if (i >= n_own) {
    p0x = (float)src.det_kp[2 * (i - n_own)];
    p0y = (float)src.det_kp[2 * (i - n_own) + 1];
  } else if (byp) {
    p0x = vo_ld_agent(&prev_xy[2 * i]);
    p0y = vo_ld_agent(&prev_xy[2 * i + 1]);
  } else {
    p0x = prev_xy[2 * i];
    p0y = prev_xy[2 * i + 1];
  }
  bool ok = true;
  float e_out = 0.f;
  float nx = 0.f, ny = 0.f;

  // Template blocks depend on the previous keypoint only, so the block of level l - 1 is requested
  // while level l is being worked on (tplI holds the block of the level about to start).
  staged_block<n3, n3, LPK> tplI;
  auto request_template = [&](int level) {
    const float sc = (float)(1. / (double)(1 << level));
    const float px = p0x * sc - half, py = p0y * sc - half;
    int ipx = (int)floorf(px), ipy = (int)floorf(py);
    ipx = min(max(ipx, -win), P.W[level] - 1);       // (a level whose block would leave the border is skipped below)
    ipy = min(max(ipy, -win), P.H[level] - 1);
    stage_load<n3, n3, LPK>(P.prev[level] + pyr_off, P.pitch[level], ipx - 1, ipy - 1, r, tplI);
  };
  request_template(P.n_levels - 1);

  for (int level = P.n_levels - 1; level >= 0; --level) {
    const uint8_t* J = P.next[level] + pyr_off;
    const int H = P.H[level], W = P.W[level], pitch = P.pitch[level];
    const float sc = (float)(1. / (double)(1 << level));
    const staged_block<n3, n3, LPK> curI = tplI;
    if (level > 0) request_template(level - 1);
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
    // (w11 = 2^14 - w00 - w01 - w10 can come out as -1: the packed weights are signed 16-bit halves)
    unsigned wa = ((unsigned)w00 & 0xffffu) | ((unsigned)w01 << 16), wb = ((unsigned)w10 & 0xffffu) | ((unsigned)w11 << 16);

    // the search region around the starting guess is requested now and lands while the template is built
    qx -= half;
    qy -= half;
    int rx0 = (int)floorf(qx) - KLT_MARGIN, ry0 = (int)floorf(qy) - KLT_MARGIN;
    const bool first_in = rx0 + KLT_MARGIN >= -win && rx0 + KLT_MARGIN < W && ry0 + KLT_MARGIN >= -win && ry0 + KLT_MARGIN < H;
    staged_block<RS, RS, LPK> regJ;
    stage_load<RS, RS, LPK>(J, pitch, first_in ? rx0 : 0, first_in ? ry0 : 0, r, regJ);

    // ---- template: image block, Scharr derivatives, interpolated patch (all in registers) ----
    wave_sync();
    stage_store<n3, n3, LPK, K16_PITCH>(curI, s_reg, r);
    wave_sync();
    int tI[win], tX[win], tY[win];
    int a11 = 0, a12 = 0, a22 = 0;   // per-lane partial sums stay below 2^31
    {
      unsigned A0[G::WA], A1[G::WA], A2[G::WA];   // block rows r, r+1, r+2 (n3 bytes each): the rows around derivative row r
      {
        const unsigned* q0 = reinterpret_cast<const unsigned*>(s_reg + rd * K16_PITCH);
#pragma unroll
        for (int k = 0; k < G::WA; ++k) {
          A0[k] = q0[k];
          A1[k] = q0[k + K16_PITCH / 4];
          A2[k] = q0[k + 2 * (K16_PITCH / 4)];
        }
      }
      int cs[n3], cd[n3];             // per column: 3 (a0 + a2) + 10 a1, a2 - a0
#pragma unroll
      for (int c = 0; c < n3; ++c) {
        const int b0 = byte_at(A0, c), b1 = byte_at(A1, c), b2 = byte_at(A2, c);
        cs[c] = (b0 + b2) * 3 + b1 * 10;
        cd[c] = b2 - b0;
      }
      const int gy = ipy + rd;
      const bool row_in = gy >= 0 && gy < H;
      unsigned der[n1], dern[n1];     // (dx & 0xffff) | dy << 16 of derivative rows r and r + 1
#pragma unroll
      for (int x = 0; x < n1; ++x) {
        const int gx = ipx + x;
        int dx = cs[x + 2] - cs[x];
        int dy = (cd[x + 2] + cd[x]) * 3 + cd[x + 1] * 10;
        if (!(row_in && gx >= 0 && gx < W)) {
          dx = 0;
          dy = 0;
        }
        der[x] = ((unsigned)dx & 0xffffu) | ((unsigned)dy << 16);
      }
#pragma unroll
      for (int x = 0; x < n1; ++x)   // lane r reads lane r + 1: row_shl:1 inside a DPP row, wave_shl:1 across the two rows of a keypoint
        dern[x] = LPK == 16 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)der[x], 0x101, 0xF, 0xF, true)
                            : (unsigned)__builtin_amdgcn_update_dpp(0, (int)der[x], 0x130, 0xF, 0xF, true);
#pragma unroll
      for (int x = 0; x < win; ++x) {
        const int ival = sdot2(pair_at(A1, x + 1), wa, sdot2(pair_at(A2, x + 1), wb, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        const unsigned dxa = __builtin_amdgcn_perm(der[x + 1], der[x], 0x05040100u);
        const unsigned dya = __builtin_amdgcn_perm(der[x + 1], der[x], 0x07060302u);
        const unsigned dxb = __builtin_amdgcn_perm(dern[x + 1], dern[x], 0x05040100u);
        const unsigned dyb = __builtin_amdgcn_perm(dern[x + 1], dern[x], 0x07060302u);
        const int ix = live * (sdot2(dxa, wa, sdot2(dxb, wb, 1 << (W_BITS - 1))) >> W_BITS);
        const int iy = live * (sdot2(dya, wa, sdot2(dyb, wb, 1 << (W_BITS - 1))) >> W_BITS);
        tI[x] = ival;
        tX[x] = ix;
        tY[x] = iy;
        a11 += ix * ix;
        a12 += ix * iy;
        a22 += iy * iy;
      }
    }
    const float A11 = (float)kp_sum_exact<LPK>(a11) * FLT_SCALE;
    const float A12 = (float)kp_sum_exact<LPK>(a12) * FLT_SCALE;
    const float A22 = (float)kp_sum_exact<LPK>(a22) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww);
    if (minEig < min_eig_thr || D < 1.1920929e-07f) {
      if (level == 0) ok = false;
      continue;
    }
    D = 1.f / D;
    float pdx = 0.f, pdy = 0.f;
    // search region of `next`: staged once with KLT_MARGIN pixels of slack (requested above), re-staged
    // only when the window walks out of it
    bool staged = false;
    if (first_in) {
      wave_sync();
      stage_store<RS, RS, LPK, K16_PITCH>(regJ, s_reg, r);
      wave_sync();
      staged = true;
    }
    for (int j = 0; j < max_iter; ++j) {
      const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
      if (iqx < -win || iqx >= W || iqy < -win || iqy >= H) {
        if (level == 0) ok = false;
        break;
      }
      bilinear_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
      wa = ((unsigned)w00 & 0xffffu) | ((unsigned)w01 << 16);
      wb = ((unsigned)w10 & 0xffffu) | ((unsigned)w11 << 16);
      if (!staged || iqx < rx0 || iqy < ry0 || iqx + n1 > rx0 + RS || iqy + n1 > ry0 + RS) {
        rx0 = iqx - KLT_MARGIN;
        ry0 = iqy - KLT_MARGIN;
        wave_sync();
        stage16<RS, RS, LPK, K16_PITCH>(J, pitch, rx0, ry0, s_reg, r);
        wave_sync();
        staged = true;
      }
      const uint8_t* base = s_reg + (iqy - ry0 + rr) * K16_PITCH + (iqx - rx0);
      unsigned B0[6], B1[6];
      load_row_bytes<n1>(base, B0);
      load_row_bytes<n1>(base + K16_PITCH, B1);
      int b1 = 0, b2 = 0;
#pragma unroll
      for (int x = 0; x < win; ++x) {
        const int jv = sdot2(pair_at(B0, x), wa, sdot2(pair_at(B1, x), wb, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        const int diff = jv - tI[x];
        b1 += diff * tX[x];
        b2 += diff * tY[x];
      }
      const float fb1 = (float)kp_sum_exact<LPK>(b1) * FLT_SCALE;
      const float fb2 = (float)kp_sum_exact<LPK>(b2) * FLT_SCALE;
      const float ddx = (A12 * fb2 - A22 * fb1) * D;
      const float ddy = (A12 * fb1 - A11 * fb2) * D;
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
      wa = ((unsigned)w00 & 0xffffu) | ((unsigned)w01 << 16);
      wb = ((unsigned)w10 & 0xffffu) | ((unsigned)w11 << 16);
      if (!staged || iex < rx0 || iey < ry0 || iex + n1 > rx0 + RS || iey + n1 > ry0 + RS) {
        rx0 = iex - KLT_MARGIN;
        ry0 = iey - KLT_MARGIN;
        wave_sync();
        stage16<RS, RS, LPK, K16_PITCH>(J, pitch, rx0, ry0, s_reg, r);
        wave_sync();
        staged = true;
      }
      const uint8_t* base = s_reg + (iey - ry0 + rr) * K16_PITCH + (iex - rx0);
      unsigned B0[6], B1[6];
      load_row_bytes<n1>(base, B0);
      load_row_bytes<n1>(base + K16_PITCH, B1);
      int sabs = 0;
#pragma unroll
      for (int x = 0; x < win; ++x) {
        const int jv = sdot2(pair_at(B0, x), wa, sdot2(pair_at(B1, x), wb, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        const int diff = jv - tI[x];
        sabs += diff < 0 ? -diff : diff;
      }
      e_out = (float)kp_sum_i32<LPK>(live * sabs) / (float)(32 * ww);
    }
  }
  if (r == 0) {
    if (byp) {
      vo_st_agent(&next_xy[2 * i], nx);
      vo_st_agent(&next_xy[2 * i + 1], ny);
      vo_st_agent(&status[i], (uint8_t)(ok ? 1 : 0));
      vo_st_agent(&err[i], e_out);
    } else {
      next_xy[2 * i] = nx;
      next_xy[2 * i + 1] = ny;
      status[i] = ok ? 1 : 0;
      err[i] = e_out;
    }
  }
}

// The kernel: (optionally) a device-side gate in front of the body -- the previous flight's regroup has published the
// features this launch tracks -- and an arrival behind it, so that this flight's regroup can poll for the tracker's end
// instead of waiting for a stream event.
template <int WIN, int LPK>
__global__ __launch_bounds__(64) void klt_track16_kernel(pyr_t P, const float* __restrict__ prev_xy, int N, const int* __restrict__ d_n, vo_klt_source src, vo_klt_batch B, int max_iter,
                                                         double eps2, float min_eig_thr, float* __restrict__ next_xy,
                                                         uint8_t* __restrict__ status, float* __restrict__ err) {
  const size_t coff = (size_t)blockIdx.y * B.ctl;
  if (src.gate_wait && src.gate_want) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(src.gate_wait) + coff);
    if (!vo_gate_wait(w, src.gate_want, src.gate_mode != 2) && threadIdx.x == 0 && src.gate_fault)
      atomicOr(reinterpret_cast<int*>(reinterpret_cast<char*>(src.gate_fault) + coff), (int)VO_FAULT_GATE_BIT);
  }
  klt_track16_body<WIN, LPK>(P, prev_xy, N, d_n, src, B, max_iter, eps2, min_eig_thr, next_xy, status, err);
  if (src.gate_set) {
    if (src.gate_mode == 2) vo_stores_done();          // (one wave per workgroup)
    else __threadfence();
    if (threadIdx.x == 0)
      vo_gate_arrive(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(src.gate_cnt) + coff), gridDim.x,
                     reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(src.gate_set) + coff), src.gate_set_to,
                     src.gate_mode != 2);
  }
}

size_t klt_lds_bytes(int win) {
  const int n1 = win + 1, n3 = win + 3, RS = n1 + 2 * KLT_MARGIN;
  const int reg = n3 * n3 > RS * RS ? n3 * n3 : RS * RS;
  return (size_t)((reg + 15) & ~15) + (size_t)(n1 * n1 + 1) * 4 + (size_t)win * win * 8;
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
  for (int l = 0; l < n_levels; ++l) {
    total += pyr_level_bytes(h, w);
    h = (h + 1) / 2;
    w = (w + 1) / 2;
  }
  return total ? total : 256;
}

// d_pyr receives levels 0 .. n_levels-1 back to back, each with its reflected border (see pyr_t)
int vo_pyramid_build_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int n_levels, uint8_t* d_pyr) {
  return vo_pyramid_build_batch_dev(ctx, d_img, 0, 1, H, W, n_levels, d_pyr, 0);
}

}  // extern "C"

// S frames (d_img + s * img_stride) -> S pyramids (d_pyr + s * pyr_stride), every launch once for all sequences
int vo_pyramid_build_batch_dev(vo_ctx* ctx, const uint8_t* d_img, size_t img_stride, int S, int H, int W, int n_levels,
                               uint8_t* d_pyr, size_t pyr_stride) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, S >= 1 && S <= 65535, "pyramid_build: bad sequence count");
  VO_REQUIRE(ctx, d_img && d_pyr, "pyramid_build: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && n_levels >= 1 && n_levels <= MAX_LEVELS, "pyramid_build: bad arguments");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int h1 = (H + 1) / 2, w1 = (W + 1) / 2, h2 = (h1 + 1) / 2, w2 = (w1 + 1) / 2;
  int first_plain = 1;              // first level the per-level kernel still has to make
  const uint8_t* src = d_img;       // interior origin and pitch of the level above it
  int spitch = W, h = H, w = W;
  uint8_t* dst = d_pyr + pyr_level_bytes(H, W);
  if (n_levels >= 3 && h2 > PYR_PAD && w2 > PYR_PAD) {
    // levels 0..2 in one launch
    uint8_t* d1 = dst;
    uint8_t* d2 = d1 + pyr_level_bytes(h1, w1);
    if (W % 4 == 0 && ((uintptr_t)d_img & 3) == 0 && (img_stride & 3) == 0 && (pyr_stride & 3) == 0 && ((uintptr_t)d_pyr & 3) == 0) {
      vo_prof_scope ps(ctx, VO_K_PYR_DOWN);
      static const int forced = getenv("VO_PYR_TILE") ? atoi(getenv("VO_PYR_TILE")) : 0;     // tests: 64 / 32
      if (forced == 64 || (forced != 32 && (long)S * vo_cdiv(w1, 64) * vo_cdiv(h1, 32) >= 1024)) {
        const int tiles_x = vo_cdiv(w1, 64);
        hipLaunchKernelGGL((pyramid3_tiled_kernel<64, 32>), dim3(tiles_x * vo_cdiv(h1, 32), S), dim3(256), 0, ctx->stream,
                           d_img, H, W, d_pyr, pyr_pitch(W), d1, pyr_pitch(w1), h1, w1, d2, pyr_pitch(w2), h2, w2, tiles_x,
                           img_stride, pyr_stride);
      } else {
        const int tiles_x = vo_cdiv(w1, 32);
        hipLaunchKernelGGL((pyramid3_tiled_kernel<32, 16>), dim3(tiles_x * vo_cdiv(h1, 16), S), dim3(256), 0, ctx->stream,
                           d_img, H, W, d_pyr, pyr_pitch(W), d1, pyr_pitch(w1), h1, w1, d2, pyr_pitch(w2), h2, w2, tiles_x,
                           img_stride, pyr_stride);
      }
    } else {
      const int tiles2_x = vo_cdiv(w2, 16), nB = tiles2_x * vo_cdiv(h2, 16);
      const int blocks1_x = vo_cdiv(w1, 32), nA = blocks1_x * vo_cdiv(h1, 8);
      vo_prof_scope ps(ctx, VO_K_PYR_DOWN);
      hipLaunchKernelGGL(pyramid3_kernel, dim3(nB + nA, S), dim3(256), 0, ctx->stream, d_img, H, W, d_pyr, pyr_pitch(W), d1,
                         pyr_pitch(w1), h1, w1, d2, pyr_pitch(w2), h2, w2, nB, tiles2_x, blocks1_x, img_stride, pyr_stride);
    }
    VO_TRY(vo_check_launch(ctx, "pyramid3_kernel"));
    first_plain = 3;
    src = d2 + (size_t)PYR_PAD * pyr_pitch(w2) + PYR_PAD;
    spitch = pyr_pitch(w2);
    dst = d2 + pyr_level_bytes(h2, w2);
    h = h2;
    w = w2;
  } else {
    vo_prof_scope ps(ctx, VO_K_PYR_DOWN);
    hipLaunchKernelGGL(pad_reflect_kernel, dim3(vo_cdiv(W + 2 * PYR_PAD, 64), vo_cdiv(H + 2 * PYR_PAD, 4), S), dim3(256), 0,
                       ctx->stream, d_img, H, W, d_pyr, pyr_pitch(W), img_stride, pyr_stride);
    // (the next level reads the bordered copy of level 0, which lives in the sequence's pyramid buffer like every
    //  other level: same pixels as the frame, one stride for source and destination)
    src = d_pyr + (size_t)PYR_PAD * pyr_pitch(W) + PYR_PAD;
    spitch = pyr_pitch(W);
  }
  VO_TRY(vo_check_launch(ctx, "pad_reflect_kernel"));
  for (int l = first_plain; l < n_levels; ++l) {
    const int hd = (h + 1) / 2, wd = (w + 1) / 2;
    const int dpitch = pyr_pitch(wd);
    {
      vo_prof_scope ps(ctx, VO_K_PYR_DOWN);
      hipLaunchKernelGGL(pyr_down_kernel, dim3(vo_cdiv(wd + 2 * PYR_PAD, 64), vo_cdiv(hd + 2 * PYR_PAD, 4), S), dim3(256), 0,
                         ctx->stream, src, spitch, h, w, dst, dpitch, hd, wd, pyr_stride);
    }
    VO_TRY(vo_check_launch(ctx, "pyr_down_kernel"));
    src = dst + (size_t)PYR_PAD * dpitch + PYR_PAD;
    spitch = dpitch;
    dst += pyr_level_bytes(hd, wd);
    h = hd;
    w = wd;
  }
  return VO_OK;
}

extern "C" {

int vo_klt_track_dev(vo_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_prev_pyr, const uint8_t* d_next,
                     const uint8_t* d_next_pyr, int H, int W, int n_levels, const float* d_prev_xy, int N, int win,
                     int max_iter, double eps, double min_eig, float* d_next_xy, uint8_t* d_status, float* d_err) {
  return vo_klt_track_ndev(ctx, d_prev, d_prev_pyr, d_next, d_next_pyr, H, W, n_levels, d_prev_xy, N, nullptr, win,
                           max_iter, eps, min_eig, d_next_xy, d_status, d_err);
}

}  // extern "C"

// Pipeline-internal form: at most N keypoints, the actual count is read from *d_n when the kernel runs.
int vo_klt_track_ndev(vo_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_prev_pyr, const uint8_t* d_next,
                      const uint8_t* d_next_pyr, int H, int W, int n_levels, const float* d_prev_xy, int N,
                      const int32_t* d_n, int win, int max_iter, double eps, double min_eig, float* d_next_xy,
                      uint8_t* d_status, float* d_err, const vo_klt_source* src_in, const vo_klt_batch* batch) {
  if (!ctx) return VO_EINVAL;
  const vo_klt_source src = src_in ? *src_in : vo_klt_source();
  const vo_klt_batch B = batch ? *batch : vo_klt_batch();
  const int S = B.S > 0 ? B.S : 1;
  VO_REQUIRE(ctx, N >= 0, "klt_track: bad N");
  if (N == 0) return VO_OK;
  VO_REQUIRE(ctx, d_prev && d_next && d_prev_xy && d_next_xy && d_status && d_err, "klt_track: null pointer");
  VO_REQUIRE(ctx, n_levels >= 1 && n_levels <= MAX_LEVELS, "klt_track: n_levels must be in 1..%d", MAX_LEVELS);
  VO_REQUIRE(ctx, d_prev_pyr && d_next_pyr, "klt_track: pyramid buffers missing");
  VO_REQUIRE(ctx, win >= 3 && win <= MAX_WIN, "klt_track: window must be in 3..%d", MAX_WIN);
  VO_REQUIRE(ctx, H > 0 && W > 0, "klt_track: bad image size");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  pyr_t P;
  memset(&P, 0, sizeof(P));
  P.n_levels = n_levels;
  {
    // every level, level 0 included, is read from the bordered copies vo_pyramid_build_dev made
    size_t off = 0;
    int h = H, w = W;
    for (int l = 0; l < n_levels; ++l) {
      P.H[l] = h;
      P.W[l] = w;
      P.pitch[l] = pyr_pitch(w);
      const size_t origin = off + (size_t)PYR_PAD * P.pitch[l] + PYR_PAD;
      P.prev[l] = d_prev_pyr + origin;
      P.next[l] = d_next_pyr + origin;
      off += pyr_level_bytes(h, w);
      h = (h + 1) / 2;
      w = (w + 1) / 2;
    }
  }
  if (max_iter < 0) max_iter = 0;
  if (max_iter > 100) max_iter = 100;
  if (eps < 0) eps = 0;
  if (eps > 10) eps = 10;
  {
    vo_prof_scope ps(ctx, VO_K_KLT_TRACK);
    const int lds_wave = (int)((klt_lds_bytes(win) + 15) & ~size_t(15));
    const size_t lds = (size_t)lds_wave * KLT_WAVES;
    hipStream_t st = ctx->stream;
    const float me = (float)min_eig;
    const dim3 kgrid(vo_cdiv(N, KLT_WAVES), S), kblock(64 * KLT_WAVES);
    switch (win) {
      case 15: {
        // 16 lanes per keypoint (four per wave).  VO_KLT_LPK=32: two per wave, the upper half of each group idle in the row
        // loops, the 18 rows of the template block in one pass instead of two -- measured slower (round 3: step 87.6 ->
        // 91.3 us at one sequence, the kernel 199 -> 351 us at 16), kept as the measurement's knob.
        static const int lpk_env = getenv("VO_KLT_LPK") ? atoi(getenv("VO_KLT_LPK")) : 0;
        if (lpk_env == 32)
          vo_launch_stop(ctx, klt_track16_kernel<15, 32>, dim3(vo_cdiv(N, 2), S), dim3(64), 0, st, P, d_prev_xy, N, d_n, src, B,
                         max_iter, eps * eps, me, d_next_xy, d_status, d_err);
        else
          vo_launch_stop(ctx, klt_track16_kernel<15, 16>, dim3(vo_cdiv(N, 4), S), dim3(64), 0, st, P, d_prev_xy, N, d_n, src, B,
                         max_iter, eps * eps, me, d_next_xy, d_status, d_err);
        break;
      }
      case 17:   // the reference's default window (klt.py:29)
        vo_launch_stop(ctx, klt_track16_kernel<17, 32>, dim3(vo_cdiv(N, 2), S), dim3(64), 0, st, P, d_prev_xy, N, d_n, src, B,
                       max_iter, eps * eps, me, d_next_xy, d_status, d_err);
        break;
      case 21:
        vo_launch_stop(ctx, klt_track16_kernel<21, 32>, dim3(vo_cdiv(N, 2), S), dim3(64), 0, st, P, d_prev_xy, N, d_n, src, B,
                       max_iter, eps * eps, me, d_next_xy, d_status, d_err);
        break;
      default:
        vo_launch_stop(ctx, klt_track_kernel<0>, kgrid, kblock, lds, st, P, d_prev_xy, N, d_n, src, B, win, max_iter, eps * eps,
                       me, d_next_xy, d_status, d_err, lds_wave);
    }
  }
  return vo_check_launch(ctx, "klt_track_kernel");
}

extern "C" {

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
  {
    const uint8_t* l1 = (const uint8_t*)ctx->scratch[8].p + pyr_level_bytes(H, W);
    const int pitch = pyr_pitch(wd);
    VO_HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)wd, l1 + (size_t)PYR_PAD * pitch + PYR_PAD, (size_t)pitch, (size_t)wd,
                                     (size_t)hd, hipMemcpyDeviceToHost, ctx->stream));
  }
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

}  // extern "C"
