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

// Row pass and column pass of one Gaussian in one launch: an output tile, its input with `r` pixels of halo and the
// row-blurred intermediate live in LDS (the image-wide version reads every input 2 (2r + 1) times through the cache
// hierarchy and round-trips the intermediate through HBM).  Each sum runs in tap order without FMA, as above.
//
// Radius known at compile time (every radius cv2.SIFT_create()'s default sigmas produce): a 32 x 64 tile -- tall, the
// row pass also runs over the 2r halo rows --, and every work item makes several neighbouring outputs from one set of
// registers: four along x in the row pass (4 + 2r inputs in 16-byte LDS reads), eight down y in the column pass (8 + 2r
// reads) -- a sixth of the LDS reads of one output per work item.  Measured: no faster (octave 0: 21 us at r = 5, 42 us
// at r = 13, i.e. 54 MB at 1.3-2.6 TB/s; the launches of octaves >= 2 are 5-6 us whatever the kernel does).
constexpr int BT_W = 64, BT_H = 32;      // tile of the run-time-radius kernel
constexpr int BR_W = 32, BR_H = 64;      // tile of the compile-time-radius kernel
constexpr int BR_RB = 4, BR_CB = 8;      // outputs per work item: row pass, column pass
template <int R>
struct blur_geom {
  static constexpr int PW = BR_W + 2 * R;            // input columns of a tile
  static constexpr int PI = (PW + 3) & ~3;           // their pitch in LDS: rows start on 16 bytes
  static constexpr int RH = BR_H + 2 * R;            // input rows
  static constexpr int NV = 4 + ((2 * R + 3) & ~3);  // values a row-pass work item reads (whole float4s)
  static constexpr size_t lds = (size_t)RH * (PI + BR_W) * 4;
};
template <int R>
__global__ __launch_bounds__(256) void blur2d_rb_kernel(const float* __restrict__ src, int H, int W, taps_t k,
                                                        float* __restrict__ dst) {
  typedef blur_geom<R> G;
  extern __shared__ __align__(16) float blur_smem[];
  float* s_in = blur_smem;                  // RH x PI
  float* s_mid = blur_smem + G::RH * G::PI; // RH x BR_W
  const int tid = threadIdx.x;
  const unsigned tile = vo_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int x0 = (int)(tile % gridDim.x) * BR_W, y0 = (int)(tile / gridDim.x) * BR_H;
  // the tile with its halo, eight loads per work item in flight at a time
  {
    constexpr int total = G::RH * G::PW;
    const bool inside = x0 - R >= 0 && y0 - R >= 0 && x0 - R + G::PW <= W && y0 - R + G::RH <= H;
    const float* base_p = src + (ptrdiff_t)(y0 - R) * W + (x0 - R);
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(base + u * 256 + tid, total - 1);
        const int ky = i / G::PW, kx = i - ky * G::PW;
        v[u] = inside ? base_p[(ptrdiff_t)ky * W + kx] : src[(size_t)refl(y0 - R + ky, H) * W + refl(x0 - R + kx, W)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + tid;
        const int ky = i / G::PW, kx = i - ky * G::PW;
        if (i < total) s_in[ky * G::PI + kx] = v[u];
      }
    }
  }
  __syncthreads();
  // row pass: four neighbouring outputs per work item
  {
    constexpr int GR = BR_W / BR_RB;
    for (int it = tid; it < G::RH * GR; it += 256) {
      const int ky = it / GR, g = it - ky * GR;
      const float4* q = reinterpret_cast<const float4*>(s_in + ky * G::PI + BR_RB * g);
      float v[G::NV];
#pragma unroll
      for (int m = 0; m < G::NV / 4; ++m) {
        const float4 t = q[m];
        v[4 * m] = t.x;
        v[4 * m + 1] = t.y;
        v[4 * m + 2] = t.z;
        v[4 * m + 3] = t.w;
      }
      float o[BR_RB];
#pragma unroll
      for (int e = 0; e < BR_RB; ++e) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) a += k.w[j] * v[e + j];
        o[e] = a;
      }
      *reinterpret_cast<float4*>(s_mid + ky * BR_W + BR_RB * g) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  // column pass: eight outputs down a column per work item (256 work items = 32 columns x 8 strips)
  {
    static_assert(BR_W * (BR_H / BR_CB) == 256, "one column strip per work item");
    const int lx = tid & (BR_W - 1), ys = (tid / BR_W) * BR_CB;
    const float* q = s_mid + ys * BR_W + lx;
    float v[BR_CB + 2 * R];
#pragma unroll
    for (int j = 0; j < BR_CB + 2 * R; ++j) v[j] = q[j * BR_W];
    const int x = x0 + lx;
#pragma unroll
    for (int e = 0; e < BR_CB; ++e) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j <= 2 * R; ++j) a += k.w[j] * v[e + j];
      const int y = y0 + ys + e;
      if (x < W && y < H) dst[(size_t)y * W + x] = a;
    }
  }
}

// any radius (run-time trip counts: every tap waits for its own LDS read)
__global__ __launch_bounds__(256) void blur2d_kernel(const float* __restrict__ src, int H, int W, taps_t k,
                                                     float* __restrict__ dst) {
  extern __shared__ __align__(16) float blur_smem[];
  const int r = k.r, PI = BT_W + 2 * r, RH = BT_H + 2 * r;
  float* s_in = blur_smem;                 // RH x PI
  float* s_mid = blur_smem + RH * PI;      // RH x BT_W
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
  {
    const int total = RH * PI;
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(base + u * 256 + (int)threadIdx.x, total - 1);
        const int ky = i / PI, kx = i - ky * PI;
        v[u] = src[(size_t)refl(y0 - r + ky, H) * W + refl(x0 - r + kx, W)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        if (i < total) s_in[i] = v[u];
      }
    }
  }
  __syncthreads();
  for (int ky = ty; ky < RH; ky += 4) {
    const float* q = s_in + ky * PI + tx;
    float s = 0.f;
    for (int j = 0; j <= 2 * r; ++j) s += k.w[j] * q[j];
    s_mid[ky * BT_W + tx] = s;
  }
  __syncthreads();
  const int x = x0 + tx;
  for (int ly = ty; ly < BT_H; ly += 4) {
    const int y = y0 + ly;
    if (x >= W || y >= H) continue;
    const float* q = s_mid + ly * BT_W + tx;
    float s = 0.f;
    for (int j = 0; j <= 2 * r; ++j) s += k.w[j] * q[j * BT_W];
    dst[(size_t)y * W + x] = s;
  }
}

__global__ __launch_bounds__(256) void decimate_kernel(const float* __restrict__ src, int pw, int H, int W,
                                                       float* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < W && y < H) dst[(size_t)y * W + x] = src[(size_t)(2 * y) * pw + 2 * x];
}

// ---------------- detection ----------------
struct oct_t {
  const float* g[NG];
  int H, W, o;
};
// DoG layer l at pixel p.  The difference images are never stored: every reader takes the difference of the two
// Gaussian layers itself (one subtraction, the value a stored image would hold).
struct dog_t {
  const float *a, *b;
  __device__ __forceinline__ float operator[](size_t p) const { return b[p] - a[p]; }
};
__device__ __forceinline__ dog_t dog_layer(const oct_t& O, int l) { return dog_t{O.g[l], O.g[l + 1]}; }
struct octs_t {
  oct_t o[MAX_OCT];
};

// counters of one vo_sift call (device): [0] extrema candidates, [1] keypoints, [2] overflow flag,
// [3] refined survivors, [4] rows selected for description
enum { C_CAND = 0, C_KP = 1, C_OVER = 2, C_SURV = 3, C_SEL = 4 };
// The extrema are appended to EX_SUB lists, not one: a returning atomic on ONE counter completes at ~90 per microsecond
// chip-wide, and a textured frame makes ~10^4 wave-level reservations per octave -- with one list the append was 80 us of
// the 100 us an octave-1 launch took.  List j: entries [j * subcap, ...) of the candidate array, its counter C_SUB + 32 j
// (a 128-byte line of its own) in the counter block; a tile appends to list (tile mod EX_SUB).
constexpr int EX_SUB = 32, C_SUB = 16, C_WORDS = C_SUB + 32 * EX_SUB;

// 26-neighbour extrema of the three inner DoG layers of one octave in one pass over its six Gaussian layers.  A 64 x 16
// tile per workgroup: the five difference values of the tile and one ring of neighbours go to LDS (six coalesced loads
// per pixel, all in flight together), the tests then read LDS.  (With the neighbourhood read from memory, 54 dependent
// loads per centre that passes the threshold and most centres of a textured frame passing, the launch took 270 us on
// octave 0: 0.6 TB/s for 164 MB.)
constexpr int EX_W = 64, EX_H = 16, EX_PW = EX_W + 2, EX_PH = EX_H + 2;
__global__ __launch_bounds__(256) void extrema_kernel(oct_t O, float threshold, int4* __restrict__ cand,
                                                      unsigned* __restrict__ sub_cnt, unsigned subcap) {
  __shared__ float s_d[NG - 1][EX_PH][EX_PW];
  const int tid = threadIdx.x;
  const unsigned tile = vo_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int x0 = (int)(tile % gridDim.x) * EX_W, y0 = (int)(tile / gridDim.x) * EX_H;
  unsigned* n_cand = sub_cnt + 32u * (tile % (unsigned)EX_SUB);
  cand += (size_t)(tile % (unsigned)EX_SUB) * subcap;
  {
    constexpr int total = EX_PH * EX_PW, PER = (total + 255) / 256;
    float gv[PER][NG];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = min(tid + u * 256, total - 1);
      const int ly = i / EX_PW, lx = i - ly * EX_PW;
      const int gy = min(max(y0 - 1 + ly, 0), O.H - 1), gx = min(max(x0 - 1 + lx, 0), O.W - 1);   // (clamped entries are never a
      const size_t p = (size_t)gy * O.W + gx;                                                     //  neighbour of a tested centre)
#pragma unroll
      for (int l = 0; l < NG; ++l) gv[u][l] = O.g[l][p];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * 256;
      if (i < total) {
        const int ly = i / EX_PW, lx = i - ly * EX_PW;
#pragma unroll
        for (int l = 0; l < NG - 1; ++l) s_d[l][ly][lx] = gv[u][l + 1] - gv[u][l];
      }
    }
  }
  __syncthreads();
  // A centre is an extremum when no neighbour lies beyond it: for a positive centre the largest of the 27 values is the
  // centre itself, for a negative one the smallest.  Largest and smallest separably: over the three layers at every
  // position, over three columns, over three rows -- a work item owns a strip of four rows of one column, 90 LDS reads
  // and ~300 operations for its 12 (pixel, layer) tests (the plain 26-neighbour loop: 81 reads and ~200 operations per
  // test that passes the threshold, most of a textured frame -- the launch was bound by it, 120 us on octave 0).
  const int lx = tid & 63, ls = (tid >> 6) * (EX_H / 4);
  static_assert(EX_H / 4 == 4 && NOL == 3 && NG == 6, "strips of four rows, three tested layers out of five differences");
  float hmx[6][3], hmn[6][3], ctr[4][3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {                  // region row ls + j = tile row ls + j - 1
    float M[3][3], N[3][3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const float d0 = s_d[0][ls + j][lx + cc], d1 = s_d[1][ls + j][lx + cc], d2 = s_d[2][ls + j][lx + cc],
                  d3 = s_d[3][ls + j][lx + cc], d4 = s_d[4][ls + j][lx + cc];
      const float m12 = fmaxf(d1, d2), m23 = fmaxf(d2, d3), n12 = fminf(d1, d2), n23 = fminf(d2, d3);
      M[cc][0] = fmaxf(d0, m12);
      M[cc][1] = fmaxf(m12, d3);
      M[cc][2] = fmaxf(m23, d4);
      N[cc][0] = fminf(d0, n12);
      N[cc][1] = fminf(n12, d3);
      N[cc][2] = fminf(n23, d4);
      if (cc == 1 && j >= 1 && j <= 4) {
        ctr[j - 1][0] = d1;
        ctr[j - 1][1] = d2;
        ctr[j - 1][2] = d3;
      }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      hmx[j][l] = fmaxf(fmaxf(M[0][l], M[1][l]), M[2][l]);
      hmn[j][l] = fminf(fminf(N[0][l], N[1][l]), N[2][l]);
    }
  }
  const int c = x0 + lx;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = y0 + ls + k;
    const bool inner = c >= BORDER && c < O.W - BORDER && r >= BORDER && r < O.H - BORDER;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const float val = ctr[k][l];
      const float mx = fmaxf(fmaxf(hmx[k][l], hmx[k + 1][l]), hmx[k + 2][l]);
      const float mn = fminf(fminf(hmn[k][l], hmn[k + 1][l]), hmn[k + 2][l]);
      const bool ext = inner && fabsf(val) > threshold && (val > 0 ? val >= mx : val <= mn);
      // one reservation per wave
      const unsigned long long m = __ballot(ext);
      if (m != 0ull) {
        const int leader = __builtin_ctzll(m);
        unsigned base = 0;
        if ((tid & 63) == leader) base = atomicAdd(n_cand, (unsigned)__popcll(m));
        base = __shfl(base, leader);
        const unsigned pos = base + (unsigned)__popcll(m & ((1ull << (tid & 63)) - 1ull));
        if (ext && pos < subcap) cand[pos] = make_int4(O.o, l + 1, r, c);
      }
    }
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

// a refined extremum before its orientations are known
struct surv_t {
  skp_t kp;
  int r, c;
};

// Quadratic-fit refinement, contrast and edge tests: one lane per candidate of any octave (grid-stride).  Candidates
// that converge to the same (octave, layer, row, column) produce identical keypoints, which the reference removes
// after sorting (KeyPointsFilter::removeDuplicatedSorted): only the first to claim the cell in the hash table goes on.
__global__ __launch_bounds__(256) void refine_kernel(octs_t OS, const int4* __restrict__ cand,
                                                     const unsigned* __restrict__ sub_cnt, unsigned subcap,
                                                     float contrast_thr, float edge_thr, float sigma,
                                                     unsigned long long* __restrict__ table, unsigned table_mask,
                                                     surv_t* __restrict__ out, unsigned* __restrict__ n_out,
                                                     unsigned cap_out) {
  // candidate v of the EX_SUB lists taken one behind the other
  unsigned first[EX_SUB + 1];
  first[0] = 0;
#pragma unroll
  for (int j = 0; j < EX_SUB; ++j) first[j + 1] = first[j] + min(sub_cnt[32 * j], subcap);
  const unsigned n = first[EX_SUB];
  for (unsigned v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
    unsigned k = v;
#pragma unroll
    for (int j = 1; j < EX_SUB; ++j)
      if (v >= first[j]) k = (unsigned)j * subcap + (v - first[j]);
    const int4 cd = cand[k];
    const oct_t& O = OS.o[cd.x];
    int layer = cd.y, r = cd.z, c = cd.w;
    const int W = O.W, H = O.H;
    const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i;
    bool dead = false;
    for (i = 0; i < 5; ++i) {
      const dog_t img = dog_layer(O, layer), prv = dog_layer(O, layer - 1), nxt = dog_layer(O, layer + 1);
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
      if (!solve3(Hm, dD, X)) {
        dead = true;
        break;
      }
      xi = -X[2];
      xr = -X[1];
      xc = -X[0];
      if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
      if (fabsf(xi) > 7e8f || fabsf(xr) > 7e8f || fabsf(xc) > 7e8f) {
        dead = true;
        break;
      }
      c += (int)rintf(xc);
      r += (int)rintf(xr);
      layer += (int)rintf(xi);
      if (layer < 1 || layer > NOL || c < BORDER || c >= W - BORDER || r < BORDER || r >= H - BORDER) {
        dead = true;
        break;
      }
    }
    if (dead || i >= 5) continue;
    surv_t sv;
    {
      const dog_t img = dog_layer(O, layer), prv = dog_layer(O, layer - 1), nxt = dog_layer(O, layer + 1);
      const size_t p = (size_t)r * W + c;
      const float d0 = (img[p + 1] - img[p - 1]) * deriv_scale, d1 = (img[p + W] - img[p - W]) * deriv_scale,
                  d2 = (nxt[p] - prv[p]) * deriv_scale;
      const float t = d0 * xc + d1 * xr + d2 * xi;
      const float contr = img[p] * img_scale + t * 0.5f;
      if (fabsf(contr) * NOL < contrast_thr) continue;
      const float v2 = img[p] * 2.f;
      const float dxx = (img[p + 1] + img[p - 1] - v2) * second, dyy = (img[p + W] + img[p - W] - v2) * second;
      const float dxy = (img[p + W + 1] - img[p + W - 1] - img[p - W + 1] + img[p - W - 1]) * cross;
      const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
      if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) continue;
      const float po = (float)(1 << O.o);
      sv.kp.oct_x = c + xc;
      sv.kp.oct_y = r + xr;
      sv.kp.x = (c + xc) * po;
      sv.kp.y = (r + xr) * po;
      sv.kp.oct = O.o;
      sv.kp.layer = layer;
      sv.kp.size = sigma * sift_exp(((layer + xi) / NOL) * 0.6931471805599453f) * po * 2;
      sv.kp.response = fabsf(contr);
      sv.kp.angle = 0.f;
      sv.kp.octave = 0.f;
      sv.r = r;
      sv.c = c;
    }
    // claim (octave, layer, row, column): key never 0
    {
      const unsigned long long key = 1ull | ((unsigned long long)O.o << 1) | ((unsigned long long)layer << 5) |
                                     ((unsigned long long)r << 8) | ((unsigned long long)c << 32);
      unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & table_mask;
      bool mine = false;
      for (unsigned probe = 0; probe <= table_mask; ++probe) {
        const unsigned long long old = atomicCAS(&table[h], 0ull, key);
        if (old == 0ull) {
          mine = true;
          break;
        }
        if (old == key) break;
        h = (h + 1) & table_mask;
      }
      if (!mine) continue;
    }
    const unsigned pos = atomicAdd(n_out, 1u);
    if (pos < cap_out) out[pos] = sv;
  }
}

// The (2 R + 3)^2 pixels around (cy, cx) of an octave image -> LDS (zeros outside the image), all loads of a batch in
// flight together: the sample loops below then run out of LDS instead of paying a global round trip per 64 samples.
template <int BATCH>
__device__ __forceinline__ void stage_patch(const float* __restrict__ g, int H, int W, int cy, int cx, int R,
                                            float* __restrict__ s_patch, int lane) {
  const int PW = 2 * R + 3, n = PW * PW;
  const int y0 = cy - R - 1, x0 = cx - R - 1;
  for (int base = 0; base < n; base += 64 * BATCH) {
    float v[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int i = base + b * 64 + lane;
      const int yy = i / PW, xx = i - yy * PW;
      const int gy = y0 + yy, gx = x0 + xx;
      const bool in = i < n && gy >= 0 && gy < H && gx >= 0 && gx < W;
      v[b] = in ? g[(size_t)(in ? gy : 0) * W + (in ? gx : 0)] : 0.f;
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int i = base + b * 64 + lane;
      if (i < n) s_patch[i] = v[b];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
constexpr int ORI_RMAX = 17, DESC_RMAX = 39;   // radii cv2's default parameters reach: 16 and 38 (larger: global reads)

// Orientation histogram of every refined extremum: one wave per extremum.  The reference accumulates the 36 bins
// in pixel raster order; here 64 lanes evaluate 64 consecutive raster positions (gradient, exp, atan2: the
// expensive part), their (bin, weight) pairs are compacted in raster order into LDS, and lane b adds the pairs of
// bin b in that order -- every bin sees exactly the sequence of additions of the sequential loop.
// Four waves per workgroup, each with its own staging areas; the keypoints a workgroup makes are collected in LDS and
// appended to the list with ONE reservation (one returning atomic per keypoint on the one counter -- ~10^4 of them --
// bounds the launch at the counter's ~90 operations per microsecond).
constexpr int ORI_WAVES = 4, ORI_BUF = 96;
__global__ __launch_bounds__(64 * ORI_WAVES) void orient_kernel(octs_t OS, const surv_t* __restrict__ surv,
                                                                const unsigned* __restrict__ n_surv, unsigned cap_surv,
                                                                skp_t* __restrict__ out, unsigned* __restrict__ n_out,
                                                                unsigned cap_out) {
  __shared__ float s_add_w[ORI_WAVES][64];
  __shared__ float s_tmp_w[ORI_WAVES][36];
  __shared__ float s_patch_w[ORI_WAVES][(2 * ORI_RMAX + 3) * (2 * ORI_RMAX + 3)];
  __shared__ skp_t s_buf[ORI_BUF];
  __shared__ unsigned s_nbuf, s_base;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* s_add = s_add_w[wv];
  float* s_tmp = s_tmp_w[wv];
  float* s_patch = s_patch_w[wv];
  if (threadIdx.x == 0) s_nbuf = 0;
  __syncthreads();
  const unsigned n = min(*n_surv, cap_surv);
  for (unsigned k = blockIdx.x * ORI_WAVES + wv; k < n; k += gridDim.x * ORI_WAVES) {
    const surv_t sv = surv[k];
    const oct_t& O = OS.o[sv.kp.oct];
    const int W = O.W, H = O.H, r = sv.r, c = sv.c;
    const float scl_octv = sv.kp.size * 0.5f / (float)(1 << O.o);
    const int radius = (int)rintf(4.5f * scl_octv);
    const float osig = 1.5f * scl_octv;
    const float* g = O.g[sv.kp.layer];
    const float expf_scale = -1.f / (2.f * osig * osig);
    const int side = 2 * radius + 1, total = side * side;
    const bool staged = radius <= ORI_RMAX;
    const int PW = 2 * radius + 3;
    if (staged) stage_patch<8>(g, H, W, r, c, radius, s_patch, lane);
    float acc = 0.f;                                  // lane b < 36: bin b
    for (int base = 0; base < total; base += 64) {
      const int p = base + lane;
      bool on = false;
      int bin = 0;
      float add = 0.f;
      if (p < total) {
        const int ii = p / side - radius, jj = p - (p / side) * side - radius;
        const int y = r + ii, x = c + jj;
        if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {
          float dx, dy;
          if (staged) {
            const float* q = s_patch + (ii + radius + 1) * PW + (jj + radius + 1);
            dx = q[1] - q[-1];
            dy = q[-PW] - q[PW];
          } else {
            dx = g[(size_t)y * W + x + 1] - g[(size_t)y * W + x - 1];
            dy = g[(size_t)(y - 1) * W + x] - g[(size_t)(y + 1) * W + x];
          }
          const float w = sift_exp((float)(ii * ii + jj * jj) * expf_scale);
          const float ori = sift_atan2(dy, dx);
          const float mag = sqrtf(dx * dx + dy * dy);
          bin = (int)rintf(0.1f * ori);
          if (bin >= 36) bin -= 36;
          if (bin < 0) bin += 36;
          add = w * mag;
          on = true;
        }
      }
      // lane b adds the samples of bin b in lane order = raster order; the samples come through scalar registers
      // (v_readlane), no LDS round trip per sample
      if (!on) bin = -1;
      if (__ballot(on) != 0ull) {
#pragma unroll
        for (int e = 0; e < 64; ++e) {
          const int be = __builtin_amdgcn_readlane(bin, e);
          const float ae = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, add), e));
          if (be == lane) acc += ae;
        }
      }
    }
    if (lane < 36) s_tmp[lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float h = 0.f;
    if (lane < 36) {
      const int b = lane;
      h = (s_tmp[(b + 34) % 36] + s_tmp[(b + 2) % 36]) * (1.f / 16.f) +
          (s_tmp[(b + 35) % 36] + s_tmp[(b + 1) % 36]) * (4.f / 16.f) + s_tmp[b] * (6.f / 16.f);
    }
    float mx = h;                                      // (lanes >= 36 hold 0; the reference's maximum starts at 0)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float mag_thr = mx * 0.8f;
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) s_add[lane] = h;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
      const int j = lane;
      const int l = j > 0 ? j - 1 : 35, rr = j < 35 ? j + 1 : 0;
      const float hl = s_add[l], hr = s_add[rr];
      if (h > hl && h > hr && h >= mag_thr) {
        float bin = j + 0.5f * (hl - hr) / (hl - 2 * h + hr);
        bin = bin < 0 ? 36 + bin : (bin >= 36 ? bin - 36 : bin);
        float angle = 360.f - (360.f / 36) * bin;
        if (fabsf(angle - 360.f) < 1.1920929e-07f) angle = 0.f;
        skp_t q = sv.kp;
        q.angle = angle;
        const unsigned slot = atomicAdd(&s_nbuf, 1u);
        if (slot < (unsigned)ORI_BUF) {
          s_buf[slot] = q;
        } else {                                       // (the buffer is full: this one goes to the list directly)
          const unsigned pos = atomicAdd(n_out, 1u);
          if (pos < cap_out) out[pos] = q;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const unsigned nb = min(s_nbuf, (unsigned)ORI_BUF);
  if (threadIdx.x == 0 && nb) s_base = atomicAdd(n_out, nb);
  __syncthreads();
  if (threadIdx.x < nb) {
    const unsigned pos = s_base + threadIdx.x;
    if (pos < cap_out) out[pos] = s_buf[threadIdx.x];
  }
}

// retainBest (KeyPointsFilter::retainBest): the keypoints whose response is at least the cap-th largest go on to
// description (ties at the bound are settled on the host, in sorted order, as the reference settles them).  One
// workgroup: three radix passes over the response bits (positive floats order like their bit patterns).
__global__ __launch_bounds__(1024) void select_kernel(const skp_t* __restrict__ kps, const unsigned* __restrict__ n_kp,
                                                      unsigned cap_kp, unsigned cap, unsigned* __restrict__ sel,
                                                      unsigned* __restrict__ n_sel) {
  __shared__ unsigned s_hist[2048], s_scan[1024];
  __shared__ unsigned s_prefix, s_need, s_cnt;
  const int tid = threadIdx.x;
  const unsigned n = min(*n_kp, cap_kp);
  unsigned thr_bits = 0u;
  if (n > cap) {
    if (tid == 0) {
      s_prefix = 0u;
      s_need = cap;
    }
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix, need = s_need;
      const int sh = shifts[pass], wd = widths[pass];
      const unsigned hi_mask = pass == 0 ? 0u : ~0u << (sh + wd);
      for (unsigned i = tid; i < n; i += 1024) {
        const unsigned b = __float_as_uint(kps[i].response);
        if ((b & hi_mask) == prefix) atomicAdd(&s_hist[(b >> sh) & ((1u << wd) - 1u)], 1u);
      }
      __syncthreads();
      // the digit d with  sum(hist[d+1 ..]) < need <= sum(hist[d ..])  (0 when even the whole count falls short):
      // suffix sums of bin pairs, one pair per work item
      {
        const unsigned pair = s_hist[2 * tid] + s_hist[2 * tid + 1];
        s_scan[tid] = pair;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
          const unsigned add = tid + off < 1024 ? s_scan[tid + off] : 0u;
          __syncthreads();
          s_scan[tid] += add;
          __syncthreads();
        }
        const unsigned incl = s_scan[tid], above = incl - pair;          // bins >= 2 tid; bins >= 2 tid + 2
        if (above < need && need <= incl) {
          const unsigned hi = s_hist[2 * tid + 1];
          const int d = above + hi >= need ? 2 * tid + 1 : 2 * tid;
          s_prefix = prefix | ((unsigned)d << sh);
          s_need = need - (d == 2 * tid + 1 ? above : above + hi);
        }
        if (tid == 0 && incl < need) {
          s_prefix = prefix;
          s_need = need - incl;
        }
        __syncthreads();
      }
    }
    thr_bits = s_prefix;
  }
  if (tid == 0) s_cnt = 0u;
  __syncthreads();
  for (unsigned i = tid; i < n; i += 1024)
    if (__float_as_uint(kps[i].response) >= thr_bits) sel[atomicAdd(&s_cnt, 1u)] = i;
  __syncthreads();
  if (tid == 0) *n_sel = s_cnt;
}

// ---------------- description ----------------
struct pyr_ptrs {
  const float* g[MAX_OCT][NG];
  int H[MAX_OCT], W[MAX_OCT];
};

// One workgroup (four waves) per selected keypoint.  The reference adds every sample's eight trilinear shares into
// the (4+2) x (4+2) x (8+2) histogram in pixel raster order.  Here 256 work items evaluate 256 consecutive raster
// positions per round (gradient from the staged patch, rotation, exp, atan2, the eight shares) and leave the shares in
// LDS together with ballots of each sample's first cell row, first cell column and first orientation bin.  The 16 x 9
// histogram bins that reach the output (inner cells; orientation bin 9 never receives anything) are owned by one
// work item each: it intersects the ballots into the set of samples that add to its bin -- and which of their eight
// shares -- and adds them in ascending position = raster order.  Every bin receives the additions of the sequential
// loop in the sequential loop's order; the closing normalisation is the sequential code on one work item.
constexpr int DESC_T = 256, DESC_VP = 9;           // work items; pitch of a sample's shares (odd: no bank conflicts)
__global__ __launch_bounds__(DESC_T) void descriptor_kernel(pyr_ptrs P, const skp_t* __restrict__ kps,
                                                            const unsigned* __restrict__ sel,
                                                            const unsigned* __restrict__ n_sel,
                                                            float* __restrict__ rows /* n x 134 */) {
  __shared__ float s_v[DESC_T * DESC_VP];
  __shared__ unsigned long long s_m[DESC_T / 64][18];   // per wave: 5 row ballots, 5 column ballots, 8 orientation ballots
  __shared__ float s_hist[16][10];
  __shared__ float s_patch[(2 * DESC_RMAX + 3) * (2 * DESC_RMAX + 3)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned n = *n_sel;
  const int d = 4, n8 = 8;
  // the bin work item tid < 144 owns: inner cell (cr, cc) in 1..4, orientation bin ob in 0..8
  const int ci = tid / 9, ob = tid - ci * 9;
  const int cr = 1 + (ci >> 2), cc = 1 + (ci & 3);
  for (unsigned k = blockIdx.x; k < n; k += gridDim.x) {
    const skp_t q = kps[sel[k]];
    const float* g = P.g[q.oct][q.layer];
    const int H = P.H[q.oct], W = P.W[q.oct];
    float* row = rows + (size_t)k * 134;
    const float scl = q.size * 0.5f / (float)(1 << q.oct);
    float ori_deg = 360.f - q.angle;
    if (fabsf(ori_deg - 360.f) < 1.1920929e-07f) ori_deg = 0.f;
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
    const bool staged = radius <= DESC_RMAX;
    const int PW = 2 * radius + 3;
    if (staged) {
      const int np = PW * PW, y0 = pyi - radius - 1, x0 = pxi - radius - 1;
      for (int base = 0; base < np; base += DESC_T * 4) {
        float v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = base + b * DESC_T + tid;
          const int yy = i / PW, xx = i - yy * PW;
          const int gy = y0 + yy, gx = x0 + xx;
          const bool in = i < np && gy >= 0 && gy < H && gx >= 0 && gx < W;
          v[b] = in ? g[(size_t)(in ? gy : 0) * W + (in ? gx : 0)] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = base + b * DESC_T + tid;
          if (i < np) s_patch[i] = v[b];
        }
      }
    }
    __syncthreads();
    float acc = 0.f;
    const int side = 2 * radius + 1;
    const long long total = (long long)side * side;
    // raster position of this work item: (pi, pj), advanced by DESC_T positions per round without a division
    int pi = tid / side, pj = tid - pi * side;
    const int step_i = DESC_T / side, step_j = DESC_T - step_i * side;
    for (long long base = 0; base < total; base += DESC_T) {
      int ca = -8, cb = -8, co = -8;                   // first cell row / column / orientation bin (none: matches nothing)
      if (pi < side) {
        const int i = pi - radius, j = pj - radius;
        const float c_rot = j * cos_t - i * sin_t;
        const float r_rot = j * sin_t + i * cos_t;
        float rbin = r_rot + d / 2 - 0.5f;
        float cbin = c_rot + d / 2 - 0.5f;
        const int r = pyi + i, c = pxi + j;
        if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < H - 1 && c > 0 && c < W - 1) {
          float dx, dy;
          if (staged) {
            const float* qq = s_patch + (i + radius + 1) * PW + (j + radius + 1);
            dx = qq[1] - qq[-1];
            dy = qq[-PW] - qq[PW];
          } else {
            dx = g[(size_t)r * W + c + 1] - g[(size_t)r * W + c - 1];
            dy = g[(size_t)(r - 1) * W + c] - g[(size_t)(r + 1) * W + c];
          }
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
          float* vv = s_v + tid * DESC_VP;
          vv[0] = v_rco000;
          vv[1] = v_rco001;
          vv[2] = v_rco010;
          vv[3] = v_rco011;
          vv[4] = v_rco100;
          vv[5] = v_rco101;
          vv[6] = v_rco110;
          vv[7] = v_rco111;
          ca = r0 + 1;
          cb = c0 + 1;
          co = o0;
        }
      }
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        const unsigned long long ra = __ballot(ca == x), cbm = __ballot(cb == x);
        if (lane == 0) {
          s_m[wid][x] = ra;
          s_m[wid][5 + x] = cbm;
        }
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const unsigned long long om = __ballot(co == x);
        if (lane == 0) s_m[wid][10 + x] = om;
      }
      pi += step_i;
      pj += step_j;
      if (pj >= side) {
        pj -= side;
        ++pi;
      }
      __syncthreads();
      if (tid < 144) {
#pragma unroll
        for (int w = 0; w < DESC_T / 64; ++w) {
          const unsigned long long* m = s_m[w];
          const unsigned long long r1 = m[cr - 1], c1 = m[5 + cc - 1];            // second row / column of the sample's block
          const unsigned long long o1 = ob >= 1 ? m[10 + ob - 1] : 0ull;          // its second orientation bin
          const unsigned long long o0m = ob <= 7 ? m[10 + ob] : 0ull;
          unsigned long long mine = (m[cr] | r1) & (m[5 + cc] | c1) & (o0m | o1);
          const float* vw = s_v + w * 64 * DESC_VP;
          // (tried: the masks as 32-bit halves with two samples' shares requested before the first addition -- 313 us
          //  against 265 for the launch: every loop runs as long as its busiest bin, and two loops per wave add up)
          while (mine) {
            const int e = __builtin_ctzll(mine);
            mine &= mine - 1ull;
            const int share = (int)((r1 >> e) & 1ull) * 4 + (int)((c1 >> e) & 1ull) * 2 + (int)((o1 >> e) & 1ull);
            acc += vw[e * DESC_VP + share];
          }
        }
      }
      __syncthreads();
    }
    if (tid < 144) s_hist[ci][ob] = acc;
    __syncthreads();
    // fold the orientation wrap, then threshold / normalise as the sequential code does (one work item: the two norms
    // are sums in index order)
    // (everything elementwise in parallel out of LDS; the two norms are sums in index order on one work item, their
    //  128 terms read before the first addition)
    float* s_raw = s_v;                                  // 128 entries (the samples are done with)
    if (tid < 16) s_hist[tid][0] += s_hist[tid][n8];     // (hist[1] += bin 9, which is never written: + 0)
    __syncthreads();
    if (tid < 128) s_raw[tid] = s_hist[tid >> 3][tid & 7];
    __syncthreads();
    if (tid == 0) {
      float nrm2 = 0;
#pragma unroll
      for (int b = 0; b < 128; ++b) nrm2 += s_raw[b] * s_raw[b];
      s_raw[128] = sqrtf(nrm2) * 0.2f;
    }
    __syncthreads();
    const float thr = s_raw[128];
    __syncthreads();
    if (tid < 128) s_raw[tid] = s_raw[tid] < thr ? s_raw[tid] : thr;
    __syncthreads();
    if (tid == 0) {
      float nrm2 = 0;
#pragma unroll
      for (int b = 0; b < 128; ++b) nrm2 += s_raw[b] * s_raw[b];
      s_raw[129] = 512.f / fmaxf(sqrtf(nrm2), 1.1920929e-07f);
    }
    __syncthreads();
    if (tid < 128) {
      const float t = rintf(s_raw[tid] * s_raw[129]);
      row[6 + tid] = t < 0 ? 0.f : (t > 255.f ? 255.f : t);
    } else if (tid == 128) {
      row[0] = q.x * 0.5f;
      row[1] = q.y * 0.5f;
      row[2] = q.size * 0.5f;
      row[3] = q.angle;
      row[4] = q.response;
      row[5] = (float)(q.oct - 1);
    }
    __syncthreads();
  }
}

__global__ void overflow_kernel(const unsigned* cnt, unsigned subcap, unsigned cap_kp, unsigned* flag) {
  bool over = cnt[C_SURV] > cap_kp || cnt[C_KP] > cap_kp;
  for (int j = 0; j < EX_SUB; ++j) over = over || cnt[C_SUB + 32 * j] > subcap;
  if (over) *flag = 1u;
}

// ---------------------------------------------------------------------------------------------------------------
// The final order of the described rows on the device (what vo_sift's host tail does with std::sort): sort by
// (x, y, size desc, angle, response desc, octave desc) -- KeyPointsFilter::removeDuplicatedSorted's order --, drop rows
// that repeat the previous row's (x, y, size, angle), keep the `cap` strongest by response when more are left
// (KeyPointsFilter::retainBest: everything above the cap-th response, ties in order), write keypoints and descriptors
// compactly in that order.  One workgroup: a bitonic sort of (128-bit key, row) pairs in LDS -- rows whose four leading
// fields tie are settled on the rest of the row from memory --, flags and scans over the sorted order.
constexpr int FIN_T = 1024;
constexpr int FIN_MAX = 4096;          // rows the finalize kernel orders (the capped path describes cap + ties rows)

__device__ __forceinline__ unsigned order_bits(float v) {          // order-preserving float -> uint
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// full comparison of two rows whose (x, y, size, angle) are equal: response desc, octave desc, then the descriptors'
// bytes in memory order (what the host's memcmp sees), then the row number -- a strict total order
__device__ bool row_tie_less(const float* __restrict__ rows, unsigned a, unsigned b) {
  const float* ra = rows + (size_t)a * 134;
  const float* rb = rows + (size_t)b * 134;
  if (ra[4] != rb[4]) return ra[4] > rb[4];
  if (ra[5] != rb[5]) return ra[5] > rb[5];
  for (int k = 6; k < 134; ++k) {
    const unsigned wa = __builtin_bswap32(__float_as_uint(ra[k])), wb = __builtin_bswap32(__float_as_uint(rb[k]));
    if (wa != wb) return wa < wb;
  }
  return a < b;
}

// exclusive scan of up to four flags per thread in sorted-position order (position = 4 * tid + k); returns the block total
__device__ __forceinline__ unsigned block_scan4(const unsigned (&f)[4], unsigned (&excl)[4], unsigned* s_wave /*[16]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned mine = f[0] + f[1] + f[2] + f[3];
  unsigned inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  unsigned base = 0, total = 0;
  for (int w = 0; w < FIN_T / 64; ++w) {
    const unsigned v = s_wave[w];
    if (w < wv) base += v;
    total += v;
  }
  unsigned run = base + inc - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    excl[k] = run;
    run += f[k];
  }
  return total;
}

__global__ __launch_bounds__(FIN_T) void sift_finalize_kernel(const float* __restrict__ rows, const unsigned* __restrict__ n_rows,
                                                              int cap, unsigned* __restrict__ src_out, int* __restrict__ n_out,
                                                              unsigned* __restrict__ overflow) {
  extern __shared__ __align__(16) unsigned s_fin[];
  unsigned* s_key = s_fin;                       // [FIN_MAX][4]
  unsigned* s_row = s_key + 4 * FIN_MAX;         // [FIN_MAX] row number at this sorted position
  __shared__ unsigned s_wave[FIN_T / 64];
  __shared__ unsigned s_count;
  const int tid = threadIdx.x;
  const unsigned n = *n_rows;
  if (n > (unsigned)FIN_MAX) {
    if (tid == 0) {
      *overflow = 1u;
      *n_out = 0;
    }
    return;
  }
  unsigned N = 1;
  while (N < n) N <<= 1;
  if (N < 2) N = 2;
  for (unsigned i = tid; i < N; i += FIN_T) {
    if (i < n) {
      const float* r = rows + (size_t)i * 134;
      s_key[4 * i] = order_bits(r[0]);
      s_key[4 * i + 1] = order_bits(r[1]);
      s_key[4 * i + 2] = ~order_bits(r[2]);
      s_key[4 * i + 3] = order_bits(r[3]);
      s_row[i] = i;
    } else {
      s_key[4 * i] = s_key[4 * i + 1] = s_key[4 * i + 2] = s_key[4 * i + 3] = 0xffffffffu;
      s_row[i] = 0xffffffffu;                   // padding: behind every row
    }
  }
  __syncthreads();
  auto less = [&](unsigned a, unsigned b) -> bool {      // sorted positions a, b
    const uint4 ka = *reinterpret_cast<const uint4*>(s_key + 4 * a), kb = *reinterpret_cast<const uint4*>(s_key + 4 * b);
    if (ka.x != kb.x) return ka.x < kb.x;
    if (ka.y != kb.y) return ka.y < kb.y;
    if (ka.z != kb.z) return ka.z < kb.z;
    if (ka.w != kb.w) return ka.w < kb.w;
    const unsigned ra = s_row[a], rb = s_row[b];
    if (ra == 0xffffffffu || rb == 0xffffffffu) return ra < rb;     // (padding: never before a row)
    return row_tie_less(rows, ra, rb);
  };
  for (unsigned k = 2; k <= N; k <<= 1) {
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned t = tid; t < N / 2; t += FIN_T) {
        const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
        const bool up = (i & k) == 0;
        if (less(p, i) == up) {
          const uint4 ki = *reinterpret_cast<const uint4*>(s_key + 4 * i), kp2 = *reinterpret_cast<const uint4*>(s_key + 4 * p);
          *reinterpret_cast<uint4*>(s_key + 4 * i) = kp2;
          *reinterpret_cast<uint4*>(s_key + 4 * p) = ki;
          const unsigned ri = s_row[i];
          s_row[i] = s_row[p];
          s_row[p] = ri;
        }
      }
      __syncthreads();
    }
  }
  // sorted: position 4 * tid + k.  kept = not a repeat of the previous row's four leading fields
  unsigned keep[4], pos[4];
  float resp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned sidx = 4u * tid + k;
    keep[k] = 0;
    resp[k] = 0.f;
    if (sidx < n) {
      bool dup = false;
      if (sidx > 0) {
        const uint4 a = *reinterpret_cast<const uint4*>(s_key + 4 * sidx), b = *reinterpret_cast<const uint4*>(s_key + 4 * (sidx - 1));
        dup = a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
      }
      keep[k] = dup ? 0u : 1u;
      resp[k] = rows[(size_t)s_row[sidx] * 134 + 4];
    }
  }
  unsigned n_keep = block_scan4(keep, pos, s_wave);
  if ((int)n_keep > cap && cap > 0) {
    // the cap-th largest response among the kept rows: the largest bit pattern v with #(resp >= v) >= cap
    // (responses are positive: their bit patterns order like the values)
    unsigned lo = 0u, hi = 0x7f800000u;          // invariant: count_ge(lo) >= cap > count_ge(hi + 1)
    while (lo < hi) {
      const unsigned mid = lo + (hi - lo + 1) / 2;
      if (tid == 0) s_count = 0;
      __syncthreads();
      unsigned c = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) c += (keep[k] && __float_as_uint(resp[k]) >= mid) ? 1u : 0u;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
      if ((tid & 63) == 0 && c) atomicAdd(&s_count, c);
      __syncthreads();
      const unsigned total = s_count;
      __syncthreads();
      if (total >= (unsigned)cap) lo = mid;
      else hi = mid - 1;
    }
    const float thr = __uint_as_float(lo);
    unsigned above[4], tie[4], tpos[4], dummy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      above[k] = (keep[k] && resp[k] > thr) ? 1u : 0u;
      tie[k] = (keep[k] && resp[k] == thr) ? 1u : 0u;
    }
    const unsigned n_above = block_scan4(above, dummy, s_wave);
    (void)block_scan4(tie, tpos, s_wave);
    const unsigned ties = (unsigned)cap - n_above;
#pragma unroll
    for (int k = 0; k < 4; ++k) keep[k] = (above[k] || (tie[k] && tpos[k] < ties)) ? 1u : 0u;
    n_keep = block_scan4(keep, pos, s_wave);
  }
  // destination -> row; the rows themselves are moved by the launch behind this one (sift_gather_kernel: a workgroup per
  // row -- as this workgroup's tail the copy of 2000 x 134 values was half of the kernel's 215 us)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned sidx = 4u * tid + k;
    if (keep[k]) src_out[pos[k]] = s_row[sidx];
  }
  if (tid == 0) *n_out = (int)n_keep;
}

__global__ __launch_bounds__(128) void sift_gather_kernel(const float* __restrict__ rows, const unsigned* __restrict__ src,
                                                          const int* __restrict__ n_out, float* __restrict__ kp_out,
                                                          float* __restrict__ desc_out, uint8_t* __restrict__ desc_bytes) {
  const unsigned d = blockIdx.x, c = threadIdx.x;
  if ((int)d >= *n_out) return;
  const float* r = rows + (size_t)src[d] * 134;
  const float v = r[6 + c];
  if (desc_out) desc_out[(size_t)d * 128 + c] = v;
  if (desc_bytes) desc_bytes[(size_t)d * 128 + c] = (uint8_t)v;     // (whole numbers 0..255 by construction)
  if (c < 6) kp_out[(size_t)d * 6 + c] = r[c];
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

// Everything up to the described rows, enqueued on the context's stream (no host synchronisation): d_img is the frame in
// device memory; the rows (134 floats each: x, y, size, angle, response, octave, 128 descriptor values) are left in the
// context's scratch[3], their count in scratch[2][C_SEL], an overflow flag in scratch[2][C_OVER].
static int sift_enqueue(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int cap) {
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
  // one arena for the whole scale space: per octave NG Gaussian images
  size_t total = 0;
  {
    int w = W0, h = H0;
    for (int o = 0; o < n_oct; ++o) {
      total += (size_t)NG * w * h;
      w /= 2;
      h /= 2;
    }
  }
  const unsigned cap_kp = (unsigned)vo_sift_capacity(H, W), subcap = cap_kp, cap_cand = (unsigned)EX_SUB * subcap;
  VO_TRY(vo_ensure(ctx, ctx->sift_arena, total * 4));
  unsigned table_len = 1;
  while (table_len < 4u * cap_kp) table_len <<= 1;
  VO_TRY(vo_ensure(ctx, ctx->scratch[0], (size_t)cap_cand * 16));
  VO_TRY(vo_ensure(ctx, ctx->scratch[1], (size_t)cap_kp * sizeof(skp_t)));
  VO_TRY(vo_ensure(ctx, ctx->scratch[2], ((size_t)C_WORDS + FIN_MAX) * 4));   // counters, then sift_finalize's row order
  VO_TRY(vo_ensure(ctx, ctx->scratch[3], (size_t)cap_kp * 134 * 4));
  VO_TRY(vo_ensure(ctx, ctx->scratch[4], (size_t)cap_kp * sizeof(surv_t)));
  VO_TRY(vo_ensure(ctx, ctx->scratch[5], (size_t)table_len * 8));
  VO_TRY(vo_ensure(ctx, ctx->scratch[6], (size_t)cap_kp * 4));
  unsigned* d_cnt = (unsigned*)ctx->scratch[2].p;   // counters of this call, see C_CAND ..
  VO_HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, (size_t)C_WORDS * 4, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(ctx->scratch[5].p, 0, (size_t)table_len * 8, st));

  float* arena = (float*)ctx->sift_arena.p;
  float* cur = arena;
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
  auto blur = [&](hipStream_t on, const float* src, int h, int w, const taps_t& t, float* dst) {
    const dim3 grid(vo_cdiv(w, BR_W), vo_cdiv(h, BR_H));
    switch (t.r) {   // the radii of cv2.SIFT_create()'s default sigma
      case 5: hipLaunchKernelGGL(blur2d_rb_kernel<5>, grid, dim3(256), blur_geom<5>::lds, on, src, h, w, t, dst); break;
      case 6: hipLaunchKernelGGL(blur2d_rb_kernel<6>, grid, dim3(256), blur_geom<6>::lds, on, src, h, w, t, dst); break;
      case 8: hipLaunchKernelGGL(blur2d_rb_kernel<8>, grid, dim3(256), blur_geom<8>::lds, on, src, h, w, t, dst); break;
      case 10: hipLaunchKernelGGL(blur2d_rb_kernel<10>, grid, dim3(256), blur_geom<10>::lds, on, src, h, w, t, dst); break;
      case 13: hipLaunchKernelGGL(blur2d_rb_kernel<13>, grid, dim3(256), blur_geom<13>::lds, on, src, h, w, t, dst); break;
      default:
        hipLaunchKernelGGL(blur2d_kernel, dim3(vo_cdiv(w, BT_W), vo_cdiv(h, BT_H)), dim3(256),
                           (size_t)(BT_H + 2 * t.r) * (2 * BT_W + 2 * t.r) * 4, on, src, h, w, t, dst);
        break;
    }
  };
  // Octave o + 1 starts from layer NOL of octave o: the chain  base -> g1..g3 -> decimate -> g1..g3 -> ...  is the
  // critical path (the small octaves are a launch latency each); the last two layers of every octave and its extrema
  // search run beside it: octave 0's (two thirds of that work) on one stream, the smaller octaves' on another.
  if (!ctx->aux_stream) VO_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  if (!ctx->aux_stream2) VO_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream2, hipStreamNonBlocking));
  while ((int)ctx->aux_events.size() < MAX_OCT + 2) {
    hipEvent_t e;
    VO_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->aux_events.push_back(e);
  }
  const float threshold = std::floor(0.5f * contrast_thr / NOL * 255.f);
  {
    vo_prof_scope ps(ctx, VO_K_SIFT_SCALESPACE);
    // base image: doubled, blurred to sigma
    float* up = const_cast<float*>(oct[0].g[1]);   // scratch until g[1] is produced
    hipLaunchKernelGGL(upsample2_kernel, grid2(W0, H0), dim3(256), 0, st, d_img, H, W, up);
    blur(st, up, H0, W0, taps[0], const_cast<float*>(oct[0].g[0]));
    // the dependent chain first, the side work behind it.  (The host's launch rate, ~6 us per call, is what the small
    // octaves wait for; the same ~75 launches captured once and replayed with hipGraphLaunch were no faster: 552
    // against 581 frames/s.)
    for (int o = 0; o < n_oct; ++o) {
      const int w = oct[o].W, h = oct[o].H;
      if (o > 0)
        hipLaunchKernelGGL(decimate_kernel, grid2(w, h), dim3(256), 0, st, oct[o - 1].g[NOL], oct[o - 1].W, h, w,
                           const_cast<float*>(oct[o].g[0]));
      for (int i = 1; i <= NOL; ++i) blur(st, oct[o].g[i - 1], h, w, taps[i], const_cast<float*>(oct[o].g[i]));
      VO_HIP_TRY(ctx, hipEventRecord(ctx->aux_events[o], st));
    }
    for (int o = 0; o < n_oct; ++o) {
      const int w = oct[o].W, h = oct[o].H;
      hipStream_t sb = o == 0 ? ctx->aux_stream : ctx->aux_stream2;
      VO_HIP_TRY(ctx, hipStreamWaitEvent(sb, ctx->aux_events[o], 0));
      for (int i = NOL + 1; i < NG; ++i) blur(sb, oct[o].g[i - 1], h, w, taps[i], const_cast<float*>(oct[o].g[i]));
      hipLaunchKernelGGL(extrema_kernel, dim3(vo_cdiv(w, EX_W), vo_cdiv(h, EX_H)), dim3(256), 0, sb, oct[o], threshold, (int4*)ctx->scratch[0].p,
                         d_cnt + C_SUB, subcap);
    }
    VO_HIP_TRY(ctx, hipEventRecord(ctx->aux_events[MAX_OCT], ctx->aux_stream));
    VO_HIP_TRY(ctx, hipEventRecord(ctx->aux_events[MAX_OCT + 1], ctx->aux_stream2));
    VO_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->aux_events[MAX_OCT], 0));
    VO_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->aux_events[MAX_OCT + 1], 0));
  }
  VO_TRY(vo_check_launch(ctx, "sift scale space"));
  skp_t* d_kps = (skp_t*)ctx->scratch[1].p;
  surv_t* d_surv = (surv_t*)ctx->scratch[4].p;
  unsigned* d_sel = (unsigned*)ctx->scratch[6].p;
  float* d_rows = (float*)ctx->scratch[3].p;
  octs_t OS;
  memset(&OS, 0, sizeof(OS));
  for (int o = 0; o < n_oct; ++o) OS.o[o] = oct[o];
  {
    vo_prof_scope ps(ctx, VO_K_SIFT_DETECT);
    // counts stay on the device: fixed grids, every kernel strides over what the one before it produced
    hipLaunchKernelGGL(refine_kernel, dim3(1024), dim3(256), 0, st, OS, (const int4*)ctx->scratch[0].p, d_cnt + C_SUB,
                       subcap, contrast_thr, edge_thr, sigma, (unsigned long long*)ctx->scratch[5].p, table_len - 1,
                       d_surv, d_cnt + C_SURV, cap_kp);
    hipLaunchKernelGGL(orient_kernel, dim3(1024), dim3(64 * ORI_WAVES), 0, st, OS, d_surv, d_cnt + C_SURV, cap_kp, d_kps, d_cnt + C_KP,
                       cap_kp);
    hipLaunchKernelGGL(overflow_kernel, dim3(1), dim3(1), 0, st, d_cnt, subcap, cap_kp, d_cnt + C_OVER);
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, st, d_kps, d_cnt + C_KP, cap_kp, (unsigned)cap, d_sel,
                       d_cnt + C_SEL);
  }
  VO_TRY(vo_check_launch(ctx, "sift detection"));
  {
    vo_prof_scope ps(ctx, VO_K_SIFT_DESCRIBE);
    hipLaunchKernelGGL(descriptor_kernel, dim3(2048), dim3(DESC_T), 0, st, P, d_kps, d_sel, d_cnt + C_SEL, d_rows);
  }
  return vo_check_launch(ctx, "sift descriptor_kernel");
}

// the final order on the device (sift_finalize_kernel): rows -> d_kp (cap x 6), d_desc (cap x 128 float, nullable),
// d_desc_u8 (cap x 128 bytes, nullable), d_n
static int sift_finalize(vo_ctx* ctx, int cap, float* d_kp, float* d_desc, uint8_t* d_desc_u8, int* d_n) {
  static const size_t lds = (size_t)FIN_MAX * 20;
  static bool opted[64] = {false};
  if (ctx->device >= 0 && ctx->device < 64 && !opted[ctx->device]) {
    VO_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&sift_finalize_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    opted[ctx->device] = true;
  }
  unsigned* d_cnt = (unsigned*)ctx->scratch[2].p;
  unsigned* d_src = d_cnt + C_WORDS;                  // sorted position -> row (FIN_MAX words behind the counters)
  hipLaunchKernelGGL(sift_finalize_kernel, dim3(1), dim3(FIN_T), lds, ctx->stream, (const float*)ctx->scratch[3].p,
                     d_cnt + C_SEL, cap, d_src, d_n, d_cnt + C_OVER);
  VO_TRY(vo_check_launch(ctx, "sift_finalize_kernel"));
  hipLaunchKernelGGL(sift_gather_kernel, dim3(cap > 0 ? cap : FIN_MAX), dim3(128), 0, ctx->stream, (const float*)ctx->scratch[3].p,
                     (const unsigned*)d_src, (const int*)d_n, d_kp, d_desc, d_desc_u8);
  return vo_check_launch(ctx, "sift_gather_kernel");
}

int vo_sift_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int cap, float* d_kp, float* d_desc, uint8_t* d_desc_u8,
                int32_t* d_n) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_img && d_kp && d_n && (d_desc || d_desc_u8), "sift_dev: null pointer");
  VO_REQUIRE(ctx, H >= 16 && W >= 16, "sift_dev: bad arguments");
  VO_REQUIRE(ctx, cap >= 1 && cap <= FIN_MAX - 96, "sift_dev: cap must be in 1..%d", FIN_MAX - 96);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  VO_TRY(sift_enqueue(ctx, d_img, H, W, cap));
  return sift_finalize(ctx, cap, d_kp, d_desc, d_desc_u8, d_n);
}

int vo_sift(vo_ctx* ctx, const uint8_t* img, int H, int W, int cap, float* kp_out, float* desc_out, int32_t* n_out) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && kp_out && desc_out && n_out, "sift: null pointer");
  VO_REQUIRE(ctx, H >= 16 && W >= 16, "sift: bad arguments");
  const bool capped = cap > 0 && cap <= FIN_MAX - 96;      // the order, duplicates and cap on the device (rows: cap + ties)
  if (cap <= 0) cap = vo_sift_capacity(H, W);   // keep every keypoint, as cv2.SIFT_create() (nfeatures = 0) does
  *n_out = 0;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // through the context's pinned staging buffer: a DMA each way instead of the runtime's pageable-memory path
  VO_TRY(vo_ensure(ctx, ctx->img, (size_t)H * W));
  VO_TRY(vo_ensure_pinned(ctx, (size_t)H * W));
  memcpy(ctx->h_pin, img, (size_t)H * W);
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, ctx->h_pin, (size_t)H * W, hipMemcpyHostToDevice, st));
  VO_TRY(sift_enqueue(ctx, (const uint8_t*)ctx->img.p, H, W, cap));
  unsigned* d_cnt = (unsigned*)ctx->scratch[2].p;
  float* d_rows = (float*)ctx->scratch[3].p;
  if (capped) {
    // final rows made on the device: [n | kp cap x 6 | desc cap x 128] come back in one transfer
    const size_t out_bytes = 16 + (size_t)cap * 134 * 4;
    VO_TRY(vo_ensure(ctx, ctx->scratch[7], out_bytes));
    char* d_out = (char*)ctx->scratch[7].p;
    VO_TRY(sift_finalize(ctx, cap, (float*)(d_out + 16), (float*)(d_out + 16 + (size_t)cap * 24), nullptr, (int*)d_out));
    VO_HIP_TRY(ctx, hipMemcpyAsync(d_out + 4, d_cnt + C_OVER, 4, hipMemcpyDeviceToDevice, st));
    VO_TRY(vo_ensure_pinned(ctx, out_bytes));
    VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pin, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
    const int* head = (const int*)ctx->h_pin;
    if (head[1]) return vo_set_error(ctx, VO_ECAPACITY, "sift: candidate / keypoint list overflow");
    const int n = head[0];
    memcpy(kp_out, (const char*)ctx->h_pin + 16, (size_t)n * 24);
    memcpy(desc_out, (const char*)ctx->h_pin + 16 + (size_t)cap * 24, (size_t)n * 512);
    *n_out = n;
    return VO_OK;
  }
  unsigned cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  VO_HIP_TRY(ctx, hipMemcpyAsync(cnt, d_cnt, 32, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  if (cnt[C_OVER]) return vo_set_error(ctx, VO_ECAPACITY, "sift: candidate / keypoint list overflow");
  const unsigned n_all = cnt[C_SEL];      // rows described: all keypoints, or those at or above the cap's response bound
  if (n_all == 0) return VO_OK;
  VO_TRY(vo_ensure_pinned(ctx, (size_t)n_all * 134 * 4));
  const float* rows = (const float*)ctx->h_pin;
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pin, d_rows, (size_t)n_all * 134 * 4, hipMemcpyDeviceToHost, st));
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
