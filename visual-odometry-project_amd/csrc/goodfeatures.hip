// Shi-Tomasi corner detection ("good features to track") for gfx950.
//
// Reference call site: src/vo/features/klt.py:98
//   cv2.goodFeaturesToTrack(img, mask=mask, maxCorners=500, qualityLevel=0.01,
//                           minDistance=8, blockSize=7)                  (klt.py:24-26)
// The definition (restated in oracle/csrc/goodfeatures.c): min-eigenvalue map of the
// block x block structure tensor of 3x3 Sobel gradients (reflect-101 borders, exact
// integer sums scaled once), quality threshold against the masked maximum, 3x3 local
// maxima, descending order, greedy minimum-distance selection.
// Everything on the device: eigenvalue map, maximum, thresholded local maxima -> candidate keys
// (value | address), descending radix sort (rocPRIM), then the greedy minimum-distance rule itself,
// walked by one workgroup in blocks of 512 candidates: a candidate is tested against the accepted
// corners of earlier blocks through a cell grid (cell side = minDistance, as OpenCV keeps it) and
// against the earlier candidates of its own block by the rule's own recursion -- accepted when every
// earlier neighbour is rejected, rejected when one is accepted -- which settles in a few sweeps.
#include <algorithm>
#include <cmath>

#include <cstring>

#include "vo_internal.h"

#include <rocprim/rocprim.hpp>

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
                                                               unsigned long long* __restrict__ keys,
                                                               unsigned* __restrict__ count, unsigned cap) {
  // A 64 x 16 tile per workgroup, four rows per work item; the tile's maxima are collected in LDS and appended with ONE
  // reservation (one returning atomic per maximum -- or per wave -- on the one counter: ~27k / ~16k of them at 1376x1241,
  // and the launch waited for the counter 97 % of its 94 us).
  __shared__ unsigned long long s_keys[GX * GY / 4 + 64];        // (3x3 maxima: at most one per 2x2 block, ties aside)
  __shared__ unsigned s_n, s_base;
  const unsigned mk = *max_key;
  if (mk == 0) return;                                           // empty mask
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const float thr = (float)((double)key_float(mk) * quality);
  const int x = blockIdx.x * GX + (threadIdx.x & (GX - 1));
#pragma unroll
  for (int k = 0; k < GY / (GT / GX); ++k) {
    const int y = blockIdx.y * GY + k * (GT / GX) + threadIdx.x / GX;
    bool take = x >= 1 && y >= 1 && x < W - 1 && y < H - 1;
    float v = 0.f;
    if (take) {
      v = eig[(size_t)y * W + x];
      take = v > thr && v != 0.f && (!mask || mask[(size_t)y * W + x]);
    }
    if (take) {
      float m = 0.f;
#pragma unroll
      for (int j = -1; j <= 1; ++j)
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
          float q = eig[(size_t)(y + j) * W + (x + i)];
          q = q > thr ? q : 0.f;
          m = q > m ? q : m;
        }
      take = v == m;
    }
    if (take) {
      // descending order of the key = descending value, ties: higher address first (positive floats order like their bits)
      const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(y * W + x);
      const unsigned slot = atomicAdd(&s_n, 1u);
      if (slot < (unsigned)(GX * GY / 4 + 64)) {
        s_keys[slot] = key;
      } else {                                                   // (a plateau: more maxima than one per 2x2 block)
        const unsigned pos = atomicAdd(count, 1u);
        if (pos < cap) keys[pos] = key;
      }
    }
  }
  __syncthreads();
  const unsigned n = min(s_n, (unsigned)(GX * GY / 4 + 64));
  if (threadIdx.x == 0 && n) s_base = atomicAdd(count, n);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n; i += GT) {
    const unsigned pos = s_base + i;
    if (pos < cap) keys[pos] = s_keys[i];
  }
}

// The greedy rule over the sorted candidates, one workgroup.  Cell grid in global memory: per cell a count and up to
// GRID_SLOTS accepted corners (x | y << 16); accepted corners are >= minDistance apart, so a cell of that side holds
// at most four -- more than GRID_SLOTS raises `fault` and the caller falls back to nothing (an error).
constexpr int GF_T = 512, GF_NB = 24, GRID_SLOTS = 8;
enum { GF_UNDECIDED = 0, GF_ACCEPTED = 1, GF_REJECTED = 2 };

__global__ __launch_bounds__(GF_T) void greedy_distance_kernel(const unsigned long long* __restrict__ keys, unsigned nc,
                                                               int W, int cell, int gw, int gh, double md2, int max_corners,
                                                               unsigned* __restrict__ cell_cnt,
                                                               unsigned* __restrict__ cell_pts, float* __restrict__ xy,
                                                               unsigned* __restrict__ ctl /* [2] n_out, [3] fault */) {
  __shared__ int s_xy[GF_T];
  __shared__ unsigned char s_state[GF_T];
  __shared__ unsigned short s_nb[GF_T][GF_NB];
  __shared__ int s_scan[GF_T];
  __shared__ int s_open, s_total;
  const int t = threadIdx.x;
  int n_acc = 0;
  const int limit = max_corners > 0 ? max_corners : 0x7fffffff;
  for (unsigned base = 0; base < nc && n_acc < limit; base += GF_T) {
    const unsigned k = base + t;
    const bool valid = k < nc;
    int x = 0, y = 0, state = GF_REJECTED;
    if (valid) {
      const unsigned id = (unsigned)(keys[k] & 0xffffffffull);
      y = (int)(id / (unsigned)W);
      x = (int)(id - (unsigned)y * (unsigned)W);
      state = GF_UNDECIDED;
      // accepted corners of earlier blocks, through the grid
      const int cx = x / cell, cy = y / cell;
      for (int yy = max(0, cy - 1); yy <= min(gh - 1, cy + 1) && state == GF_UNDECIDED; ++yy)
        for (int xx = max(0, cx - 1); xx <= min(gw - 1, cx + 1) && state == GF_UNDECIDED; ++xx) {
          // (agent-scope loads: the grid is written by this workgroup's earlier blocks, past the CU's L1)
          const unsigned c = (unsigned)yy * gw + xx;
          const unsigned m = min(__hip_atomic_load(&cell_cnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (unsigned)GRID_SLOTS);
          for (unsigned j = 0; j < m; ++j) {
            const unsigned pt = __hip_atomic_load(&cell_pts[c * GRID_SLOTS + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double dx = x - (int)(pt & 0xffffu), dy = y - (int)(pt >> 16);
            if (dx * dx + dy * dy < md2) state = GF_REJECTED;
          }
        }
    }
    s_xy[t] = x | (y << 16);
    s_state[t] = (unsigned char)state;
    __syncthreads();
    // earlier candidates of this block within minDistance (rejected ones can be left out: they decide nothing)
    // (only corners in the 3x3 cells around the candidate count, as in the grid walk above and in OpenCV)
    const int mcx = x / cell, mcy = y / cell;
    int nnb = 0;
    if (state == GF_UNDECIDED) {
      for (int q = 0; q < t; ++q) {
        if (s_state[q] == GF_REJECTED) continue;
        const int pq = s_xy[q];
        const int qx = pq & 0xffff, qy = pq >> 16;
        const double dx = x - qx, dy = y - qy;
        if (dx * dx + dy * dy < md2 && abs(qx / cell - mcx) <= 1 && abs(qy / cell - mcy) <= 1) {
          if (nnb < GF_NB) s_nb[t][nnb] = (unsigned short)q;
          ++nnb;
        }
      }
    }
    __syncthreads();
    for (;;) {
      if (t == 0) s_open = 0;
      __syncthreads();
      int next = state;
      if (state == GF_UNDECIDED) {
        bool any_acc = false, any_und = false;
        if (nnb <= GF_NB) {
          for (int j = 0; j < nnb; ++j) {
            const int st = s_state[s_nb[t][j]];
            any_acc |= st == GF_ACCEPTED;
            any_und |= st == GF_UNDECIDED;
          }
        } else {                                     // (more neighbours than the list holds: scan the block)
          for (int q = 0; q < t; ++q) {
            const int st = s_state[q];
            if (st == GF_REJECTED) continue;
            const int pq = s_xy[q];
            const int qx = pq & 0xffff, qy = pq >> 16;
            const double dx = x - qx, dy = y - qy;
            if (dx * dx + dy * dy < md2 && abs(qx / cell - mcx) <= 1 && abs(qy / cell - mcy) <= 1) {
              any_acc |= st == GF_ACCEPTED;
              any_und |= st == GF_UNDECIDED;
            }
          }
        }
        if (any_acc) next = GF_REJECTED;
        else if (!any_und) next = GF_ACCEPTED;
        else atomicOr(&s_open, 1);
      }
      __syncthreads();                               // all reads of this sweep are done
      state = next;
      s_state[t] = (unsigned char)state;
      __syncthreads();
      if (s_open == 0) break;
      __syncthreads();
    }
    // rank of the accepted among the accepted (priority order), output, grid insertion
    const int acc = state == GF_ACCEPTED ? 1 : 0;
    s_scan[t] = acc;
    __syncthreads();
    for (int off = 1; off < GF_T; off <<= 1) {
      const int add = t >= off ? s_scan[t - off] : 0;
      __syncthreads();
      s_scan[t] += add;
      __syncthreads();
    }
    if (t == GF_T - 1) s_total = s_scan[t];
    const int rank = n_acc + s_scan[t] - acc;
    if (acc && rank < limit) {
      xy[2 * rank] = (float)x;
      xy[2 * rank + 1] = (float)y;
      const unsigned c = (unsigned)(y / cell) * gw + (x / cell);
      const unsigned slot = atomicAdd(&cell_cnt[c], 1u);
      if (slot < (unsigned)GRID_SLOTS)
        __hip_atomic_store(&cell_pts[c * GRID_SLOTS + slot], (unsigned)x | ((unsigned)y << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      else ctl[3] = 1u;
    }
    __threadfence();
    __syncthreads();
    n_acc = min(n_acc + s_total, limit);
    __syncthreads();
  }
  if (t == 0) ctl[2] = (unsigned)n_acc;
}

// The same rule over all candidates at once, by up to GC_WG workgroups of one launch: a candidate is accepted as soon as
// every earlier (higher-priority) candidate within minDistance is rejected, rejected as soon as one of them is accepted --
// the sequential walk decides exactly that, and decisions never change, so a candidate may read any mix of its
// neighbours' old and new states.  Rounds are separated by a barrier over the launch (all workgroups are resident: at most
// one per CU); the walk above takes 40 blocks of 14 us for the 20-30 thousand candidates of a 1376x1241 frame, this
// ~15 rounds of a few microseconds.
//   prep   every candidate enters the cell grid (cell side = minDistance, as in the walk)
//   lists  its earlier candidates within minDistance in the 3x3 cells around it (up to GC_NB; more: the cells are
//          walked again in every round)
//   rounds until no candidate is undecided
//   ranks  accepted candidates in priority order; the first max_corners are the corners
// A cell with more than GC_CCAP candidates, or more candidates than GC_WG workgroups hold, raises ctl[3]: the caller
// runs the one-workgroup walk instead.
constexpr int GC_T = 512, GC_WG = 256, GC_NB = 24, GC_CCAP = 32;
enum { GC_BAR = 4, GC_OPEN = 5 /* .. 8 */, GC_WGCNT = 16 /* .. 16 + GC_WG */, GC_WORDS = 16 + GC_WG };

// (a wait is bounded: a workgroup that has not been joined within ~0.2 s of device clock -- workgroups that are not all
//  resident, which the launch's size rules out -- raises *fault and goes on; every workgroup of the launch then runs out of
//  its waits the same way, and the caller, seeing the flag, runs the one-workgroup walk)
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned& target, unsigned n_wg, unsigned* fault) {
  __syncthreads();
  target += n_wg;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 20000000ull) {
        atomicOr(fault, 8u);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(GC_T) void greedy_distance_rounds_kernel(const unsigned long long* __restrict__ keys, unsigned nc,
                                                                      int W, int cell, int gw, int gh, double md2,
                                                                      int max_corners, unsigned* __restrict__ cell_cnt,
                                                                      unsigned* __restrict__ cell_items,
                                                                      unsigned* __restrict__ state,
                                                                      unsigned* __restrict__ nb, float* __restrict__ xy,
                                                                      unsigned* __restrict__ ctl) {
  __shared__ unsigned s_open, s_red[GC_T / 64], s_base;
  const int t = threadIdx.x;
  const unsigned n_wg = gridDim.x, k = blockIdx.x * GC_T + t;
  unsigned target = 0;
  const bool valid = k < nc;
  int x = 0, y = 0;
  if (valid) {
    const unsigned id = (unsigned)(keys[k] & 0xffffffffull);
    y = (int)(id / (unsigned)W);
    x = (int)(id - (unsigned)y * (unsigned)W);
    const unsigned c = (unsigned)(y / cell) * gw + (x / cell);
    const unsigned slot = atomicAdd(&cell_cnt[c], 1u);
    if (slot < (unsigned)GC_CCAP) cell_items[c * GC_CCAP + slot] = k;
    else atomicOr(&ctl[3], 2u);
    state[k] = GF_UNDECIDED;
  }
  grid_barrier(ctl + GC_BAR, target, n_wg, ctl + 3);
  if (__hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // (uniform: read behind the barrier)
  // earlier candidates within minDistance
  const int cx = x / cell, cy = y / cell;
  auto walk = [&](auto&& visit) {
    for (int yy = max(0, cy - 1); yy <= min(gh - 1, cy + 1); ++yy)
      for (int xx = max(0, cx - 1); xx <= min(gw - 1, cx + 1); ++xx) {
        const unsigned c = (unsigned)yy * gw + xx;
        const unsigned m = min(cell_cnt[c], (unsigned)GC_CCAP);
        for (unsigned j = 0; j < m; ++j) {
          const unsigned q = cell_items[c * GC_CCAP + j];
          if (q >= k) continue;
          const unsigned idq = (unsigned)(keys[q] & 0xffffffffull);
          const int qy = (int)(idq / (unsigned)W), qx = (int)(idq - (unsigned)qy * (unsigned)W);
          const double dx = x - qx, dy = y - qy;
          if (dx * dx + dy * dy < md2) visit(q);
        }
      }
  };
  int nnb = 0;
  if (valid) walk([&](unsigned q) {
    if (nnb < GC_NB) nb[(size_t)k * GC_NB + nnb] = q;
    ++nnb;
  });
  unsigned st = valid ? (unsigned)GF_UNDECIDED : (unsigned)GF_REJECTED;
  for (unsigned round = 0;; ++round) {
    if (t == 0) s_open = 0;
    __syncthreads();
    if (st == GF_UNDECIDED) {
      bool any_acc = false, any_und = false;
      auto look = [&](unsigned q) {
        const unsigned sq = __hip_atomic_load(&state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        any_acc |= sq == GF_ACCEPTED;
        any_und |= sq == GF_UNDECIDED;
      };
      if (nnb <= GC_NB) {
        for (int j = 0; j < nnb; ++j) look(nb[(size_t)k * GC_NB + j]);
      } else {
        walk(look);
      }
      if (any_acc) st = GF_REJECTED;
      else if (!any_und) st = GF_ACCEPTED;
      if (st != GF_UNDECIDED) __hip_atomic_store(&state[k], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else s_open = 1;
    }
    __syncthreads();
    unsigned* open = ctl + GC_OPEN + (round & 3u);
    if (t == 0) {
      if (s_open) __hip_atomic_store(open, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0)       // the word of two rounds on: its readers all passed the previous barrier
        __hip_atomic_store(ctl + GC_OPEN + ((round + 2u) & 3u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(ctl + GC_BAR, target, n_wg, ctl + 3);
    if (__hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // (a barrier gave up)
    if (__hip_atomic_load(open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;   // (uniform over the launch)
    if (round > nc) {            // (every round decides the first undecided candidate: never reached)
      if (t == 0) atomicOr(&ctl[3], 4u);
      return;
    }
  }
  // ranks: accepted candidates before this one = those of the earlier workgroups + those before it here
  const unsigned acc = st == GF_ACCEPTED ? 1u : 0u;
  const unsigned long long bal = __ballot(acc != 0u);
  const int lane = t & 63, wv = t >> 6;
  if (lane == 0) s_red[wv] = (unsigned)__popcll(bal);
  __syncthreads();
  unsigned before = (unsigned)__popcll(bal & ((1ull << lane) - 1ull)), mine = 0;
  for (int w = 0; w < GC_T / 64; ++w) {
    if (w < wv) before += s_red[w];
    mine += s_red[w];
  }
  if (t == 0) __hip_atomic_store(ctl + GC_WGCNT + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  grid_barrier(ctl + GC_BAR, target, n_wg, ctl + 3);
  unsigned part = 0;
  for (unsigned w = t; w < blockIdx.x; w += GC_T) part += __hip_atomic_load(ctl + GC_WGCNT + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  __syncthreads();
  if (lane == 0) s_red[wv] = part;
  __syncthreads();
  if (t == 0) {
    unsigned b = 0;
    for (int w = 0; w < GC_T / 64; ++w) b += s_red[w];
    s_base = b;
  }
  __syncthreads();
  const unsigned limit = max_corners > 0 ? (unsigned)max_corners : 0xffffffffu;
  const unsigned rank = s_base + before;
  if (acc && rank < limit) {
    xy[2 * rank] = (float)x;
    xy[2 * rank + 1] = (float)y;
  }
  if (blockIdx.x == n_wg - 1 && t == 0) ctl[2] = min(s_base + mine, limit);
}

// minDistance < 1: the first max_corners of the sorted list
__global__ __launch_bounds__(256) void take_sorted_kernel(const unsigned long long* __restrict__ keys, unsigned n, int W,
                                                          float* __restrict__ xy) {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const unsigned id = (unsigned)(keys[k] & 0xffffffffull);
  xy[2 * k] = (float)(id % (unsigned)W);
  xy[2 * k + 1] = (float)(id / (unsigned)W);
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
  VO_REQUIRE(ctx, W < 65536 && H < 65536, "good_features: image side must be below 65536");
  const int cell = std::max(1, (int)std::lround(min_dist));
  const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
  const size_t cells = (size_t)gw * gh;
  const size_t out_cap = max_corners > 0 ? (size_t)max_corners : (size_t)cap;
  size_t sort_tmp = 0;
  VO_HIP_TRY(ctx, rocprim::radix_sort_keys_desc(nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                (size_t)cap, 0, 64, st));
  VO_TRY(vo_ensure(ctx, ctx->img, px));
  VO_TRY(vo_ensure(ctx, s[0], px * 4));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)GC_WORDS * 4));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)cap * 8));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)cap * 8));
  VO_TRY(vo_ensure(ctx, s[4], sort_tmp + 256));
  VO_TRY(vo_ensure(ctx, s[5], cells * 4));
  VO_TRY(vo_ensure(ctx, s[6], cells * GRID_SLOTS * 4));
  VO_TRY(vo_ensure(ctx, s[7], out_cap * 8));
  if (mask) VO_TRY(vo_ensure(ctx, ctx->img2, px));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, px, hipMemcpyHostToDevice, st));
  if (mask) VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img2.p, mask, px, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(s[1].p, 0, (size_t)GC_WORDS * 4, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(s[5].p, 0, cells * 4, st));
  const uint8_t* d_mask = mask ? (const uint8_t*)ctx->img2.p : nullptr;
  unsigned* d_ctl = (unsigned*)s[1].p;                            // [0] max key, [1] candidate count, [2] corners, [3] fault
  const double scale = 1.0 / (4.0 * block * 255.0);
  const int RW = GX + block - 1, RH = GY + block - 1;
  const size_t lds = ((size_t)RW * RH + (size_t)3 * RH * GX) * 4;
  hipLaunchKernelGGL(min_eig_kernel, dim3(vo_cdiv(W, GX), vo_cdiv(H, GY)), dim3(GT), lds, st, (const uint8_t*)ctx->img.p,
                     H, W, block, (float)(scale * scale), d_mask, (float*)s[0].p, d_ctl);
  VO_TRY(vo_check_launch(ctx, "min_eig_kernel"));
  hipLaunchKernelGGL(corner_candidates_kernel, dim3(vo_cdiv(W, GX), vo_cdiv(H, GY)), dim3(GT), 0, st,
                     (const float*)s[0].p, H, W, d_mask, d_ctl, quality, (unsigned long long*)s[2].p, d_ctl + 1, cap);
  VO_TRY(vo_check_launch(ctx, "corner_candidates_kernel"));
  unsigned ctl[4] = {0, 0, 0, 0};
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctl, d_ctl, 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));                      // (the sort is sized by the candidate count)
  if (ctl[1] > cap)      // (ties count as maxima: plateaus can exceed one maximum per 2x2 block)
    return vo_set_error(ctx, VO_ECAPACITY, "good_features: %u local maxima exceed the candidate capacity %u", ctl[1], cap);
  const unsigned nc = ctl[1];
  if (nc == 0) return VO_OK;
  unsigned long long* d_sorted = (unsigned long long*)s[3].p;
  VO_HIP_TRY(ctx, rocprim::radix_sort_keys_desc(s[4].p, sort_tmp, (unsigned long long*)s[2].p, d_sorted, (size_t)nc, 0, 64, st));
  float* d_xy = (float*)s[7].p;
  int n = 0;
  if (min_dist >= 1) {
    static const bool walk_only = getenv("VO_GREEDY_WALK") != nullptr;      // (measurements: the one-workgroup walk)
    const unsigned n_wg = (nc + GC_T - 1) / GC_T;
    bool done = false;
    if (!walk_only && n_wg <= (unsigned)GC_WG) {
      VO_TRY(vo_ensure(ctx, s[8], cells * GC_CCAP * 4));
      VO_TRY(vo_ensure(ctx, s[9], (size_t)nc * 4));
      VO_TRY(vo_ensure(ctx, s[10], (size_t)nc * GC_NB * 4));
      hipLaunchKernelGGL(greedy_distance_rounds_kernel, dim3(n_wg), dim3(GC_T), 0, st, d_sorted, nc, W, cell, gw, gh,
                         min_dist * min_dist, max_corners, (unsigned*)s[5].p, (unsigned*)s[8].p, (unsigned*)s[9].p,
                         (unsigned*)s[10].p, d_xy, d_ctl);
      VO_TRY(vo_check_launch(ctx, "greedy_distance_rounds_kernel"));
      VO_HIP_TRY(ctx, hipMemcpyAsync(ctl, d_ctl, 16, hipMemcpyDeviceToHost, st));
      VO_HIP_TRY(ctx, hipStreamSynchronize(st));
      done = ctl[3] == 0;
      if (!done) {                     // a crowded cell: the walk decides (its grid holds accepted corners only)
        VO_HIP_TRY(ctx, hipMemsetAsync(s[5].p, 0, cells * 4, st));
        VO_HIP_TRY(ctx, hipMemsetAsync(d_ctl + 2, 0, 8, st));
      }
    }
    if (!done) {
      hipLaunchKernelGGL(greedy_distance_kernel, dim3(1), dim3(GF_T), 0, st, d_sorted, nc, W, cell, gw, gh,
                         min_dist * min_dist, max_corners, (unsigned*)s[5].p, (unsigned*)s[6].p, d_xy, d_ctl);
      VO_TRY(vo_check_launch(ctx, "greedy_distance_kernel"));
      VO_HIP_TRY(ctx, hipMemcpyAsync(ctl, d_ctl, 16, hipMemcpyDeviceToHost, st));
      VO_HIP_TRY(ctx, hipStreamSynchronize(st));
      if (ctl[3]) return vo_set_error(ctx, VO_ECAPACITY, "good_features: more than %d corners in one grid cell", GRID_SLOTS);
    }
    n = (int)ctl[2];
  } else {
    n = (int)(max_corners > 0 ? std::min<unsigned>(nc, (unsigned)max_corners) : nc);
    hipLaunchKernelGGL(take_sorted_kernel, dim3(vo_cdiv(n, 256)), dim3(256), 0, st, d_sorted, (unsigned)n, W, d_xy);
    VO_TRY(vo_check_launch(ctx, "take_sorted_kernel"));
  }
  if (n > 0) {
    VO_HIP_TRY(ctx, hipMemcpyAsync(xy, d_xy, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  *n_out = n;
  return VO_OK;
}

}  // extern "C"
