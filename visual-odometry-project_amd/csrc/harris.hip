// Harris response, exact greedy NMS and raw-patch descriptors for gfx950.
//
// Reference behaviour: src/vo/features/harris.py:86-194 (HarrisCornerDetector).
//
// Response (harris.py:99-137): integer Sobel products, integer box sums, then
// exactly three IEEE double roundings  r = det - kappa * (trace * trace)  with FP
// contraction disabled, so the map is bit-identical to the NumPy result.
//
// NMS (harris.py:139-152) is a sequential greedy argmax loop (2*N full-map scans
// on the CPU).  The same selection is computed in parallel:
//   priority(p)  = (score desc, flat index asc)            -- np.argmax tie-break
//   L1           = pixels that are the strict priority maximum of their own
//                  (2r+1)^2 window.  Every L1 pixel is selected by the greedy
//                  loop, and it suppresses every other pixel of its window.
//   A1           = positive pixels with no L1 pixel within Chebyshev distance r.
//                  A selected pixel that is not L1 must be in A1, and whether an
//                  A1 pixel is selected depends only on higher-priority selected
//                  A1 pixels (L1 pixels are never within r of it).
//   T            = a score bound with at least N L1 pixels at or above it
//                  (65536-bin histogram of the IEEE bit pattern).  Pixels below T
//                  cannot be among the first N picks.
//   rounds       = the A1 pixels at or above T are resolved by a few launches of the
//                  greedy rule itself, in parallel: a live pixel that sees no live
//                  higher-priority pixel in its window is selected and kills its
//                  window; one that sees a selected pixel dies.  State changes are
//                  monotone (live -> selected | dead), so a stale read only delays a
//                  decision to the next launch.  ~3.5x fewer live pixels per round.
//   select       = {L1 >= T} + round-selected + the few pixels still live, sorted by
//                  priority and walked by one workgroup: decided entries pass, live
//                  entries run the greedy test against selected live entries only.
// The first N selected, in priority order, are the reference's keypoints; the
// reference's slicing corner cases are applied when writing out.
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int TX = 64;   // tile width  (outputs per workgroup row)
constexpr int TY = 16;   // tile height
constexpr int NT = 256;  // threads per workgroup

// ---------------------------------------------------------------------------------
// Harris response
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void harris_response_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                             int p, double kappa,
                                                             double* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int pr = p >> 1;
  const int GW = TX + 2 * pr, GH = TY + 2 * pr;   // gradient region
  const int IW = GW + 2, IH = GH + 2;             // image region
  const int IWp = (IW + 3) & ~3;
  uint8_t* s_img = smem;
  int* s_g = reinterpret_cast<int*>(smem + ((IWp * IH + 15) & ~15));   // packed (Ix | Iy << 16)
  int* s_hxx = s_g + GW * GH;
  int* s_hyy = s_hxx + GH * TX;
  int* s_hxy = s_hyy + GH * TX;

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int ix0 = x0 - pr - 1, iy0 = y0 - pr - 1;   // image coords of s_img[0][0]

  // A: image tile + halo (zeros outside the image; such pixels only feed outputs
  //    that the border rule forces to 0)
  for (int i = tid; i < IWp * IH; i += NT) {
    int ly = i / IWp, lx = i - ly * IWp;
    int gy = iy0 + ly, gx = ix0 + lx;
    uint8_t v = 0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(size_t)gy * W + gx];
    s_img[i] = v;
  }
  __syncthreads();

  // B: Sobel as a TRUE convolution (kernel flipped): left minus right, top minus bottom
  for (int i = tid; i < GW * GH; i += NT) {
    int ly = i / GW, lx = i - ly * GW;
    const uint8_t* r0 = s_img + ly * IWp + lx;      // row above centre
    const uint8_t* r1 = r0 + IWp;
    const uint8_t* r2 = r1 + IWp;
    int a00 = r0[0], a01 = r0[1], a02 = r0[2];
    int a10 = r1[0], a12 = r1[2];
    int a20 = r2[0], a21 = r2[1], a22 = r2[2];
    int gx = (a00 - a02) + 2 * (a10 - a12) + (a20 - a22);
    int gy = (a00 - a20) + 2 * (a01 - a21) + (a02 - a22);
    s_g[i] = (gx & 0xffff) | (gy << 16);
  }
  __syncthreads();

  // C: horizontal box sums of the three products
  for (int i = tid; i < GH * TX; i += NT) {
    int ly = i / TX, lx = i - ly * TX;
    const int* g = s_g + ly * GW + lx;
    int sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < p; ++k) {
      int v = g[k];
      int gx = (int)(short)(v & 0xffff), gy = v >> 16;
      sxx += gx * gx;
      syy += gy * gy;
      sxy += gx * gy;
    }
    s_hxx[i] = sxx;
    s_hyy[i] = syy;
    s_hxy[i] = sxy;
  }
  __syncthreads();

  // D: vertical sums + response
  const int lx = tid & (TX - 1);
  const int border = pr + 1;
  for (int ly = tid / TX; ly < TY; ly += NT / TX) {
    int gy = y0 + ly, gx = x0 + lx;
    if (gy >= H || gx >= W) continue;
    double r = 0.0;
    if (gy >= border && gy < H - border && gx >= border && gx < W - border) {
      int sxx = 0, syy = 0, sxy = 0;
      for (int k = 0; k < p; ++k) {
        int j = (ly + k) * TX + lx;
        sxx += s_hxx[j];
        syy += s_hyy[j];
        sxy += s_hxy[j];
      }
      double dxx = (double)sxx, dyy = (double)syy, dxy = (double)sxy;
      double trace = dxx + dyy;
      double det = dxx * dyy - dxy * dxy;
      r = det - kappa * (trace * trace);
      if (r < 0) r = 0;
    }
    out[(size_t)gy * W + gx] = r;
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 1: L1 / A1 candidate lists + histogram of L1 scores
// ---------------------------------------------------------------------------------
struct nms_ctl {
  unsigned n_l1, n_a1, n_c, overflow;
  unsigned long long t_bits;
  unsigned n_sel, n_cand;
};

constexpr int HIST_SHIFT = 47;           // 65536 bins over non-negative doubles
constexpr int HIST_BINS = 1 << 16;

__global__ __launch_bounds__(NT) void nms_candidates_kernel(const double* __restrict__ sc, int H, int W, int r,
                                                            unsigned long long* __restrict__ keys_l1,
                                                            unsigned* __restrict__ idx_l1,
                                                            unsigned long long* __restrict__ keys_a1,
                                                            unsigned* __restrict__ idx_a1,
                                                            unsigned* __restrict__ hist, nms_ctl* ctl,
                                                            unsigned cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int RW = TX + 4 * r, RH = TY + 4 * r;   // score region (2r halo)
  const int LW = TX + 2 * r, LH = TY + 2 * r;   // region where L1 flags are needed
  double* s_sc = reinterpret_cast<double*>(smem);
  uint8_t* s_l1 = smem + (size_t)RW * RH * sizeof(double);
  uint8_t* s_cov = s_l1 + ((LW * LH + 15) & ~15);
  __shared__ unsigned s_cnt[2], s_base[2];

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  if (tid < 2) s_cnt[tid] = 0;

  for (int i = tid; i < RW * RH; i += NT) {
    int ly = i / RW, lx = i - ly * RW;
    int gy = y0 - 2 * r + ly, gx = x0 - 2 * r + lx;
    double v = 0.0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = sc[(size_t)gy * W + gx];
    s_sc[i] = v;
  }
  for (int i = tid; i < TX * TY; i += NT) s_cov[i] = 0;
  __syncthreads();

  // L1 flags on the tile + r halo
  for (int i = tid; i < LW * LH; i += NT) {
    int ly = i / LW, lx = i - ly * LW;
    const double* c = s_sc + (ly + r) * RW + (lx + r);
    double s = *c;
    bool is = s > 0.0;
    if (is && r > 0) {
      // cheap 3x3 pre-test, then the full window
      is = !(c[-RW - 1] >= s || c[-RW] >= s || c[-RW + 1] >= s || c[-1] >= s || c[1] > s ||
             c[RW - 1] > s || c[RW] > s || c[RW + 1] > s);
      for (int dy = -r; is && dy <= r; ++dy) {
        const double* row = c + dy * RW;
        for (int dx = -r; dx <= r; ++dx) {
          double q = row[dx];
          bool before = (dy < 0) || (dy == 0 && dx < 0);   // q precedes p in flat order
          if (before ? (q >= s) : (q > s && !(dy == 0 && dx == 0))) {
            is = false;
            break;
          }
        }
      }
    }
    s_l1[i] = is ? 1 : 0;
  }
  __syncthreads();

  // every L1 pixel covers its window
  for (int i = tid; i < LW * LH; i += NT) {
    if (!s_l1[i]) continue;
    int ly = i / LW - r, lx = i % LW - r;   // tile coords of the L1 pixel
    int ya = max(ly - r, 0), yb = min(ly + r, TY - 1);
    int xa = max(lx - r, 0), xb = min(lx + r, TX - 1);
    for (int y = ya; y <= yb; ++y)
      for (int x = xa; x <= xb; ++x) s_cov[y * TX + x] = 1;
  }
  __syncthreads();

  // classify the tile's own pixels
  constexpr int PER = TX * TY / NT;
  int kind[PER];
  unsigned slot[PER];
  const int lx = tid & (TX - 1);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int ly = tid / TX + k * (NT / TX);
    int gy = y0 + ly, gx = x0 + lx;
    kind[k] = -1;
    if (gy < H && gx < W) {
      double s = s_sc[(ly + 2 * r) * RW + (lx + 2 * r)];
      if (s > 0.0) {
        if (s_l1[(ly + r) * LW + (lx + r)]) kind[k] = 0;
        else if (!s_cov[ly * TX + lx]) kind[k] = 1;
      }
    }
    if (kind[k] >= 0) slot[k] = atomicAdd(&s_cnt[kind[k]], 1u);
  }
  __syncthreads();
  if (tid < 2) {
    unsigned n = s_cnt[tid];
    s_base[tid] = n ? atomicAdd(tid == 0 ? &ctl->n_l1 : &ctl->n_a1, n) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (kind[k] < 0) continue;
    int ly = tid / TX + k * (NT / TX);
    int gy = y0 + ly, gx = x0 + lx;
    unsigned long long key = (unsigned long long)__double_as_longlong(s_sc[(ly + 2 * r) * RW + (lx + 2 * r)]);
    unsigned idx = (unsigned)gy * (unsigned)W + (unsigned)gx;
    unsigned pos = s_base[kind[k]] + slot[k];
    if (pos >= cap) {
      ctl->overflow = 1;
      continue;
    }
    if (kind[k] == 0) {
      keys_l1[pos] = key;
      idx_l1[pos] = idx;
      atomicAdd(&hist[key >> HIST_SHIFT], 1u);
    } else {
      keys_a1[pos] = key;
      idx_a1[pos] = idx;
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 2: score bound with >= N L1 entries above it; clears the histogram
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void nms_threshold_kernel(unsigned* __restrict__ hist, nms_ctl* ctl, int N) {
  // wave w owns bins [4096 w, 4096 w + 4096); lane l holds bins 4096 w + 64 j + l, j = 0..63 (coalesced rows)
  __shared__ unsigned s_wave[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned v[64];
  unsigned part = 0;
  unsigned* base = hist + wv * 4096 + lane;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    v[j] = base[64 * j];
    base[64 * j] = 0;
    part += v[j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) s_wave[wv] = part;
  __syncthreads();
  unsigned above = 0, total = 0;
  for (int w = 15; w >= 0; --w) {
    if (w > wv) above += s_wave[w];
    total += s_wave[w];
  }
  if (tid == 0 && total < (unsigned)N) ctl->t_bits = 1ull;   // fewer than N strict maxima: keep everything
  if (total >= (unsigned)N && above < (unsigned)N && above + s_wave[wv] >= (unsigned)N) {
    // the crossing lies in this wave's 4096 bins: walk its rows from the top
    unsigned acc = above;
    int bin = -1;
#pragma unroll
    for (int j = 63; j >= 0; --j) {
      unsigned row = v[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) row += __shfl_xor(row, off);
      if (bin < 0 && acc + row >= (unsigned)N) {
        // inside row j: suffix over lanes (higher lane = higher bin)
        unsigned suf = v[j];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          unsigned o = __shfl_down(suf, off);
          if (lane + off < 64) suf += o;
        }
        // suf = sum of v[j] over lanes >= lane; crossing lane = highest lane with acc + suf >= N
        unsigned long long m = __ballot(acc + suf >= (unsigned)N);
        int hl = 63 - __builtin_clzll(m);
        bin = wv * 4096 + 64 * j + hl;
      }
      if (bin < 0) acc += row;
    }
    if (lane == 0) {
      unsigned long long t = (unsigned long long)bin << HIST_SHIFT;
      ctl->t_bits = t ? t : 1ull;
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 3: candidates at or above the bound -> one list (bit 0 of idx = "is A1")
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void nms_compact_kernel(const unsigned long long* __restrict__ keys_l1,
                                                         const unsigned* __restrict__ idx_l1,
                                                         const unsigned long long* __restrict__ keys_a1,
                                                         const unsigned* __restrict__ idx_a1,
                                                         unsigned long long* __restrict__ keys_c,
                                                         unsigned* __restrict__ idx_c, unsigned* __restrict__ cand,
                                                         uint8_t* __restrict__ alive, nms_ctl* ctl, unsigned cap_c,
                                                         unsigned cap_cand) {
  // L1 entries at or above the bound are selected outright; A1 entries become live candidates
  const unsigned n_l1 = ctl->n_l1, n_a1 = ctl->n_a1;
  const unsigned long long t = ctl->t_bits;
  const unsigned total = n_l1 + n_a1;
  for (unsigned i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
    const bool a = i >= n_l1;
    const unsigned j = a ? i - n_l1 : i;
    const unsigned long long key = a ? keys_a1[j] : keys_l1[j];
    if (key < t) continue;
    const unsigned idx = a ? idx_a1[j] : idx_l1[j];
    if (a) {
      const unsigned pos = atomicAdd(&ctl->n_cand, 1u);
      if (pos >= cap_cand) {
        ctl->overflow = 1;
        continue;
      }
      cand[pos] = idx;
      alive[idx] = 1;
    } else {
      const unsigned pos = atomicAdd(&ctl->n_c, 1u);
      if (pos >= cap_c) {
        ctl->overflow = 1;
        continue;
      }
      keys_c[pos] = key;
      idx_c[pos] = idx << 1;
    }
  }
}

// One parallel round of the greedy rule on the live candidates (alive: 0 dead / not a
// candidate, 1 live, 2 selected).  Safe under stale reads: see the file header.
__global__ __launch_bounds__(NT) void nms_round_kernel(const double* __restrict__ sc, uint8_t* alive,
                                                       const unsigned* __restrict__ cand,
                                                       unsigned long long* __restrict__ keys_c,
                                                       unsigned* __restrict__ idx_c, nms_ctl* ctl, unsigned cap_c,
                                                       unsigned cap_cand, int H, int W, int r) {
  const unsigned n = min(ctl->n_cand, cap_cand);
  for (unsigned i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const unsigned idx = cand[i];
    if (alive[idx] != 1) continue;
    const int py = (int)(idx / (unsigned)W), px = (int)(idx - (unsigned)py * (unsigned)W);
    const double s = sc[idx];
    const int ya = max(py - r, 0), yb = min(py + r, H - 1);
    const int xa = max(px - r, 0), xb = min(px + r, W - 1);
    bool dead = false, blocked = false;
    for (int y = ya; y <= yb && !dead; ++y) {
      const uint8_t* arow = alive + (size_t)y * W;
      const double* srow = sc + (size_t)y * W;
      for (int x = xa; x <= xb; ++x) {
        const uint8_t av = arow[x];
        if (av == 0 || (y == py && x == px)) continue;
        if (av == 2) {
          dead = true;
          break;
        }
        if (!blocked) {
          const double q = srow[x];
          const bool before = (y < py) || (y == py && x < px);
          if (before ? (q >= s) : (q > s)) blocked = true;
        }
      }
    }
    if (dead) {
      alive[idx] = 0;
      continue;
    }
    if (blocked) continue;
    alive[idx] = 2;
    const unsigned pos = atomicAdd(&ctl->n_c, 1u);
    if (pos < cap_c) {
      keys_c[pos] = (unsigned long long)__double_as_longlong(s);
      idx_c[pos] = idx << 1;
    } else {
      ctl->overflow = 1;
    }
    for (int y = ya; y <= yb; ++y) {
      uint8_t* arow = alive + (size_t)y * W;
      for (int x = xa; x <= xb; ++x)
        if (arow[x] == 1) arow[x] = 0;
    }
  }
}

// After the rounds: candidates still live join the list as undecided entries (bit 0 set);
// every candidate's mark is cleared so the map is all-zero for the next call.
__global__ __launch_bounds__(NT) void nms_collect_kernel(const double* __restrict__ sc, uint8_t* alive,
                                                         const unsigned* __restrict__ cand,
                                                         unsigned long long* __restrict__ keys_c,
                                                         unsigned* __restrict__ idx_c, nms_ctl* ctl, unsigned cap_c,
                                                         unsigned cap_cand) {
  const unsigned n = min(ctl->n_cand, cap_cand);
  for (unsigned i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const unsigned idx = cand[i];
    const uint8_t a = alive[idx];
    if (a == 0) continue;
    alive[idx] = 0;
    if (a == 1) {
      const unsigned pos = atomicAdd(&ctl->n_c, 1u);
      if (pos < cap_c) {
        keys_c[pos] = (unsigned long long)__double_as_longlong(sc[idx]);
        idx_c[pos] = (idx << 1) | 1u;
      } else {
        ctl->overflow = 1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 4 (one workgroup): sort by priority, greedy walk, write keypoints
// ---------------------------------------------------------------------------------
constexpr int SEL_T = 1024;       // threads
constexpr int CHUNK = 8192;       // entries sorted in LDS at a time
constexpr int MAX_N = 16384;      // keypoints
constexpr int NMS_ROUNDS = 6;     // parallel greedy rounds before the single-workgroup walk

__device__ __forceinline__ bool prio_before(unsigned long long ka, unsigned ia, unsigned long long kb,
                                            unsigned ib) {
  // higher score first; equal scores: lower flat index first (bit 0 is a flag, idx is unique)
  return ka > kb || (ka == kb && ia < ib);
}

// bitonic steps j = j_hi .. 1 for merge size k on one LDS-resident chunk starting at global offset g0
__device__ void bitonic_lds_steps(unsigned long long* sk, unsigned* si, int n, unsigned g0, unsigned k,
                                  unsigned j_hi) {
  for (unsigned j = j_hi; j > 0; j >>= 1) {
    for (unsigned t = threadIdx.x; t < (unsigned)n / 2; t += SEL_T) {
      unsigned lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      unsigned hi = lo | j;
      bool desc_first = (((g0 + lo) & k) == 0);   // this run sorted "priority first"
      unsigned long long ka = sk[lo], kb = sk[hi];
      unsigned ia = si[lo], ib = si[hi];
      bool swap = desc_first ? prio_before(kb, ib, ka, ia) : prio_before(ka, ia, kb, ib);
      if (swap) {
        sk[lo] = kb;
        sk[hi] = ka;
        si[lo] = ib;
        si[hi] = ia;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(SEL_T) void nms_select_kernel(unsigned long long* __restrict__ keys_c,
                                                           unsigned* __restrict__ idx_c, nms_ctl* ctl,
                                                           unsigned cap_pow2, int W, int N, int r,
                                                           unsigned* __restrict__ sel,
                                                           double* __restrict__ kp_xy) {
  __shared__ __align__(16) unsigned long long s_keys[CHUNK];   // 64 KiB
  __shared__ __align__(16) unsigned s_idx[CHUNK];              // 32 KiB
  __shared__ unsigned s_wsum[SEL_T / 64];
  __shared__ unsigned s_nsel, s_nsela, s_nalist, s_flag, s_edge;
  const int tid = threadIdx.x;

  const unsigned M = min(ctl->n_c, cap_pow2);
  unsigned Mp = 1;
  while (Mp < M) Mp <<= 1;
  if (Mp < 2) Mp = 2;
  // sentinels (lowest priority) behind the real entries
  for (unsigned i = M + tid; i < Mp; i += SEL_T) {
    keys_c[i] = 0ull;
    idx_c[i] = 0xffffffffu;
  }
  __syncthreads();

  // ---- sort: LDS bitonic per chunk, global steps for strides >= CHUNK ----
  const unsigned chunk = Mp < (unsigned)CHUNK ? Mp : (unsigned)CHUNK;
  for (unsigned g0 = 0; g0 < Mp; g0 += chunk) {
    for (unsigned i = tid; i < chunk; i += SEL_T) {
      s_keys[i] = keys_c[g0 + i];
      s_idx[i] = idx_c[g0 + i];
    }
    __syncthreads();
    for (unsigned k = 2; k <= chunk; k <<= 1) bitonic_lds_steps(s_keys, s_idx, chunk, g0, k, k >> 1);
    for (unsigned i = tid; i < chunk; i += SEL_T) {
      keys_c[g0 + i] = s_keys[i];
      idx_c[g0 + i] = s_idx[i];
    }
    __syncthreads();
  }
  for (unsigned k = chunk << 1; k <= Mp; k <<= 1) {
    for (unsigned j = k >> 1; j >= chunk; j >>= 1) {
      for (unsigned t = tid; t < Mp / 2; t += SEL_T) {
        unsigned lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        unsigned hi = lo | j;
        bool desc_first = ((lo & k) == 0);
        unsigned long long ka = keys_c[lo], kb = keys_c[hi];
        unsigned ia = idx_c[lo], ib = idx_c[hi];
        bool swap = desc_first ? prio_before(kb, ib, ka, ia) : prio_before(ka, ia, kb, ib);
        if (swap) {
          keys_c[lo] = kb;
          keys_c[hi] = ka;
          idx_c[lo] = ib;
          idx_c[hi] = ia;
        }
      }
      __syncthreads();
    }
    for (unsigned g0 = 0; g0 < Mp; g0 += chunk) {
      for (unsigned i = tid; i < chunk; i += SEL_T) {
        s_keys[i] = keys_c[g0 + i];
        s_idx[i] = idx_c[g0 + i];
      }
      __syncthreads();
      bitonic_lds_steps(s_keys, s_idx, chunk, g0, k, chunk >> 1);
      for (unsigned i = tid; i < chunk; i += SEL_T) {
        keys_c[g0 + i] = s_keys[i];
        idx_c[g0 + i] = s_idx[i];
      }
      __syncthreads();
    }
  }

  // ---- greedy walk in priority order ----
  // LDS overlays: selected A1 positions (packed x | y << 16) on the key array,
  // per-batch arrays on the index array.
  unsigned* s_sela = reinterpret_cast<unsigned*>(s_keys);   // up to MAX_N entries
  unsigned* b_xy = s_idx;                                   // [SEL_T]
  unsigned* b_stat = s_idx + SEL_T;                         // [SEL_T] 0 undecided, 1 selected, 2 dead
  unsigned* b_alist = s_idx + 2 * SEL_T;                    // [SEL_T] batch slots of live A1 entries
  if (tid == 0) {
    s_nsel = 0;
    s_nsela = 0;
    s_edge = 0xffffffffu;
  }
  __syncthreads();

  for (unsigned base = 0; base < M; base += SEL_T) {
    if (s_nsel >= (unsigned)N) break;
    const unsigned j = base + tid;
    const bool valid = j < M;
    unsigned e = valid ? idx_c[j] : 0u;
    const bool is_a = (e & 1u) != 0;
    const unsigned idx = e >> 1;
    const int py = (int)(idx / (unsigned)W), px = (int)(idx - (unsigned)py * (unsigned)W);
    unsigned stat = !valid ? 2u : (is_a ? 0u : 1u);
    if (tid == 0) s_nalist = 0;
    __syncthreads();
    if (valid && is_a) {
      const unsigned ns = s_nsela;
      for (unsigned q = 0; q < ns; ++q) {
        unsigned v = s_sela[q];
        int dx = (int)(v & 0xffffu) - px, dy = (int)(v >> 16) - py;
        if (dx <= r && dx >= -r && dy <= r && dy >= -r) {
          stat = 2u;
          break;
        }
      }
    }
    b_xy[tid] = (unsigned)px | ((unsigned)py << 16);
    b_stat[tid] = stat;
    __syncthreads();
    // ordered list of undecided A1 entries of this batch
    {
      unsigned long long m = __ballot(stat == 0u);
      unsigned lane = tid & 63, wv = tid >> 6;
      unsigned before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wsum[wv] = __popcll(m);
      __syncthreads();
      unsigned off = 0;
      for (unsigned w = 0; w < wv; ++w) off += s_wsum[w];
      if (stat == 0u) b_alist[off + before] = tid;
      if (tid == SEL_T - 1) s_nalist = off + before + (stat == 0u ? 1u : 0u);
      __syncthreads();
    }
    // resolve the undecided entries among themselves (rounds of the greedy rule)
    const unsigned na = s_nalist;
    for (;;) {
      if (tid == 0) s_flag = 0;
      __syncthreads();
      unsigned nstat = stat;
      if (stat == 0u) {
        bool blocked = false;
        for (unsigned q = 0; q < na; ++q) {
          unsigned o = b_alist[q];
          if (o >= (unsigned)tid) break;
          unsigned os = b_stat[o];
          if (os == 2u) continue;
          unsigned v = b_xy[o];
          int dx = (int)(v & 0xffffu) - px, dy = (int)(v >> 16) - py;
          if (dx <= r && dx >= -r && dy <= r && dy >= -r) {
            if (os == 1u) {
              nstat = 2u;
              blocked = false;
              break;
            }
            blocked = true;
          }
        }
        if (nstat == 0u && !blocked) nstat = 1u;
      }
      __syncthreads();
      if (nstat != stat) {
        stat = nstat;
        b_stat[tid] = stat;
      }
      if (stat == 0u) s_flag = 1;
      __syncthreads();
      if (!s_flag) break;
    }
    // ordered append of the selected entries
    {
      const bool selb = (stat == 1u);
      unsigned long long m = __ballot(selb);
      unsigned lane = tid & 63, wv = tid >> 6;
      unsigned before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wsum[wv] = __popcll(m);
      __syncthreads();
      unsigned off = 0, tot = 0;
      for (unsigned w = 0; w < SEL_T / 64; ++w) {
        if (w < wv) off += s_wsum[w];
        tot += s_wsum[w];
      }
      const unsigned n0 = s_nsel;
      if (selb) {
        unsigned pos = n0 + off + before;
        if (pos < (unsigned)N) {
          sel[pos] = idx;
          if (py < r || px < r) atomicMin(&s_edge, pos);   // reference: empty slice, no suppression
        }
        if (is_a) {
          unsigned q = atomicAdd(&s_nsela, 1u);
          if (q < (unsigned)MAX_N) s_sela[q] = (unsigned)px | ((unsigned)py << 16);
        }
      }
      __syncthreads();
      if (tid == 0) s_nsel = n0 + tot;
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- keypoints (x, y) as float64, with the reference's slicing corner cases ----
  const unsigned nsel = min(s_nsel, (unsigned)N);
  const unsigned edge = s_edge;
  for (unsigned i = tid; i < (unsigned)N; i += SEL_T) {
    double x = 0.0, y = 0.0;
    unsigned src = i;
    if (edge != 0xffffffffu && i > edge) src = edge;   // same pixel re-selected forever
    if (src < nsel) {
      unsigned idx = sel[src];
      unsigned yy = idx / (unsigned)W;
      y = (double)yy;
      x = (double)(idx - yy * (unsigned)W);
    }
    kp_xy[2 * i] = x;
    kp_xy[2 * i + 1] = y;
  }
  if (tid == 0) ctl->n_sel = nsel;
}

// ---------------------------------------------------------------------------------
// raw patch descriptors
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void patch_desc_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        const double* __restrict__ kp_xy, int N, int r,
                                                        double* __restrict__ desc) {
  const int k = blockIdx.x;
  const int d = 2 * r + 1;
  const int x = (int)kp_xy[2 * k], y = (int)kp_xy[2 * k + 1];
  double* o = desc + (size_t)k * d * d;
  for (int i = threadIdx.x; i < d * d; i += NT) {
    int dy = i / d, dx = i - dy * d;
    int gy = y - r + dy, gx = x - r + dx;
    uint8_t v = 0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(size_t)gy * W + gx];
    o[i] = (double)v;
  }
}

size_t response_lds_bytes(int p) {
  int pr = p >> 1;
  int GW = TX + 2 * pr, GH = TY + 2 * pr;
  int IW = GW + 2, IH = GH + 2;
  int IWp = (IW + 3) & ~3;
  return (size_t)((IWp * IH + 15) & ~15) + (size_t)GW * GH * 4 + (size_t)3 * GH * TX * 4;
}

size_t candidates_lds_bytes(int r) {
  int RW = TX + 4 * r, RH = TY + 4 * r;
  int LW = TX + 2 * r, LH = TY + 2 * r;
  return (size_t)RW * RH * 8 + (size_t)((LW * LH + 15) & ~15) + (size_t)TX * TY;
}

unsigned next_pow2(unsigned v) {
  unsigned p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" {

int vo_harris_response_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int patch, double kappa,
                           double* d_scores) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_img && d_scores, "harris_response: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "harris_response: bad image size %dx%d", W, H);
  VO_REQUIRE(ctx, patch >= 3 && patch <= 31 && (patch & 1), "harris_response: patch must be odd in 3..31");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  dim3 grid(vo_cdiv(W, TX), vo_cdiv(H, TY));
  size_t lds = response_lds_bytes(patch);
  {
    vo_prof_scope ps(ctx, VO_K_HARRIS_RESPONSE);
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(NT), lds, ctx->stream, d_img, H, W, patch, kappa,
                       d_scores);
  }
  return vo_check_launch(ctx, "harris_response_kernel");
}

int vo_nms_keypoints_dev(vo_ctx* ctx, const double* d_scores, int H, int W, int N, int r, double* d_kp_xy) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_scores && d_kp_xy, "nms: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "nms: bad map size %dx%d", W, H);
  VO_REQUIRE(ctx, W < 65536 && H < 65536, "nms: map side must be < 65536");
  VO_REQUIRE(ctx, N >= 1 && N <= MAX_N, "nms: N must be in 1..%d", MAX_N);
  VO_REQUIRE(ctx, r >= 0 && r <= 12, "nms: radius must be in 0..12");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const unsigned cap = (unsigned)H * (unsigned)W;
  const unsigned cap_c = next_pow2(2u * cap);
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_l1, (size_t)cap * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_l1, (size_t)cap * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_a1, (size_t)cap * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_a1, (size_t)cap * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_c, (size_t)cap_c * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_c, (size_t)cap_c * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_sel, (size_t)MAX_N * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_cand, (size_t)cap * 4));
  if (ctx->nms_alive.cap < (size_t)cap || ctx->nms_alive_dirty) {
    VO_TRY(vo_ensure(ctx, ctx->nms_alive, (size_t)cap));
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->nms_alive.p, 0, ctx->nms_alive.cap, ctx->stream));
    ctx->nms_alive_dirty = false;
  }
  if (!ctx->nms_hist.p) {
    VO_TRY(vo_ensure(ctx, ctx->nms_hist, (size_t)HIST_BINS * 4));
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->nms_hist.p, 0, (size_t)HIST_BINS * 4, ctx->stream));
  }
  VO_TRY(vo_ensure(ctx, ctx->nms_ctl, sizeof(nms_ctl)));
  nms_ctl* ctl = (nms_ctl*)ctx->nms_ctl.p;
  VO_HIP_TRY(ctx, hipMemsetAsync(ctl, 0, sizeof(nms_ctl), ctx->stream));

  dim3 grid(vo_cdiv(W, TX), vo_cdiv(H, TY));
  unsigned long long* keys_c = (unsigned long long*)ctx->nms_keys_c.p;
  unsigned* idx_c = (unsigned*)ctx->nms_idx_c.p;
  unsigned* cand = (unsigned*)ctx->nms_cand.p;
  uint8_t* alive = (uint8_t*)ctx->nms_alive.p;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_CANDIDATES);
    hipLaunchKernelGGL(nms_candidates_kernel, grid, dim3(NT), candidates_lds_bytes(r), ctx->stream, d_scores,
                       H, W, r, (unsigned long long*)ctx->nms_keys_l1.p, (unsigned*)ctx->nms_idx_l1.p,
                       (unsigned long long*)ctx->nms_keys_a1.p, (unsigned*)ctx->nms_idx_a1.p,
                       (unsigned*)ctx->nms_hist.p, ctl, cap);
  }
  VO_TRY(vo_check_launch(ctx, "nms_candidates_kernel"));
  {
    vo_prof_scope ps(ctx, VO_K_NMS_THRESHOLD);
    hipLaunchKernelGGL(nms_threshold_kernel, dim3(1), dim3(1024), 0, ctx->stream, (unsigned*)ctx->nms_hist.p,
                       ctl, N);
  }
  VO_TRY(vo_check_launch(ctx, "nms_threshold_kernel"));
  ctx->nms_alive_dirty = true;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_COMPACT);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(512), dim3(NT), 0, ctx->stream,
                       (const unsigned long long*)ctx->nms_keys_l1.p, (const unsigned*)ctx->nms_idx_l1.p,
                       (const unsigned long long*)ctx->nms_keys_a1.p, (const unsigned*)ctx->nms_idx_a1.p, keys_c,
                       idx_c, cand, alive, ctl, cap_c, cap);
  }
  VO_TRY(vo_check_launch(ctx, "nms_compact_kernel"));
  for (int round = 0; round < NMS_ROUNDS; ++round) {
    vo_prof_scope ps(ctx, VO_K_NMS_ROUND);
    hipLaunchKernelGGL(nms_round_kernel, dim3(512), dim3(NT), 0, ctx->stream, d_scores, alive, cand, keys_c, idx_c,
                       ctl, cap_c, cap, H, W, r);
  }
  VO_TRY(vo_check_launch(ctx, "nms_round_kernel"));
  {
    vo_prof_scope ps(ctx, VO_K_NMS_COLLECT);
    hipLaunchKernelGGL(nms_collect_kernel, dim3(512), dim3(NT), 0, ctx->stream, d_scores, alive, cand, keys_c, idx_c,
                       ctl, cap_c, cap);
  }
  VO_TRY(vo_check_launch(ctx, "nms_collect_kernel"));
  ctx->nms_alive_dirty = false;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_SELECT);
    hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(SEL_T), 0, ctx->stream, keys_c, idx_c, ctl, cap_c, W, N, r,
                       (unsigned*)ctx->nms_sel.p, d_kp_xy);
  }
  return vo_check_launch(ctx, "nms_select_kernel");
}

int vo_patch_descriptors_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, const double* d_kp_xy, int N,
                             int r, double* d_desc) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_img && d_kp_xy && d_desc, "patch_descriptors: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && N >= 0 && r >= 0 && r <= 64, "patch_descriptors: bad arguments");
  if (N == 0) return VO_OK;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    vo_prof_scope ps(ctx, VO_K_PATCH_DESC);
    hipLaunchKernelGGL(patch_desc_kernel, dim3(N), dim3(NT), 0, ctx->stream, d_img, H, W, d_kp_xy, N, r, d_desc);
  }
  return vo_check_launch(ctx, "patch_desc_kernel");
}

// ---- host-buffer wrappers ----------------------------------------------------------

static int upload_image(vo_ctx* ctx, const uint8_t* img, int H, int W) {
  size_t n = (size_t)H * W;
  VO_TRY(vo_ensure(ctx, ctx->img, n));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, img, n, hipMemcpyHostToDevice, ctx->stream));
  return VO_OK;
}

int vo_harris_response(vo_ctx* ctx, const uint8_t* img, int H, int W, int patch, double kappa, double* scores) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && scores, "harris_response: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0, "harris_response: bad image size");
  size_t n = (size_t)H * W;
  VO_TRY(upload_image(ctx, img, H, W));
  VO_TRY(vo_ensure(ctx, ctx->scores, n * 8));
  VO_TRY(vo_harris_response_dev(ctx, (const uint8_t*)ctx->img.p, H, W, patch, kappa, (double*)ctx->scores.p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(scores, ctx->scores.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

static int check_overflow(vo_ctx* ctx) {
  nms_ctl h;
  VO_HIP_TRY(ctx, hipMemcpyAsync(&h, ctx->nms_ctl.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (h.overflow) return vo_set_error(ctx, VO_ECAPACITY, "nms: candidate list overflow");
  return VO_OK;
}

int vo_harris_keypoints(vo_ctx* ctx, const uint8_t* img, int H, int W, int patch, double kappa, int N, int r,
                        double* kp_xy, double* scores) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && kp_xy, "harris_keypoints: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && N >= 1, "harris_keypoints: bad arguments");
  size_t n = (size_t)H * W;
  VO_TRY(upload_image(ctx, img, H, W));
  VO_TRY(vo_ensure(ctx, ctx->scores, n * 8));
  VO_TRY(vo_ensure(ctx, ctx->kp, (size_t)N * 16));
  VO_TRY(vo_harris_response_dev(ctx, (const uint8_t*)ctx->img.p, H, W, patch, kappa, (double*)ctx->scores.p));
  VO_TRY(vo_nms_keypoints_dev(ctx, (const double*)ctx->scores.p, H, W, N, r, (double*)ctx->kp.p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(kp_xy, ctx->kp.p, (size_t)N * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (scores) VO_HIP_TRY(ctx, hipMemcpyAsync(scores, ctx->scores.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  return check_overflow(ctx);
}

int vo_nms_keypoints(vo_ctx* ctx, const double* scores, int H, int W, int N, int r, double* kp_xy) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, scores && kp_xy, "nms_keypoints: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && N >= 1, "nms_keypoints: bad arguments");
  size_t n = (size_t)H * W;
  VO_TRY(vo_ensure(ctx, ctx->scores, n * 8));
  VO_TRY(vo_ensure(ctx, ctx->kp, (size_t)N * 16));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->scores.p, scores, n * 8, hipMemcpyHostToDevice, ctx->stream));
  VO_TRY(vo_nms_keypoints_dev(ctx, (const double*)ctx->scores.p, H, W, N, r, (double*)ctx->kp.p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(kp_xy, ctx->kp.p, (size_t)N * 16, hipMemcpyDeviceToHost, ctx->stream));
  return check_overflow(ctx);
}

int vo_patch_descriptors(vo_ctx* ctx, const uint8_t* img, int H, int W, const double* kp_xy, int N, int r,
                         double* desc) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, img && kp_xy && desc, "patch_descriptors: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && N >= 0 && r >= 0, "patch_descriptors: bad arguments");
  if (N == 0) return VO_OK;
  size_t dd = (size_t)(2 * r + 1) * (2 * r + 1);
  VO_TRY(upload_image(ctx, img, H, W));
  VO_TRY(vo_ensure(ctx, ctx->kp, (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, ctx->desc, (size_t)N * dd * 8));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->kp.p, kp_xy, (size_t)N * 16, hipMemcpyHostToDevice, ctx->stream));
  VO_TRY(vo_patch_descriptors_dev(ctx, (const uint8_t*)ctx->img.p, H, W, (const double*)ctx->kp.p, N, r,
                                  (double*)ctx->desc.p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(desc, ctx->desc.p, (size_t)N * dd * 8, hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

}  // extern "C"
