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

constexpr int NT = 256;  // threads per workgroup
constexpr int RX = 64;   // response tile width  (one lane per column)
constexpr int RY = 32;   // response tile height (four strips of eight rows)

// ---------------------------------------------------------------------------------
// Harris response
// ---------------------------------------------------------------------------------
// One workgroup per 64x32 tile.  All sums are integers (exact in any order):
//   A  image tile + halo -> LDS bytes          (all global loads issued before the first use)
//   B  Sobel pair, four pixels per work item   (word reads, one 16-byte store)
//   C  horizontal box sums of Ix^2, Iy^2, IxIy (one lane per column)
//   D  vertical box sums as running sums down a strip of rows, then the fp64 formula
//   (compile-time patch: C and D run twice, over the upper and the lower 16 output rows, so that the three planes
//    of horizontal sums hold 24 rows instead of 40: 33 KB of LDS per workgroup instead of 45, four workgroups per CU
//    instead of three -- the kernel is occupancy-bound on its write-out)
// P_T > 0: patch size known at compile time; P_T == 0: runtime patch size, plain loops.
struct resp_geom {
  int pr, GW, GWp, GH, IWp, IH;
};
typedef short pk16 __attribute__((ext_vector_type(2)));   // two 16-bit lanes of one register (packed math)
__device__ __forceinline__ pk16 as_pk16(unsigned v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ unsigned as_u32(pk16 v) { return __builtin_bit_cast(unsigned, v); }
__host__ __device__ inline resp_geom response_geometry(int p) {
  resp_geom g;
  g.pr = p >> 1;
  g.GW = RX + 2 * g.pr;
  g.GWp = (g.GW + 3) & ~3;      // gradient row pitch: whole groups of four
  g.GH = RY + 2 * g.pr;
  g.IWp = g.GWp + 4;            // image bytes per row (GWp + 2 used), word multiple
  g.IH = g.GH + 2;
  return g;
}

template <int P_T>
__global__ __launch_bounds__(NT) void harris_response_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                             int p_arg, double kappa,
                                                             double* __restrict__ out, size_t img_stride,
                                                             const int* __restrict__ go) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (go && !go[blockIdx.z]) return;                 // this sequence's detection is not needed for this frame
  img += (size_t)blockIdx.z * img_stride;            // several sequences per launch: grid.z = sequence
  out += (size_t)blockIdx.z * ((size_t)H * W);
  const int p = P_T > 0 ? P_T : p_arg;
  const resp_geom g = response_geometry(p);
  const int pr = g.pr, GWp = g.GWp, GH = g.GH, IWp = g.IWp, IH = g.IH;
  int* s_g = reinterpret_cast<int*>(smem);                              // packed (Ix | Iy << 16), GH x GWp
  const int HR = P_T > 0 ? RY / 2 + P_T - 1 : GH;                       // rows of horizontal sums held at a time
  int* s_hxx = s_g + GWp * GH;                                         // HR x RX each
  int* s_hyy = s_hxx + HR * RX;
  int* s_hxy = s_hyy + HR * RX;
  uint8_t* s_img = reinterpret_cast<uint8_t*>(s_hxx);   // the image bytes are dead once B has made s_g: the sums' planes
                                                        // take their place (29.3 KB per workgroup, five per CU, not four)

  const int tid = threadIdx.x;
  const unsigned tile = vo_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int x0 = (int)(tile % gridDim.x) * RX, y0 = (int)(tile / gridDim.x) * RY;
  const int ix0 = x0 - pr - 1, iy0 = y0 - pr - 1;   // image coords of s_img[0][0]

  // A: image tile + halo (zeros outside the image; such pixels only feed outputs
  //    that the border rule forces to 0)
  if (P_T > 0) {
    constexpr int PG = ((RX + 2 * (P_T >> 1) + 3) & ~3) + 4;      // = IWp
    constexpr int PH = RY + 2 * (P_T >> 1) + 2;                   // = IH
    if (ix0 >= 0 && iy0 >= 0 && ix0 + PG <= W && iy0 + PH <= H) {
      // interior tile: whole words (global addresses of any alignment), all in flight together
      constexpr int WPR = PG / 4, NWORD = WPR * PH, PERW = (NWORD + NT - 1) / NT;
      unsigned v[PERW];
#pragma unroll
      for (int k = 0; k < PERW; ++k) {
        const int i = min(tid + k * NT, NWORD - 1);
        const int ly = i / WPR, j = i - ly * WPR;
        __builtin_memcpy(&v[k], img + (size_t)(iy0 + ly) * W + ix0 + 4 * j, 4);
      }
#pragma unroll
      for (int k = 0; k < PERW; ++k) {
        const int i = tid + k * NT;
        if (i < NWORD) reinterpret_cast<unsigned*>(s_img)[i] = v[k];
      }
    } else {
      constexpr int PER = (PG * PH + NT - 1) / NT;
      uint8_t v[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * NT;
        const int ly = i / IWp, lx = i - ly * IWp;
        const int gy = iy0 + ly, gx = ix0 + lx;
        const bool in = i < IWp * IH && gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[k] = in ? img[(size_t)gy * W + gx] : (uint8_t)0;
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * NT;
        if (i < IWp * IH) s_img[i] = v[k];
      }
    }
  } else {
    for (int i = tid; i < IWp * IH; i += NT) {
      const int ly = i / IWp, lx = i - ly * IWp;
      const int gy = iy0 + ly, gx = ix0 + lx;
      uint8_t v = 0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(size_t)gy * W + gx];
      s_img[i] = v;
    }
  }
  __syncthreads();

  const int groups = GWp >> 2;
  if (P_T > 0) {
    // B (packed): Sobel as a TRUE convolution (kernel flipped: left minus right, top minus bottom) on
    // pairs of 16-bit lanes.  Per row the six bytes b0..b5 under four neighbouring outputs become three
    // pairs; c = r0 + 2 r1 + r2 (column smoothing), d = r0 - r2; Ix = c[k] - c[k+2], Iy = d[k] + 2 d[k+1] + d[k+2].
    // The two gradients go to separate 16-bit planes so that step C can use packed dot products.
    short* s_ix = reinterpret_cast<short*>(s_g);
    short* s_iy = s_ix + GWp * GH;
    for (int i = tid; i < GH * groups; i += NT) {
      const int ly = i / groups, j = i - ly * groups;
      const unsigned* r0 = reinterpret_cast<const unsigned*>(s_img + ly * IWp + 4 * j);   // row above centre
      const unsigned* r1 = r0 + (IWp >> 2);
      const unsigned* r2 = r1 + (IWp >> 2);
      pk16 c[3], d[3];
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        const unsigned w0 = h < 2 ? r0[0] : r0[1], w1 = h < 2 ? r1[0] : r1[1], w2 = h < 2 ? r2[0] : r2[1];
        const unsigned sel = h == 1 ? 0x0c030c02u : 0x0c010c00u;
        const pk16 a0 = as_pk16(__builtin_amdgcn_perm(0u, w0, sel));
        const pk16 a1 = as_pk16(__builtin_amdgcn_perm(0u, w1, sel));
        const pk16 a2 = as_pk16(__builtin_amdgcn_perm(0u, w2, sel));
        c[h] = a0 + a1 + a1 + a2;
        d[h] = a0 - a2;
      }
      const pk16 d12 = as_pk16(__builtin_amdgcn_alignbit(as_u32(d[1]), as_u32(d[0]), 16));
      const pk16 d34 = as_pk16(__builtin_amdgcn_alignbit(as_u32(d[2]), as_u32(d[1]), 16));
      const pk16 gx01 = c[0] - c[1], gx23 = c[1] - c[2];
      const pk16 gy01 = d[0] + d12 + d12 + d[1], gy23 = d[1] + d34 + d34 + d[2];
      *reinterpret_cast<uint2*>(s_ix + ly * GWp + 4 * j) = make_uint2(as_u32(gx01), as_u32(gx23));
      *reinterpret_cast<uint2*>(s_iy + ly * GWp + 4 * j) = make_uint2(as_u32(gy01), as_u32(gy23));
    }
    __syncthreads();

    // C (packed): horizontal box sums of Ix^2, Iy^2, Ix Iy for two neighbouring output columns per work
    // item.  Output 2q sums taps 2q .. 2q+P-1, output 2q+1 taps 2q+1 .. 2q+P: the (P-3)/2 aligned pairs
    // in the middle are shared, each output adds one more pair and one single tap.
    constexpr int NW = P_T > 0 ? (P_T + 1) / 2 : 2;   // words (pairs of columns) a work item reads per plane
    const int wpr = GWp >> 1;                   // words per gradient row
    for (int half = 0; half < 2; ++half) {
    for (int i = tid; i < HR * (RX / 2); i += NT) {
      const int lr = i / (RX / 2), q = i - lr * (RX / 2);
      const int ly = half * (RY / 2) + lr;                     // gradient row of this row of sums
      const unsigned* px = reinterpret_cast<const unsigned*>(s_ix) + ly * wpr + q;
      const unsigned* py = reinterpret_cast<const unsigned*>(s_iy) + ly * wpr + q;
      unsigned wx[NW], wy[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        wx[k] = px[k];
        wy[k] = py[k];
      }
      int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
      for (int k = 1; k < NW - 1; ++k) {
        sxx = __builtin_amdgcn_sdot2(as_pk16(wx[k]), as_pk16(wx[k]), sxx, false);
        syy = __builtin_amdgcn_sdot2(as_pk16(wy[k]), as_pk16(wy[k]), syy, false);
        sxy = __builtin_amdgcn_sdot2(as_pk16(wx[k]), as_pk16(wy[k]), sxy, false);
      }
      const pk16 x0 = as_pk16(wx[0]), y0p = as_pk16(wy[0]);
      const pk16 x0h = as_pk16(wx[0] & 0xffff0000u), y0h = as_pk16(wy[0] & 0xffff0000u);
      const pk16 xl = as_pk16(wx[NW - 1]), yl = as_pk16(wy[NW - 1]);
      const pk16 xll = as_pk16(wx[NW - 1] & 0x0000ffffu), yll = as_pk16(wy[NW - 1] & 0x0000ffffu);
      int axx = __builtin_amdgcn_sdot2(x0, x0, sxx, false);
      int ayy = __builtin_amdgcn_sdot2(y0p, y0p, syy, false);
      int axy = __builtin_amdgcn_sdot2(x0, y0p, sxy, false);
      axx = __builtin_amdgcn_sdot2(xll, xll, axx, false);
      ayy = __builtin_amdgcn_sdot2(yll, yll, ayy, false);
      axy = __builtin_amdgcn_sdot2(xll, yll, axy, false);
      int bxx = __builtin_amdgcn_sdot2(xl, xl, sxx, false);
      int byy = __builtin_amdgcn_sdot2(yl, yl, syy, false);
      int bxy = __builtin_amdgcn_sdot2(xl, yl, sxy, false);
      bxx = __builtin_amdgcn_sdot2(x0h, x0h, bxx, false);
      byy = __builtin_amdgcn_sdot2(y0h, y0h, byy, false);
      bxy = __builtin_amdgcn_sdot2(x0h, y0h, bxy, false);
      const int o = lr * RX + 2 * q;
      *reinterpret_cast<int2*>(s_hxx + o) = make_int2(axx, bxx);
      *reinterpret_cast<int2*>(s_hyy + o) = make_int2(ayy, byy);
      *reinterpret_cast<int2*>(s_hxy + o) = make_int2(axy, bxy);
    }
    __syncthreads();
    // D (packed), this half: vertical sums as running sums down a strip of four rows, then the fp64 formula
    {
      const int lx = tid & (RX - 1);
      const int border = pr + 1;
      const int gx = x0 + lx;
      const int ys = (tid / RX) * 4;               // strip of four output rows of this half
      int vxx[4 + P_T - 1], vyy[4 + P_T - 1], vxy[4 + P_T - 1];
#pragma unroll
      for (int k = 0; k < 4 + P_T - 1; ++k) {
        const int j = (ys + k) * RX + lx;
        vxx[k] = s_hxx[j];
        vyy[k] = s_hyy[j];
        vxy[k] = s_hxy[j];
      }
      int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
      for (int k = 0; k < P_T; ++k) {
        sxx += vxx[k];
        syy += vyy[k];
        sxy += vxy[k];
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int gy = y0 + half * (RY / 2) + ys + o;
        if (gy < H && gx < W) {
          double r = 0.0;
          if (gy >= border && gy < H - border && gx >= border && gx < W - border) {
            const double dxx = (double)sxx, dyy = (double)syy, dxy = (double)sxy;
            const double trace = dxx + dyy;
            const double det = dxx * dyy - dxy * dxy;
            r = det - kappa * (trace * trace);
            if (r < 0) r = 0;
          }
          __builtin_nontemporal_store(r, &out[(size_t)gy * W + gx]);   // (written once, read by the next kernel: 3 % faster)
        }
        if (o < 3) {
          sxx += vxx[o + P_T] - vxx[o];
          syy += vyy[o + P_T] - vyy[o];
          sxy += vxy[o + P_T] - vxy[o];
        }
      }
    }
    __syncthreads();                               // (the sums' planes are free for the other half)
    }
    return;
  } else {
    // B: Sobel as a TRUE convolution (kernel flipped): left minus right, top minus bottom
    for (int i = tid; i < GH * groups; i += NT) {
      const int ly = i / groups, j = i - ly * groups;
      const unsigned* r0 = reinterpret_cast<const unsigned*>(s_img + ly * IWp + 4 * j);   // row above centre
      const unsigned* r1 = r0 + (IWp >> 2);
      const unsigned* r2 = r1 + (IWp >> 2);
      const unsigned long long a0 = r0[0] | ((unsigned long long)r0[1] << 32);
      const unsigned long long a1 = r1[0] | ((unsigned long long)r1[1] << 32);
      const unsigned long long a2 = r2[0] | ((unsigned long long)r2[1] << 32);
      int o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a00 = (int)((a0 >> (8 * k)) & 255), a01 = (int)((a0 >> (8 * k + 8)) & 255),
                  a02 = (int)((a0 >> (8 * k + 16)) & 255);
        const int a10 = (int)((a1 >> (8 * k)) & 255), a12 = (int)((a1 >> (8 * k + 16)) & 255);
        const int a20 = (int)((a2 >> (8 * k)) & 255), a21 = (int)((a2 >> (8 * k + 8)) & 255),
                  a22 = (int)((a2 >> (8 * k + 16)) & 255);
        const int gx = (a00 - a02) + 2 * (a10 - a12) + (a20 - a22);
        const int gy = (a00 - a20) + 2 * (a01 - a21) + (a02 - a22);
        o[k] = (gx & 0xffff) | (gy << 16);
      }
      *reinterpret_cast<int4*>(s_g + ly * GWp + 4 * j) = make_int4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // C: horizontal box sums of the three products
    for (int i = tid; i < GH * RX; i += NT) {
      const int ly = i / RX, lx = i - ly * RX;
      const int* gp = s_g + ly * GWp + lx;
      int sxx = 0, syy = 0, sxy = 0;
      for (int k = 0; k < p; ++k) {
        const int v = gp[k];
        const int gx = (int)(short)(v & 0xffff), gy = v >> 16;
        sxx += gx * gx;
        syy += gy * gy;
        sxy += gx * gy;
      }
      s_hxx[i] = sxx;
      s_hyy[i] = syy;
      s_hxy[i] = sxy;
    }
    __syncthreads();
  }

  // D (run-time patch size): vertical sums + response
  const int lx = tid & (RX - 1);
  const int border = pr + 1;
  const int gx = x0 + lx;
  {
    for (int ly = tid / RX; ly < RY; ly += NT / RX) {
      const int gy = y0 + ly;
      if (gy >= H || gx >= W) continue;
      double r = 0.0;
      if (gy >= border && gy < H - border && gx >= border && gx < W - border) {
        int sxx = 0, syy = 0, sxy = 0;
        for (int k = 0; k < p; ++k) {
          const int j = (ly + k) * RX + lx;
          sxx += s_hxx[j];
          syy += s_hyy[j];
          sxy += s_hxy[j];
        }
        const double dxx = (double)sxx, dyy = (double)syy, dxy = (double)sxy;
        const double trace = dxx + dyy;
        const double det = dxx * dyy - dxy * dxy;
        r = det - kappa * (trace * trace);
        if (r < 0) r = 0;
      }
      out[(size_t)gy * W + gx] = r;
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 1: per-tile L1 / A1 lists + histogram of L1 scores
// ---------------------------------------------------------------------------------
struct nms_ctl {
  unsigned n_c, n_rem, overflow, n_sel;
  unsigned long long t_bits;
  unsigned pad[2];
};

// Several sequences per launch: every buffer of the chain holds S consecutive per-sequence blocks, the
// sequence is the grid's extra dimension (z for the tiled kernels, y for the 1-D ones) and each kernel moves
// its pointers to its block first.  Strides in elements of the respective array; all zero for one sequence.
struct nms_batch {
  size_t sc = 0;        // score map (doubles)
  size_t seg = 0;       // per-tile list segments (keys_l1/a1, idx_l1/a1, cand)
  size_t segcnt = 0;    // per-tile counters
  size_t comp = 0;      // compacted lists (keys_c, idx_c)
  size_t alive = 0;     // state map
  size_t hist = 0;      // one histogram
  size_t rank = 0, sel = 0, kp = 0;
  const int* go = nullptr;   // one word per sequence; 0: the sequence sits this call out (every kernel returns at once)
};

constexpr int HIST_SHIFT = 47;           // 65536 bins over non-negative doubles
constexpr int HIST_BINS = 1 << 16;
constexpr int HIST_TOTAL = HIST_BINS + 256;   // fine bins, then 256 coarse bins (fine >> 8)
constexpr int CX = 64, CY = 32;          // candidate tile
constexpr int SEG = CX * CY;             // capacity of one tile's list segment
constexpr int RANK_MAX = 32768;          // entries the rank kernel orders (above: sort path)

// State word of a pixel in the candidate map: 0 = dead / not a candidate, 1 = selected,
// >= 3 = live, holding the top 32 bits of its score (a monotone 32-bit view of the
// priority: a larger word means a larger score, equal words need the full compare).
__device__ __forceinline__ unsigned live_code(unsigned long long key) {
  const unsigned hi = (unsigned)(key >> 32);
  return hi < 3u ? 3u : hi;
}

// compacting append inside a workgroup: one LDS atomic per wave
__device__ __forceinline__ unsigned wave_slot(bool take, unsigned* counter) {
  const unsigned long long m = __ballot(take);
  if (m == 0ull) return 0u;
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(m);
  unsigned base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
  base = __shfl(base, leader);
  return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}

// 32-bit view of a score for the window maxima: 0 for scores that are not positive, else the
// top half of the double (monotone; equal words are settled on the full values).
__device__ __forceinline__ unsigned score_word(double v) {
  const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
  return v > 0.0 ? (hi ? hi : 1u) : 0u;
}

// Sliding maxima over a window of WN entries: doubling makes every entry the maximum of the SP = 2^k <= WN entries
// starting at it, and two of those (overlapping) cover a window -- 6.4 operations per result for WN = 11 and eight
// results, against 10 for the plain loop.
__host__ __device__ constexpr int window_span(int wn) {
  int s = 1;
  while (2 * s <= wn) s *= 2;
  return s;
}
template <int LEN, int SP>
__device__ __forceinline__ void window_doubling(unsigned* v) {
#pragma unroll
  for (int s = 1; s < SP; s *= 2) {
#pragma unroll
    for (int k = 0; k + s < LEN; ++k) v[k] = max(v[k], v[k + s]);   // (ascending: v[k + s] is still the previous level's)
  }
}

// One workgroup per 64x32 tile; L1 flags are needed on the tile + r halo, hence scores on
// the tile + 2r halo.
//   A  score words of the region -> LDS          (all global loads issued before the first use)
//   B  column maxima, running down strips of eight rows (one lane per column)
//   C  row maxima of those = window maxima; a pixel whose word equals it is a "qualifier"
//   D  one wave per qualifier: neighbours with the same word are compared on the full score
//      and, for equal scores, on flat order -> L1 bit rows
//   E  tile pixels: L1 -> segment 0 (+ histogram); positive and not within r of any L1 -> segment 1
// R_T > 0: radius known at compile time (window loops unroll); R_T == 0: runtime radius.
template <int R_T>
__global__ __launch_bounds__(NT) void nms_candidates_kernel(const double* __restrict__ sc, int H, int W, int r_arg,
                                                            unsigned long long* __restrict__ seg_keys_l1,
                                                            unsigned* __restrict__ seg_idx_l1,
                                                            unsigned long long* __restrict__ seg_keys_a1,
                                                            unsigned* __restrict__ seg_idx_a1,
                                                            uint4* __restrict__ seg_cnt,
                                                            unsigned* __restrict__ hist, nms_ctl* ctl, nms_batch B) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (B.go && !B.go[blockIdx.z]) return;
  {
    const size_t q = blockIdx.z;
    sc += q * B.sc;
    seg_keys_l1 += q * B.seg;
    seg_idx_l1 += q * B.seg;
    seg_keys_a1 += q * B.seg;
    seg_idx_a1 += q * B.seg;
    seg_cnt += q * B.segcnt;
    hist += q * B.hist;
    ctl += q;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // counters of this call (first kernel of the chain)
    ctl->n_c = 0;
    ctl->n_rem = 0;
    ctl->overflow = 0;
    ctl->n_sel = 0;
  }
  const int r = R_T > 0 ? R_T : r_arg;
  const int RW = CX + 4 * r, RH = CY + 4 * r;   // score region (2r halo)
  const int LW = CX + 2 * r, LH = CY + 2 * r;   // region where L1 flags are needed
  const int WN = 2 * r + 1;
  unsigned* s_w = reinterpret_cast<unsigned*>(smem);   // RH x RW score words
  unsigned* s_cm = s_w + RW * RH;                      // LH x RW column maxima
  unsigned* s_mask = s_cm + RW * LH;                   // LH rows x 4 words (LW <= 128 bits)
  unsigned* s_comb = s_mask + LH * 4;                  // CY rows x 4 words
  unsigned short* s_list = reinterpret_cast<unsigned short*>(s_comb + CY * 4);   // qualifiers (L cells)
  __shared__ unsigned s_cnt[3];
  __shared__ unsigned s_coarse[256];   // the tile's share of the coarse histogram level

  const int tid = threadIdx.x;
  const unsigned blk = vo_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);   // (tile = segment index)
  const int x0 = (int)(blk % gridDim.x) * CX, y0 = (int)(blk / gridDim.x) * CY;
  if (tid < 3) s_cnt[tid] = 0;
  s_coarse[tid] = 0;

  // the tile's own scores stay in registers for step E
  const int lx = tid & (CX - 1);
  double own[SEG / NT];
#pragma unroll
  for (int k = 0; k < SEG / NT; ++k) {
    const int gy = y0 + tid / CX + k * (NT / CX), gx = x0 + lx;
    const bool in = gy < H && gx < W;
    own[k] = in ? sc[(size_t)(in ? gy : 0) * W + (in ? gx : 0)] : 0.0;
  }
  // A
  if (R_T > 0) {
    constexpr int PER = ((CX + 4 * R_T) * (CY + 4 * R_T) + NT - 1) / NT;
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * NT;
      const int ly = i / RW, lxx = i - ly * RW;
      const int gy = y0 - 2 * r + ly, gx = x0 - 2 * r + lxx;
      const bool in = i < RW * RH && gy >= 0 && gy < H && gx >= 0 && gx < W;
      v[k] = in ? sc[(size_t)(in ? gy : 0) * W + (in ? gx : 0)] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * NT;
      if (i < RW * RH) s_w[i] = score_word(v[k]);
    }
  } else {
    for (int i = tid; i < RW * RH; i += NT) {
      const int ly = i / RW, lxx = i - ly * RW;
      const int gy = y0 - 2 * r + ly, gx = x0 - 2 * r + lxx;
      double v = 0.0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = sc[(size_t)gy * W + gx];
      s_w[i] = score_word(v);
    }
  }
  for (int i = tid; i < LH * 4; i += NT) s_mask[i] = 0;
  __syncthreads();

  // B: s_cm[ly][x] = max of s_w[ly .. ly + 2r][x]
  if (R_T > 0) {
    const int strips = (LH + 7) / 8;
    for (int it = tid; it < RW * strips; it += NT) {
      const int st = it / RW, x = it - st * RW;
      const int ys = st * 8;
      unsigned v[8 + 2 * R_T];
#pragma unroll
      for (int k = 0; k < 8 + 2 * R_T; ++k) v[k] = ys + k < RH ? s_w[(ys + k) * RW + x] : 0u;
      constexpr int SP = window_span(2 * R_T + 1);
      window_doubling<8 + 2 * R_T, SP>(v);           // v[k] = max of entries k .. k + SP - 1
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (ys + o < LH) s_cm[(ys + o) * RW + x] = max(v[o], v[o + 2 * R_T + 1 - SP]);
    }
  } else {
    for (int i = tid; i < RW * LH; i += NT) {
      const int ly = i / RW, x = i - ly * RW;
      const unsigned* col = s_w + ly * RW + x;
      unsigned m = col[0];
      for (int d = 1; d < WN; ++d) m = max(m, col[d * RW]);
      s_cm[i] = m;
    }
  }
  __syncthreads();

  // C: window maximum = max of s_cm[ly][lx .. lx + 2r]
  if (R_T > 0 && (CX + 4 * R_T) % 4 == 0) {
    // eight neighbouring cells per work item: 18 words of the row in five wide reads, maxima by doubling.  (The last
    // group of a row reads past the row's end: those words only reach cells >= LW, which are not kept.)
    constexpr int G = (CX + 2 * R_T + 7) / 8, LEN = 8 + 2 * R_T, SP = window_span(2 * R_T + 1);
    static_assert(LEN <= 20, "five reads of four words");
    for (int it = tid; it < LH * G; it += NT) {
      const int ly = it / G, j = it - ly * G;
      unsigned v[20];
      const uint4* src = reinterpret_cast<const uint4*>(s_cm + ly * RW + 8 * j);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const uint4 q4 = src[k];
        v[4 * k] = q4.x;
        v[4 * k + 1] = q4.y;
        v[4 * k + 2] = q4.z;
        v[4 * k + 3] = q4.w;
      }
      window_doubling<LEN, SP>(v);
      const unsigned* cwp = s_w + (ly + R_T) * RW + 8 * j + R_T;
      unsigned qual = 0;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const unsigned m = max(v[o], v[o + 2 * R_T + 1 - SP]);
        const unsigned cw = cwp[o];
        if (8 * j + o < LW && cw != 0u && cw == m) qual |= 1u << o;
      }
      if (qual) {
        unsigned slot = atomicAdd(&s_cnt[2], (unsigned)__popc(qual));
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (qual & (1u << o)) s_list[slot++] = (unsigned short)(ly * LW + 8 * j + o);
      }
    }
  } else
  for (int b0 = 0; b0 < LW * LH; b0 += NT) {
    const int i = b0 + tid;
    bool q = false;
    if (i < LW * LH) {
      const int ly = i / LW, lxx = i - ly * LW;
      const unsigned* row = s_cm + ly * RW + lxx;
      unsigned m = row[0];
      if (R_T > 0) {
#pragma unroll
        for (int d = 1; d <= 2 * R_T; ++d) m = max(m, row[d]);
      } else {
        for (int d = 1; d < WN; ++d) m = max(m, row[d]);
      }
      const unsigned cw = s_w[(ly + r) * RW + (lxx + r)];
      q = cw != 0u && cw == m;
    }
    const unsigned slot = wave_slot(q, &s_cnt[2]);
    if (q) s_list[slot] = (unsigned short)i;
  }
  __syncthreads();

  // D: one wave per qualifier
  {
    const unsigned nq = s_cnt[2];
    const int lane = tid & 63;
    for (unsigned k = tid >> 6; k < nq; k += NT / 64) {
      const int cell = s_list[k];
      const int ly = cell / LW, lxx = cell - ly * LW;
      const unsigned* c = s_w + (ly + r) * RW + (lxx + r);
      const unsigned cw = *c;
      const int py = y0 - r + ly, px = x0 - r + lxx;     // inside the image: its word is not 0
      bool beaten = false;
      for (int t = lane; t < WN * WN; t += 64) {
        const int j = t / WN - r, d = t - (t / WN) * WN - r;
        if (j == 0 && d == 0) continue;
        if (c[j * RW + d] == cw) {
          const double s = sc[(size_t)py * W + px];
          const double q = sc[(size_t)(py + j) * W + (px + d)];
          const bool before = (j < 0) || (j == 0 && d < 0);
          if (before ? (q >= s) : (q > s)) beaten = true;
        }
      }
      if (__ballot(beaten) == 0ull && lane == 0) atomicOr(&s_mask[ly * 4 + (lxx >> 5)], 1u << (lxx & 31));
    }
  }
  __syncthreads();

  // L1 flags are bit rows (4 words per L row); a tile row's cover mask is the OR of 2r+1 of them
  for (int i = tid; i < CY * 4; i += NT) {
    const int ly = i >> 2, w = i & 3;
    unsigned m = 0;
    for (int d = 0; d < WN; ++d) m |= s_mask[(ly + d) * 4 + w];
    s_comb[i] = m;
  }
  __syncthreads();

  // E: classify the tile's own pixels and append them to the tile's segments
  const size_t seg0 = (size_t)blk * SEG;
#pragma unroll
  for (int k = 0; k < SEG / NT; ++k) {
    const int ly = tid / CX + k * (NT / CX);
    const int gy = y0 + ly, gx = x0 + lx;
    int kind = -1;
    const double sv = own[k];
    if (gy < H && gx < W && sv > 0.0) {
      const int bx = lx + r;                                   // own bit in the L row
      const bool is_l1 = (s_mask[(ly + r) * 4 + (bx >> 5)] >> (bx & 31)) & 1u;
      // any L1 bit in columns lx .. lx + 2r of the combined row?
      const unsigned* cw = s_comb + ly * 4;
      const unsigned long long lo = cw[0] | ((unsigned long long)cw[1] << 32);
      const unsigned long long hi = cw[2] | ((unsigned long long)cw[3] << 32);
      const unsigned long long win = lx == 0 ? lo : ((lo >> lx) | (hi << (64 - lx)));
      const bool covered = (win & ((1ull << WN) - 1ull)) != 0ull;
      if (is_l1) kind = 0;
      else if (!covered) kind = 1;
    }
    const unsigned long long key = (unsigned long long)__double_as_longlong(sv);
    const unsigned idx = (unsigned)gy * (unsigned)W + (unsigned)gx;
    const unsigned s0 = wave_slot(kind == 0, &s_cnt[0]);
    const unsigned s1 = wave_slot(kind == 1, &s_cnt[1]);
    if (kind == 0) {
      seg_keys_l1[seg0 + s0] = key;
      seg_idx_l1[seg0 + s0] = idx;
      const unsigned bin = (unsigned)(key >> HIST_SHIFT);
      atomicAdd(&hist[bin], 1u);
      atomicAdd(&s_coarse[bin >> 8], 1u);   // coarse level (256 fine bins each): the scores of a frame share
                                            // a few of these bins, so they are summed per tile first
    } else if (kind == 1) {
      seg_keys_a1[seg0 + s1] = key;
      seg_idx_a1[seg0 + s1] = idx;
    }
  }
  __syncthreads();
  if (tid == 0) seg_cnt[blk] = make_uint4(s_cnt[0], s_cnt[1], 0u, 0u);
  if (s_coarse[tid]) atomicAdd(&hist[HIST_BINS + tid], s_coarse[tid]);
}

// ---------------------------------------------------------------------------------
// NMS stage 2: score bound with >= N L1 entries above it (two-level histogram walk)
// ---------------------------------------------------------------------------------
// suffix sums over the 256 values held one per thread: returns sum of v over threads >= tid
__device__ __forceinline__ unsigned suffix_sum_256(unsigned v, unsigned* s_w) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned suf = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_down(suf, off);
    if (lane + off < 64) suf += o;
  }
  __syncthreads();
  if (lane == 0) s_w[wv] = suf;
  __syncthreads();
  for (int w = wv + 1; w < 4; ++w) suf += s_w[w];
  return suf;
}

// The bound as IEEE bits (every thread of the 256-thread workgroup gets it); each compact
// workgroup computes it for itself rather than waiting for a one-workgroup kernel.
__device__ __forceinline__ unsigned long long nms_threshold_bits(const unsigned* __restrict__ hist, int N) {
  __shared__ unsigned s_w[4];
  __shared__ unsigned s_above;
  const int tid = threadIdx.x;
  // coarse level: thread t holds the count of fine bins [256 t, 256 t + 256)
  const unsigned v1 = hist[HIST_BINS + tid];
  const unsigned suf1 = suffix_sum_256(v1, s_w);
  const int n_ge = __syncthreads_count(suf1 >= (unsigned)N);   // suffix sums do not increase with t
  if (n_ge == 0) return 1ull;           // fewer than N strict maxima: keep everything
  const int cb = n_ge - 1;              // crossing coarse bin
  if (tid == cb) s_above = suf1 - v1;   // strict maxima in coarse bins above it
  __syncthreads();
  const unsigned above = s_above;
  const unsigned v2 = hist[cb * 256 + tid];
  const unsigned suf2 = suffix_sum_256(v2, s_w);
  const int m_ge = __syncthreads_count(above + suf2 >= (unsigned)N);   // >= 1: bin 0 of cb reaches suf1(cb) >= N
  const unsigned long long t = (unsigned long long)(cb * 256 + (m_ge - 1)) << HIST_SHIFT;
  return t ? t : 1ull;
}

// List entry format (idx_c): flat pixel index << 2 | flags.  bit 0: undecided (the sorted walk
// must still test it); bit 1: came from the candidate segments (a selected one suppresses
// later undecided entries; strict maxima never do, no candidate lies within r of one).
constexpr unsigned ENT_UNDECIDED = 1u, ENT_CAND = 2u;

// ---------------------------------------------------------------------------------
// NMS stage 3 (one workgroup per tile): L1 >= T -> selected list; A1 >= T -> live candidates.
// Also clears the other call parity's histogram (all its readers ran during the previous call).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void nms_compact_kernel(const unsigned long long* __restrict__ seg_keys_l1,
                                                         const unsigned* __restrict__ seg_idx_l1,
                                                         const unsigned long long* __restrict__ seg_keys_a1,
                                                         const unsigned* __restrict__ seg_idx_a1,
                                                         uint4* __restrict__ seg_cnt, unsigned* __restrict__ seg_cand,
                                                         unsigned long long* __restrict__ keys_c,
                                                         unsigned* __restrict__ idx_c, unsigned* __restrict__ alive,
                                                         nms_ctl* ctl, unsigned cap_c,
                                                         const unsigned* __restrict__ hist,
                                                         unsigned* __restrict__ hist_other, int N, nms_batch B) {
  {
    const size_t q = blockIdx.y;
    seg_keys_l1 += q * B.seg;
    seg_idx_l1 += q * B.seg;
    seg_keys_a1 += q * B.seg;
    seg_idx_a1 += q * B.seg;
    seg_cnt += q * B.segcnt;
    seg_cand += q * B.seg;
    keys_c += q * B.comp;
    idx_c += q * B.comp;
    alive += q * B.alive;
    ctl += q;
    hist += q * B.hist;
    hist_other += q * B.hist;
  }
  __shared__ unsigned s_n, s_l1n, s_base;
  static_assert(NT == 256, "nms_threshold_bits is written for 256 threads");
  const unsigned blk = blockIdx.x;
  const size_t seg0 = (size_t)blk * SEG;
  const uint4 cnt = seg_cnt[blk];
  const int tid = threadIdx.x;
  for (unsigned i = blk * NT + tid; i < (unsigned)HIST_TOTAL; i += gridDim.x * NT) hist_other[i] = 0;
  if (B.go && !B.go[blockIdx.y]) return;      // (a sequence that sits the call out still keeps the histogram rotation)
  // the tile's segment entries are requested before the bound is computed (its two dependent
  // histogram reads then overlap these loads instead of preceding one round trip per chunk)
  constexpr int CH = SEG / NT;
  unsigned long long ka[CH];
  unsigned ia[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const unsigned i = (unsigned)k * NT + tid;
    const bool in = i < cnt.y;
    ka[k] = in ? seg_keys_a1[seg0 + i] : 0ull;
    ia[k] = in ? seg_idx_a1[seg0 + i] : 0u;
  }
  unsigned long long kl = tid < cnt.x ? seg_keys_l1[seg0 + tid] : 0ull;
  unsigned il = tid < cnt.x ? seg_idx_l1[seg0 + tid] : 0u;
  const unsigned long long t = nms_threshold_bits(hist, N);
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (unsigned b0 = 0; b0 < cnt.x; b0 += NT) {
    const unsigned i = b0 + tid;
    if (b0 > 0) {   // tiles with more than NT strict maxima (r = 0 ...): later chunks are fetched here
      kl = i < cnt.x ? seg_keys_l1[seg0 + i] : 0ull;
      il = i < cnt.x ? seg_idx_l1[seg0 + i] : 0u;
    }
    const bool keep = i < cnt.x && kl >= t;
    if (tid == 0) s_l1n = 0;
    __syncthreads();
    const unsigned slot = wave_slot(keep, &s_l1n);
    __syncthreads();
    if (tid == 0 && s_l1n) s_base = atomicAdd(&ctl->n_c, s_l1n);   // one list reservation per tile and chunk
    __syncthreads();
    if (keep) {
      const unsigned pos = s_base + slot;
      if (pos < cap_c) {
        keys_c[pos] = kl;
        idx_c[pos] = il << 2;
      } else {
        ctl->overflow = 1;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    if ((unsigned)k * NT >= cnt.y) break;                 // uniform
    const bool keep = (unsigned)k * NT + tid < cnt.y && ka[k] >= t;
    const unsigned slot = wave_slot(keep, &s_n);          // (chunks in order, waves race within a chunk: any order is fine)
    if (keep) {
      seg_cand[seg0 + slot] = ia[k];
      alive[ia[k]] = live_code(ka[k]);
    }
  }
  __syncthreads();
  if (tid == 0) {
    seg_cnt[blk].z = s_n;   // candidates of this tile
    seg_cnt[blk].w = s_n;   // of which still live
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 3b: the greedy rule among the candidates
// ---------------------------------------------------------------------------------
// Parallel rounds of the greedy rule, one workgroup per tile.  Each iteration the tile
//   0. re-reads its part of the global state map (tile + r halo) into LDS -- the map is only
//      ever written with agent-scope stores, so selections and kills made by neighbouring
//      tiles during this same launch become visible here without another launch;
//   1. M = (2r+1)^2 window maximum of the state words (separable, dense over the tile);
//   2. a live candidate whose word equals M has no live neighbour with a larger 32-bit view
//      of the score; one wave scans its window: a selected word in it means the candidate was
//      killed (its cell is cleared); equal words are settled by the full score and the index;
//      if it still stands it is selected: own word := 1, live words of its window := 0, in
//      LDS and (agent scope) in the global map.
// State only moves live -> selected | dead, so a word that is stale for an iteration only
// postpones a decision; so step 0 is skipped while the tile is still deciding things on its
// own.  A tile leaves when none of its candidates is live; tiles blocked on a neighbour poll
// for a bounded number of iterations, what is left goes to the second launch, and what that
// leaves (FINAL) joins the list as undecided entries for the sorted walk.
constexpr int ROUND_ITERS = 64;
constexpr int RT = 512;           // threads of a round workgroup
constexpr int LIVE_SPARSE = 32;   // at most this many live candidates: no dense pass, each looks around itself
static_assert((CX * CY) % RT == 0 && RT % CY == 0, "the row pass of the round kernel maps one strip to one thread");

__device__ __forceinline__ unsigned load_state(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_state(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int R_T, bool FINAL>
__global__ __launch_bounds__(RT) void nms_round_kernel(const double* __restrict__ sc, unsigned* alive,
                                                       uint4* __restrict__ seg_cnt,
                                                       const unsigned* __restrict__ seg_cand,
                                                       unsigned long long* __restrict__ keys_c,
                                                       unsigned* __restrict__ idx_c, nms_ctl* ctl, unsigned cap_c,
                                                       int H, int W, int r_arg, int tiles_x, nms_batch B) {
  extern __shared__ __align__(16) unsigned s_dyn[];
  if (B.go && !B.go[blockIdx.y]) return;
  {
    const size_t q = blockIdx.y;
    sc += q * B.sc;
    alive += q * B.alive;
    seg_cnt += q * B.segcnt;
    seg_cand += q * B.seg;
    keys_c += q * B.comp;
    idx_c += q * B.comp;
    ctl += q;
  }
  const int r = R_T > 0 ? R_T : r_arg;
  const int WN = 2 * r + 1;
  const int LW = CX + 2 * r, LH = CY + 2 * r;
  // odd row pitches: a wave's lanes walk either along x (stride 1) or along y (stride = pitch),
  // and both patterns then spread over all LDS banks
  const int SP = LW | 1;                         // pitch of the state snapshot and of the column maxima
  constexpr int PM = CX + 1;                     // pitch of the window maxima
  unsigned* s_state = s_dyn;                     // LH x LW (pitch SP)
  unsigned* s_v = s_state + SP * LH;             // CY x LW (pitch SP)  column maxima
  unsigned* s_m = s_v;                           // CY x CX (pitch PM)  window maxima, written over the column maxima
  __shared__ unsigned short s_cell[SEG];         // LDS cell of candidate i
  __shared__ unsigned short s_pass[SEG];         // live candidates of this iteration
  __shared__ unsigned short s_sel[SEG];          // candidates selected in this launch
  __shared__ unsigned short s_pass2[SEG];        // ... after the dense filter
  __shared__ unsigned s_npass, s_npass2, s_nsel, s_prog;
  // (tile = dispatch order here, not vo_xcd_tile's bands: a tile polls its neighbours' decisions, and with the bands the
  //  first rows of a band run long before the last rows of the band above them -- at 3840x2160, where the tiles are not
  //  all resident, those rows gave up undecided and the sorted walk of the leftovers took 3.5 ms)
  const unsigned blk = blockIdx.x;
  const unsigned n = seg_cnt[blk].z;
  const int tid = threadIdx.x;
  if (n == 0 || seg_cnt[blk].w == 0) return;
  const int x0 = (int)(blk % (unsigned)tiles_x) * CX, y0 = (int)(blk / (unsigned)tiles_x) * CY;
  const size_t seg0 = (size_t)blk * SEG;
  for (unsigned i = tid; i < n; i += RT) {
    const unsigned idx = seg_cand[seg0 + i];
    const int py = (int)(idx / (unsigned)W), px = (int)(idx - (unsigned)py * (unsigned)W);
    s_cell[i] = (unsigned short)((py - y0 + r) * SP + (px - x0 + r));
  }
  if (tid == 0) {
    s_nsel = 0;
    s_prog = 0;
  }
  __syncthreads();
  bool reload = true;
  for (int iter = 0; iter < ROUND_ITERS; ++iter) {
    // ---- fresh copy of the tile's part of the state map ----
    if (!reload) {
    } else if (R_T > 0) {
      constexpr int PER = ((CX + 2 * R_T) * (CY + 2 * R_T) + RT - 1) / RT;
      unsigned v[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * RT;
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = y0 - r + ly, gx = x0 - r + lx;
        const bool in = i < LW * LH && gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[k] = in ? load_state(alive + (size_t)(in ? gy : 0) * W + (in ? gx : 0)) : 0u;
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * RT;
        const int ly = i / LW, lx = i - ly * LW;
        if (i < LW * LH) s_state[ly * SP + lx] = v[k];
      }
    } else {
      for (int i = tid; i < LW * LH; i += RT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = y0 - r + ly, gx = x0 - r + lx;
        unsigned v = 0u;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = load_state(alive + (size_t)gy * W + gx);
        s_state[ly * SP + lx] = v;
      }
    }
    __syncthreads();
    // ---- the tile's live candidates ----
    if (tid == 0) {
      s_npass = 0;
      s_npass2 = 0;
      s_prog = 0;
    }
    __syncthreads();
    for (unsigned i = tid; i < n; i += RT) {
      const int cell = s_cell[i];
      if (s_state[cell] >= 3u) s_pass[atomicAdd(&s_npass, 1u)] = (unsigned short)cell;
    }
    __syncthreads();
    const unsigned nlive = s_npass;
    if (nlive == 0) break;                                     // nothing left to decide in this tile
    // Many live candidates: one dense pass finds those whose word tops their window, and only
    // they get a wave.  Few (most tiles, and the late iterations of the busy ones): every live
    // candidate gets a wave and looks for a larger live word in its window itself.
    const unsigned short* list = s_pass;
    unsigned npass = nlive;
    if (nlive > (unsigned)LIVE_SPARSE) {
      // ---- window maximum of the state words: columns (lanes along x), then rows (lanes along y) ----
      for (int it = tid; it < LW * (CY / 8); it += RT) {
        const int st = it / LW, x = it - st * LW;
        const int ys = st * 8;
        const unsigned* col = s_state + ys * SP + x;
        if (R_T > 0) {
          unsigned v[8 + 2 * R_T];
#pragma unroll
          for (int k = 0; k < 8 + 2 * R_T; ++k) v[k] = col[k * SP];
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            unsigned m = v[o];
#pragma unroll
            for (int d = 1; d <= 2 * R_T; ++d) m = max(m, v[o + d]);
            s_v[(ys + o) * SP + x] = m;
          }
        } else {
          for (int o = 0; o < 8; ++o) {
            unsigned m = col[o * SP];
            for (int d = 1; d < WN; ++d) m = max(m, col[(o + d) * SP]);
            s_v[(ys + o) * SP + x] = m;
          }
        }
      }
      __syncthreads();
      {
        // one strip of RS window maxima per thread (CY * CX / RS == RT), lanes along y; they are
        // written over the column maxima, so every thread reads before any thread writes
        constexpr int RS = CX * CY / RT;
        const int y = tid & (CY - 1), xs = (tid / CY) * RS;
        const unsigned* row = s_v + y * SP + xs;
        unsigned out[RS];
        if (R_T > 0) {
          unsigned v[RS + 2 * R_T];
#pragma unroll
          for (int k = 0; k < RS + 2 * R_T; ++k) v[k] = row[k];
#pragma unroll
          for (int o = 0; o < RS; ++o) {
            unsigned m = v[o];
#pragma unroll
            for (int d = 1; d <= 2 * R_T; ++d) m = max(m, v[o + d]);
            out[o] = m;
          }
        } else {
#pragma unroll
          for (int o = 0; o < RS; ++o) {
            unsigned m = row[o];
            for (int d = 1; d < WN; ++d) m = max(m, row[o + d]);
            out[o] = m;
          }
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < RS; ++o) s_m[y * PM + xs + o] = out[o];
      }
      __syncthreads();
      for (unsigned k = tid; k < nlive; k += RT) {
        const int cell = s_pass[k];
        const int ly = cell / SP, lx = cell - ly * SP;
        if (s_m[(ly - r) * PM + (lx - r)] == s_state[cell]) s_pass2[atomicAdd(&s_npass2, 1u)] = (unsigned short)cell;
      }
      __syncthreads();
      list = s_pass2;
      npass = s_npass2;
    }
    // ---- one wave per passing candidate: lanes share the window ----
    // Equal words are settled by the full score, then the flat index (a total order, so of
    // two tied neighbours exactly one proceeds).  The winner is selected and kills the live
    // words of its window, in LDS and in the global map.
    const int lane = tid & 63;
    for (unsigned k = tid >> 6; k < npass; k += RT / 64) {
      const int cell = list[k];
      unsigned* c = s_state + cell;
      const unsigned cp = *c;
      if (cp < 3u) continue;                                   // killed by a tied winner meanwhile
      const int ly = cell / SP, lx = cell - ly * SP;
      const int py = y0 - r + ly, px = x0 - r + lx;
      bool blocked = false, killed = false;
      for (int t = lane; t < WN * WN; t += 64) {
        const int j = t / WN - r, d = t - (t / WN) * WN - r;
        if (j == 0 && d == 0) continue;
        const unsigned v = c[j * SP + d];
        if (v == 1u) killed = true;                            // a selected pixel owns this window
        if (v >= 3u && v > cp) blocked = true;                 // a live neighbour has a larger word
        if (v == cp) {
          const double s = sc[(size_t)py * W + px];
          const double q = sc[(size_t)(py + j) * W + (px + d)];
          const bool before = (j < 0) || (j == 0 && d < 0);
          if (before ? (q >= s) : (q > s)) blocked = true;
        }
      }
      if (__ballot(killed) != 0ull) {
        if (lane == 0) {
          *c = 0u;
          store_state(alive + (size_t)py * W + px, 0u);
          s_prog = 1;
        }
        continue;
      }
      if (__ballot(blocked) != 0ull) continue;
      for (int t = lane; t < WN * WN; t += 64) {
        const int j = t / WN - r, d = t - (t / WN) * WN - r;
        if (j == 0 && d == 0) continue;
        unsigned* q = c + j * SP + d;
        if (*q >= 3u) {                                        // live words only (always inside the image)
          *q = 0u;
          store_state(alive + (size_t)(py + j) * W + (px + d), 0u);
        }
      }
      if (lane == 0) {
        *c = 1u;
        store_state(alive + (size_t)py * W + px, 1u);
        s_sel[atomicAdd(&s_nsel, 1u)] = (unsigned short)cell;   // flushed to the global list at the end
        s_prog = 1;
      }
    }
    __syncthreads();   // LDS and global updates of this iteration are complete before the reload
    reload = s_prog == 0;   // nothing decided here: look at what the neighbours did
  }
  // append this launch's selections (and, in the last launch, what is still undecided) to the
  // global list: one reservation per tile
  __syncthreads();
  {
    if (tid == 0) s_npass = 0;
    __syncthreads();
    for (unsigned b0 = 0; b0 < n; b0 += RT) {
      const unsigned i = b0 + tid;
      const bool live = i < n && s_state[s_cell[i]] >= 3u;
      const unsigned slot = wave_slot(live, &s_npass);
      if (FINAL && live) s_pass[slot] = s_cell[i];
    }
    __syncthreads();
    if (tid == 0) seg_cnt[blk].w = s_npass;
    __syncthreads();
  }
  const unsigned nsel = s_nsel;
  const unsigned nrem = FINAL ? s_npass : 0u;
  if (nsel + nrem == 0) return;
  __syncthreads();
  if (tid == 0) {
    s_npass = atomicAdd(&ctl->n_c, nsel + nrem);
    if (nrem) atomicAdd(&ctl->n_rem, nrem);
  }
  __syncthreads();
  const unsigned base = s_npass;
  for (unsigned k = tid; k < nsel + nrem; k += RT) {
    const bool und = k >= nsel;
    const int cell = und ? s_pass[k - nsel] : s_sel[k];
    const int ly = cell / SP, lx = cell - ly * SP;
    const unsigned idx = (unsigned)(y0 - r + ly) * (unsigned)W + (unsigned)(x0 - r + lx);
    const unsigned pos = base + k;
    if (pos < cap_c) {
      keys_c[pos] = (unsigned long long)__double_as_longlong(sc[idx]);
      idx_c[pos] = (idx << 2) | ENT_CAND | (und ? ENT_UNDECIDED : 0u);
    } else {
      ctl->overflow = 1;
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 4, usual case (nothing left undecided): rank by counting, then emit
// ---------------------------------------------------------------------------------
constexpr int RK_I = 256;     // entries ranked per tile (one per thread)
constexpr int RK_J = 128;     // entries compared against per tile
constexpr int RK_BLOCKS = 1024;

__device__ __forceinline__ bool prio_before(unsigned long long ka, unsigned ia, unsigned long long kb,
                                            unsigned ib) {
  // higher score first; equal scores: lower flat index first (the low bits are flags, idx is unique)
  return ka > kb || (ka == kb && ia < ib);
}

// Also returns the state map to all-zero for the next call (the candidates' marks are the only
// words ever set in it).
__global__ __launch_bounds__(NT) void nms_rank_kernel(const unsigned long long* __restrict__ keys_c,
                                                      const unsigned* __restrict__ idx_c, const nms_ctl* ctl,
                                                      unsigned* __restrict__ rank, unsigned* __restrict__ alive,
                                                      const uint4* __restrict__ seg_cnt,
                                                      const unsigned* __restrict__ seg_cand, unsigned nblk,
                                                      nms_batch B) {
  if (B.go && !B.go[blockIdx.y]) return;
  {
    const size_t q = blockIdx.y;
    keys_c += q * B.comp;
    idx_c += q * B.comp;
    ctl += q;
    rank += q * B.rank;
    alive += q * B.alive;
    seg_cnt += q * B.segcnt;
    seg_cand += q * B.seg;
  }
  __shared__ unsigned long long s_k[RK_J];
  __shared__ unsigned s_i[RK_J];
  const int tid = threadIdx.x;
  for (unsigned tile = blockIdx.x; tile < nblk; tile += gridDim.x) {
    const unsigned n = seg_cnt[tile].z;
    for (unsigned i = tid; i < n; i += NT) alive[seg_cand[(size_t)tile * SEG + i]] = 0u;
  }
  const unsigned M = ctl->n_c;
  if (ctl->n_rem != 0 || M > (unsigned)RANK_MAX) return;
  const unsigned ti = (M + RK_I - 1) / RK_I, tj = (M + RK_J - 1) / RK_J;
  for (unsigned tile = blockIdx.x; tile < ti * tj; tile += gridDim.x) {
    const unsigned i0 = (tile / tj) * RK_I, j0 = (tile % tj) * RK_J;
    __syncthreads();
    if (tid < RK_J) {
      const unsigned j = j0 + tid;
      s_k[tid] = j < M ? keys_c[j] : 0ull;
      s_i[tid] = j < M ? idx_c[j] : 0xffffffffu;
    }
    const unsigned i = i0 + tid;
    const unsigned long long ka = i < M ? keys_c[i] : ~0ull;
    const unsigned xa = i < M ? idx_c[i] : 0u;
    __syncthreads();
    unsigned c = 0;
#pragma unroll 16
    for (int k = 0; k < RK_J; ++k) c += prio_before(s_k[k], s_i[k], ka, xa) ? 1u : 0u;
    if (i < M && c) atomicAdd(&rank[i], c);
  }
}

constexpr int SEL_T = 1024;       // threads of the single-workgroup kernel
constexpr int CHUNK = 8192;       // entries sorted in LDS at a time
constexpr int MAX_N = 16384;      // keypoints

__device__ __forceinline__ void write_keypoints(const unsigned* sel, unsigned nsel, unsigned edge, int W, int N,
                                                double* __restrict__ kp_xy, float* __restrict__ kp_f32) {
  // reference slicing corner cases: a pick with y < r or x < r suppresses nothing and is
  // returned for every later slot; once the scores are exhausted the picks are (0, 0)
  for (unsigned i = threadIdx.x; i < (unsigned)N; i += SEL_T) {
    double x = 0.0, y = 0.0;
    unsigned src = i;
    if (edge != 0xffffffffu && i > edge) src = edge;
    if (src < nsel) {
      const unsigned idx = sel[src];
      const unsigned yy = idx / (unsigned)W;
      y = (double)yy;
      x = (double)(idx - yy * (unsigned)W);
    }
    kp_xy[2 * i] = x;
    kp_xy[2 * i + 1] = y;
    if (kp_f32) {   // the same points as float pairs, the form the KLT tracker takes
      kp_f32[2 * i] = (float)x;
      kp_f32[2 * i + 1] = (float)y;
    }
  }
}

// ---------------------------------------------------------------------------------
// NMS stage 4, general case (one workgroup): sort by priority, greedy walk, write keypoints
// ---------------------------------------------------------------------------------
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

// Last stage, one workgroup.  Usual case (nothing undecided, short list): the ranks are final,
// the first N entries by rank are the keypoints.  Otherwise: sort + sequential walk.
__global__ __launch_bounds__(SEL_T) void nms_finalize_kernel(unsigned long long* __restrict__ keys_c,
                                                             unsigned* __restrict__ idx_c, nms_ctl* ctl,
                                                             unsigned* __restrict__ rank,
                                                             unsigned cap_pow2, int W, int N, int r,
                                                             unsigned* __restrict__ sel,
                                                             double* __restrict__ kp_xy,
                                                             float* __restrict__ kp_f32, nms_batch B) {
  if (B.go && !B.go[blockIdx.y]) return;
  {
    const size_t q = blockIdx.y;
    keys_c += q * B.comp;
    idx_c += q * B.comp;
    ctl += q;
    rank += q * B.rank;
    sel += q * B.sel;
    kp_xy += q * B.kp;
    if (kp_f32) kp_f32 += q * B.kp;
  }
  __shared__ __align__(16) unsigned long long s_keys[CHUNK];   // 64 KiB
  __shared__ __align__(16) unsigned s_idx[CHUNK];              // 32 KiB
  __shared__ unsigned s_wsum[SEL_T / 64];
  __shared__ unsigned s_nsel, s_nsela, s_nalist, s_flag, s_edge;
  const int tid = threadIdx.x;

  if (ctl->n_rem == 0 && ctl->n_c <= (unsigned)RANK_MAX) {
    const unsigned Mr = ctl->n_c;
    if (tid == 0) s_edge = 0xffffffffu;
    __syncthreads();
    for (unsigned i = tid; i < Mr; i += SEL_T) {
      const unsigned rk = rank[i];
      rank[i] = 0;   // all-zero again for the next call
      if (rk < (unsigned)N) {
        const unsigned idx = idx_c[i] >> 2;
        sel[rk] = idx;
        const unsigned py = idx / (unsigned)W, px = idx - py * (unsigned)W;
        if ((int)py < r || (int)px < r) atomicMin(&s_edge, rk);
      }
    }
    __threadfence_block();
    __syncthreads();
    const unsigned nsel = min(Mr, (unsigned)N);
    write_keypoints(sel, nsel, s_edge, W, N, kp_xy, kp_f32);
    if (tid == 0) ctl->n_sel = nsel;
    return;
  }
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
    const bool is_a = (e & ENT_UNDECIDED) != 0;            // still needs the greedy test
    const bool is_cand = (e & (ENT_UNDECIDED | ENT_CAND)) != 0;   // once selected it suppresses later undecided entries
    const unsigned idx = e >> 2;
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
      // ... and of the candidates of this batch that are already selected: they suppress too
      const bool listed = stat == 0u || (valid && is_cand && stat == 1u);
      unsigned long long m = __ballot(listed);
      unsigned lane = tid & 63, wv = tid >> 6;
      unsigned before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wsum[wv] = __popcll(m);
      __syncthreads();
      unsigned off = 0;
      for (unsigned w = 0; w < wv; ++w) off += s_wsum[w];
      if (listed) b_alist[off + before] = tid;
      if (tid == SEL_T - 1) s_nalist = off + before + (listed ? 1u : 0u);
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
        if (is_cand) {
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

  const unsigned nsel = min(s_nsel, (unsigned)N);
  write_keypoints(sel, nsel, s_edge, W, N, kp_xy, kp_f32);
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

// the same patches as bytes, one row of row_bytes (>= (2r+1)^2, the rest zero) per keypoint: what the device-resident
// harris tracker mode matches on the matrix cores (the values are pixels: whole numbers 0..255 by construction)
__global__ __launch_bounds__(128) void patch_desc_u8_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                            const double* __restrict__ kp_xy, int N, int r,
                                                            uint8_t* __restrict__ desc, int row_bytes) {
  const int k = blockIdx.x;
  const int d = 2 * r + 1;
  const int x = (int)kp_xy[2 * k], y = (int)kp_xy[2 * k + 1];
  uint8_t* o = desc + (size_t)k * row_bytes;
  for (int i = threadIdx.x; i < row_bytes; i += 128) {
    uint8_t v = 0;
    if (i < d * d) {
      const int dy = i / d, dx = i - dy * d;
      const int gy = y - r + dy, gx = x - r + dx;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = img[(size_t)gy * W + gx];
    }
    o[i] = v;
  }
}

size_t response_lds_bytes(int p) {
  const resp_geom g = response_geometry(p);
  const int HR = p == 9 ? RY / 2 + p - 1 : g.GH;     // (the compile-time patch holds half the rows of sums at a time)
  const size_t img = (size_t)((g.IWp * g.IH + 15) & ~15), sums = (size_t)3 * HR * RX * 4;   // (share their bytes)
  return (size_t)g.GWp * g.GH * 4 + (img > sums ? img : sums);
}

size_t candidates_lds_bytes(int r) {
  const int RW = CX + 4 * r, RH = CY + 4 * r;
  const int LW = CX + 2 * r, LH = CY + 2 * r;
  return (size_t)RW * RH * 4 + (size_t)RW * LH * 4 + (size_t)LH * 16 + (size_t)CY * 16 + (size_t)LW * LH * 2;
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
  return vo_harris_response_batch_dev(ctx, d_img, 0, 1, H, W, patch, kappa, d_scores);
}

}  // extern "C"

// S images (d_img + s * img_stride) -> S score maps (d_scores + s * H * W), one launch (grid.z = sequence)
int vo_harris_response_batch_dev(vo_ctx* ctx, const uint8_t* d_img, size_t img_stride, int S, int H, int W, int patch,
                                 double kappa, double* d_scores, const int* d_go) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, S >= 1 && S <= 65535, "harris_response: bad sequence count");
  VO_REQUIRE(ctx, d_img && d_scores, "harris_response: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "harris_response: bad image size %dx%d", W, H);
  VO_REQUIRE(ctx, patch >= 3 && patch <= 31 && (patch & 1), "harris_response: patch must be odd in 3..31");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  dim3 grid(vo_cdiv(W, RX), vo_cdiv(H, RY), S);
  size_t lds = response_lds_bytes(patch);
  {
    bool& lds_opt_in = ctx->lds_opt_in[0];   // dynamic LDS above 64 KiB must be requested per kernel (and device)
    if (!lds_opt_in) {
      VO_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&harris_response_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)response_lds_bytes(31)));
      lds_opt_in = true;
    }
    vo_prof_scope ps(ctx, VO_K_HARRIS_RESPONSE);
    if (patch == 9)
      hipLaunchKernelGGL(harris_response_kernel<9>, grid, dim3(NT), lds, ctx->stream, d_img, H, W, patch, kappa,
                         d_scores, img_stride, d_go);
    else
      hipLaunchKernelGGL(harris_response_kernel<0>, grid, dim3(NT), lds, ctx->stream, d_img, H, W, patch, kappa,
                         d_scores, img_stride, d_go);
  }
  return vo_check_launch(ctx, "harris_response_kernel");
}

extern "C" {

int vo_nms_keypoints_dev(vo_ctx* ctx, const double* d_scores, int H, int W, int N, int r, double* d_kp_xy) {
  return vo_nms_keypoints_batch_dev(ctx, d_scores, 1, H, W, N, r, d_kp_xy, 0);
}

}  // extern "C"

// S score maps (d_scores + s * H * W) -> S keypoint lists (d_kp_xy + s * kp_stride doubles; the float copies, when
// ctx->nms_kp_f32 is set, at the same element stride), every kernel of the chain launched once for all sequences
int vo_nms_keypoints_batch_dev(vo_ctx* ctx, const double* d_scores, int S, int H, int W, int N, int r, double* d_kp_xy,
                               size_t kp_stride, const int* d_go) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, S >= 1 && S <= 65535, "nms: bad sequence count");
  VO_REQUIRE(ctx, d_scores && d_kp_xy, "nms: null pointer");
  VO_REQUIRE(ctx, H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "nms: bad map size %dx%d", W, H);
  VO_REQUIRE(ctx, W < 65536 && H < 65536, "nms: map side must be < 65536");
  VO_REQUIRE(ctx, N >= 1 && N <= MAX_N, "nms: N must be in 1..%d", MAX_N);
  VO_REQUIRE(ctx, r >= 0 && r <= 12, "nms: radius must be in 0..12");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const unsigned cap = (unsigned)H * (unsigned)W;
  const unsigned cap_c = next_pow2(2u * cap);
  const dim3 grid(vo_cdiv(W, CX), vo_cdiv(H, CY), S);
  const unsigned nblk = grid.x * grid.y;
  const size_t seg_total = (size_t)nblk * SEG;
  const size_t Sz = (size_t)S;
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_l1, Sz * seg_total * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_l1, Sz * seg_total * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_a1, Sz * seg_total * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_a1, Sz * seg_total * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_cand, Sz * seg_total * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_segcnt, Sz * nblk * sizeof(uint4)));
  VO_TRY(vo_ensure(ctx, ctx->nms_keys_c, Sz * cap_c * 8));
  VO_TRY(vo_ensure(ctx, ctx->nms_idx_c, Sz * cap_c * 4));
  VO_TRY(vo_ensure(ctx, ctx->nms_sel, Sz * MAX_N * 4));
  if (ctx->nms_hist.cap < Sz * 2 * HIST_TOTAL * 4 || ctx->nms_S != S) {     // one histogram per call parity and sequence;
    VO_TRY(vo_ensure(ctx, ctx->nms_hist, Sz * 2 * HIST_TOTAL * 4));         // where each lies depends on S
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->nms_hist.p, 0, ctx->nms_hist.cap, ctx->stream));
    ctx->nms_S = S;
  }
  if (ctx->nms_rank.cap < Sz * RANK_MAX * 4) {
    VO_TRY(vo_ensure(ctx, ctx->nms_rank, Sz * RANK_MAX * 4));
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->nms_rank.p, 0, ctx->nms_rank.cap, ctx->stream));
  }
  if (ctx->nms_alive.cap < Sz * cap * 4 || ctx->nms_alive_dirty) {
    VO_TRY(vo_ensure(ctx, ctx->nms_alive, Sz * cap * 4));
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->nms_alive.p, 0, ctx->nms_alive.cap, ctx->stream));
    ctx->nms_alive_dirty = false;
  }
  VO_TRY(vo_ensure(ctx, ctx->nms_ctl, Sz * sizeof(nms_ctl)));
  nms_batch B;
  if (S > 1) {
    B.sc = cap;
    B.seg = seg_total;
    B.segcnt = nblk;
    B.comp = cap_c;
    B.alive = cap;
    B.hist = HIST_TOTAL;
    B.rank = RANK_MAX;
    B.sel = MAX_N;
    B.kp = kp_stride;
  }
  B.go = d_go;
  nms_ctl* ctl = (nms_ctl*)ctx->nms_ctl.p;   // its counters are reset by the threshold kernel

  unsigned long long* keys_l1 = (unsigned long long*)ctx->nms_keys_l1.p;
  unsigned long long* keys_a1 = (unsigned long long*)ctx->nms_keys_a1.p;
  unsigned* idx_l1 = (unsigned*)ctx->nms_idx_l1.p;
  unsigned* idx_a1 = (unsigned*)ctx->nms_idx_a1.p;
  unsigned long long* keys_c = (unsigned long long*)ctx->nms_keys_c.p;
  unsigned* idx_c = (unsigned*)ctx->nms_idx_c.p;
  unsigned* cand = (unsigned*)ctx->nms_cand.p;
  uint4* segcnt = (uint4*)ctx->nms_segcnt.p;
  unsigned* alive = (unsigned*)ctx->nms_alive.p;
  ctx->nms_parity ^= 1;
  unsigned* hist = (unsigned*)ctx->nms_hist.p + (size_t)ctx->nms_parity * Sz * HIST_TOTAL;
  unsigned* hist_other = (unsigned*)ctx->nms_hist.p + (size_t)(1 - ctx->nms_parity) * Sz * HIST_TOTAL;
  unsigned* rank = (unsigned*)ctx->nms_rank.p;
  hipStream_t st = ctx->stream;
  {
    bool& lds_opt_in = ctx->lds_opt_in[1];   // dynamic LDS above 64 KiB must be requested per kernel (and device)
    if (!lds_opt_in) {
      const int max_dyn = (int)candidates_lds_bytes(12);   // largest radius accepted above
      VO_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_candidates_kernel<5>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
      VO_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_candidates_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
      lds_opt_in = true;
    }
    vo_prof_scope ps(ctx, VO_K_NMS_CANDIDATES);
    const size_t lds = candidates_lds_bytes(r);
    if (r == 5)
      hipLaunchKernelGGL(nms_candidates_kernel<5>, grid, dim3(NT), lds, st, d_scores, H, W, r, keys_l1, idx_l1,
                         keys_a1, idx_a1, segcnt, hist, ctl, B);
    else
      hipLaunchKernelGGL(nms_candidates_kernel<0>, grid, dim3(NT), lds, st, d_scores, H, W, r, keys_l1, idx_l1,
                         keys_a1, idx_a1, segcnt, hist, ctl, B);
  }
  VO_TRY(vo_check_launch(ctx, "nms_candidates_kernel"));
  ctx->nms_alive_dirty = true;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_COMPACT);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(nblk, S), dim3(NT), 0, st, keys_l1, idx_l1, keys_a1, idx_a1, segcnt,
                       cand, keys_c, idx_c, alive, ctl, cap_c, hist, hist_other, N, B);
  }
  VO_TRY(vo_check_launch(ctx, "nms_compact_kernel"));
  const size_t round_lds = (size_t)((CX + 2 * r) | 1) * ((CY + 2 * r) + CY) * 4;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_ROUND);
    const dim3 g(nblk, S), b(RT);
    if (r == 5)
      hipLaunchKernelGGL((nms_round_kernel<5, true>), g, b, round_lds, st, d_scores, alive, segcnt, cand, keys_c,
                         idx_c, ctl, cap_c, H, W, r, (int)grid.x, B);
    else
      hipLaunchKernelGGL((nms_round_kernel<0, true>), g, b, round_lds, st, d_scores, alive, segcnt, cand, keys_c,
                         idx_c, ctl, cap_c, H, W, r, (int)grid.x, B);
  }
  VO_TRY(vo_check_launch(ctx, "nms_round_kernel"));
  {
    // ranks for the usual case (nothing undecided, short list); returns the state map to all-zero
    vo_prof_scope ps(ctx, VO_K_NMS_RANK);
    hipLaunchKernelGGL(nms_rank_kernel, dim3(RK_BLOCKS, S), dim3(NT), 0, st, keys_c, idx_c, ctl, rank, alive, segcnt,
                       cand, nblk, B);
  }
  VO_TRY(vo_check_launch(ctx, "nms_rank_kernel"));
  ctx->nms_alive_dirty = false;
  {
    vo_prof_scope ps(ctx, VO_K_NMS_SELECT);
    hipLaunchKernelGGL(nms_finalize_kernel, dim3(1, S), dim3(SEL_T), 0, st, keys_c, idx_c, ctl, rank, cap_c, W, N, r,
                       (unsigned*)ctx->nms_sel.p, d_kp_xy, ctx->nms_kp_f32, B);
  }
  return vo_check_launch(ctx, "nms_finalize_kernel");
}

extern "C" {

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

int vo_patch_descriptors_u8_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, const double* d_kp_xy, int N, int r,
                                uint8_t* d_desc, int row_bytes) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_img && d_kp_xy && d_desc && N >= 1 && r >= 0 && (2 * r + 1) * (2 * r + 1) <= row_bytes,
             "patch_descriptors_u8: bad arguments");
  {
    vo_prof_scope ps(ctx, VO_K_PATCH_DESC);
    hipLaunchKernelGGL(patch_desc_u8_kernel, dim3(N), dim3(128), 0, ctx->stream, d_img, H, W, d_kp_xy, N, r, d_desc, row_bytes);
  }
  return vo_check_launch(ctx, "patch_desc_u8_kernel");
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

int vo_harris_keypoints_batch(vo_ctx* ctx, const uint8_t* imgs, int S, int H, int W, int patch, double kappa, int N,
                              int r, double* kp_xy, double* scores) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, imgs && kp_xy, "harris_keypoints_batch: null pointer");
  VO_REQUIRE(ctx, S >= 1 && H > 0 && W > 0 && N >= 1, "harris_keypoints_batch: bad arguments");
  const size_t n = (size_t)H * W;
  VO_TRY(vo_ensure(ctx, ctx->img, (size_t)S * n));
  VO_TRY(vo_ensure(ctx, ctx->scores, (size_t)S * n * 8));
  VO_TRY(vo_ensure(ctx, ctx->kp, (size_t)S * N * 16));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ctx->img.p, imgs, (size_t)S * n, hipMemcpyHostToDevice, ctx->stream));
  VO_TRY(vo_harris_response_batch_dev(ctx, (const uint8_t*)ctx->img.p, n, S, H, W, patch, kappa, (double*)ctx->scores.p));
  VO_TRY(vo_nms_keypoints_batch_dev(ctx, (const double*)ctx->scores.p, S, H, W, N, r, (double*)ctx->kp.p, (size_t)N * 2));
  VO_HIP_TRY(ctx, hipMemcpyAsync(kp_xy, ctx->kp.p, (size_t)S * N * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (scores) VO_HIP_TRY(ctx, hipMemcpyAsync(scores, ctx->scores.p, (size_t)S * n * 8, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<nms_ctl> h((size_t)S);
  VO_HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->nms_ctl.p, (size_t)S * sizeof(nms_ctl), hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int q = 0; q < S; ++q)
    if (h[q].overflow) return vo_set_error(ctx, VO_ECAPACITY, "nms: candidate list overflow (sequence %d)", q);
  return VO_OK;
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
