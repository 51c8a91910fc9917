// Brute-force 2-nearest-neighbour descriptor matching with ratio test for gfx950.
//
// Reference call sites: src/vo/features/harris.py:246-262 (raw 19x19 patches, D = 361,
// ratio 0.85) and src/vo/features/sift.py:38-54 (D = 128, ratio 0.8):
//   cv2.BFMatcher().knnMatch(desc1, desc2, k=2), keep m if m.distance < ratio * n.distance
//   and the train index has not been used yet (queries in order).
// Both descriptor kinds are integer-valued in 0..255, so the squared distance
//   |a|^2 + |b|^2 - 2 a.b
// is an exact integer: descriptors are packed to bytes and a.b runs on the matrix cores
// (v_mfma_i32_32x32x32_i8, knn2_mfma_kernel: unsigned bytes through the -128 offset identity, exact in int32;
// descriptor lengths 128 and 361) or on the packed 4-way byte dot product (v_dot4_u32_u8, any other length).
// Any other input takes the float path (float64 accumulation in index order).  Either
// way the result equals the oracle's definition (oracle/csrc/match.c) bit for bit.
// Packing to bytes (with the check that every value is a whole number in 0..255), the ratio test and the
// first-come uniqueness filter run on the device too: a train index goes to the first query, in query order,
// that passes the ratio test with it -- the smallest such query index (atomicMin), then an ordered compaction.
#include <cmath>

#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int MT = 256;    // threads per workgroup
constexpr int QB = 8;      // queries per workgroup

struct top2 {
  unsigned long long d0, d1;   // order-preserving keys: (distance bits << 32) | index
};

__device__ __forceinline__ void top2_insert(unsigned long long& k0, unsigned long long& k1, unsigned long long k) {
  // (selects only: written with branches the compiler kept the pair in scratch memory inside the MFMA kernel's loop)
  const unsigned long long lo = k < k0 ? k : k0, hi = k < k0 ? k0 : k;
  k0 = lo;
  k1 = hi < k1 ? hi : k1;
}

// descriptors as bytes, rows padded to Dp = multiple of 4
__global__ __launch_bounds__(MT) void knn2_u8_kernel(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                     int nt, int Dp, int* __restrict__ best, double* __restrict__ d2) {
  extern __shared__ __align__(16) unsigned s_mem[];
  unsigned* s_q = s_mem;                                    // QB rows of Dp/4 words
  unsigned long long* s_k = reinterpret_cast<unsigned long long*>(s_q + QB * (Dp / 4));   // [QB][MT][2]
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * QB;
  const int words = Dp / 4;
  for (int i = tid; i < QB * words; i += MT) {
    const int r = i / words, w = i - r * words;
    s_q[i] = (q0 + r < nq) ? reinterpret_cast<const unsigned*>(q)[(size_t)(q0 + r) * words + w] : 0u;
  }
  __syncthreads();
  unsigned nq2[QB];
#pragma unroll
  for (int r = 0; r < QB; ++r) {
    unsigned s = 0;
    for (int w = 0; w < words; ++w) s = __builtin_amdgcn_udot4(s_q[r * words + w], s_q[r * words + w], s, false);
    nq2[r] = s;
  }
  unsigned long long k0[QB], k1[QB];
#pragma unroll
  for (int r = 0; r < QB; ++r) k0[r] = k1[r] = ~0ull;
  for (int j = tid; j < nt; j += MT) {
    const unsigned* row = reinterpret_cast<const unsigned*>(t) + (size_t)j * words;
    unsigned nb = 0, ab[QB];
#pragma unroll
    for (int r = 0; r < QB; ++r) ab[r] = 0;
    for (int w = 0; w < words; ++w) {
      const unsigned b = row[w];
      nb = __builtin_amdgcn_udot4(b, b, nb, false);
#pragma unroll
      for (int r = 0; r < QB; ++r) ab[r] = __builtin_amdgcn_udot4(s_q[r * words + w], b, ab[r], false);
    }
#pragma unroll
    for (int r = 0; r < QB; ++r) {
      const unsigned dist = nq2[r] + nb - 2u * ab[r];
      top2_insert(k0[r], k1[r], ((unsigned long long)dist << 32) | (unsigned)j);
    }
  }
#pragma unroll
  for (int r = 0; r < QB; ++r) {
    s_k[(r * MT + tid) * 2] = k0[r];
    s_k[(r * MT + tid) * 2 + 1] = k1[r];
  }
  __syncthreads();
  // one wave per query row reduces the 2 * MT keys
  const int lane = tid & 63, wv = tid >> 6;
  for (int r = wv; r < QB; r += MT / 64) {
    unsigned long long a0 = ~0ull, a1 = ~0ull;
    for (int i = lane; i < 2 * MT; i += 64) top2_insert(a0, a1, s_k[r * MT * 2 + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long b0 = __shfl_xor(a0, off), b1 = __shfl_xor(a1, off);
      top2_insert(a0, a1, b0);
      top2_insert(a0, a1, b1);
    }
    if (lane == 0 && q0 + r < nq) {
      const int qi = q0 + r;
      best[2 * qi] = a0 == ~0ull ? -1 : (int)(a0 & 0xffffffffu);
      best[2 * qi + 1] = a1 == ~0ull ? -1 : (int)(a1 & 0xffffffffu);
      d2[2 * qi] = a0 == ~0ull ? 0.0 : (double)(unsigned)(a0 >> 32);
      d2[2 * qi + 1] = a1 == ~0ull ? 0.0 : (double)(unsigned)(a1 >> 32);
    }
  }
}

// ---- the same on the matrix cores ----
// A wave owns a 32 (train rows) x 32 (queries) tile of a . b per step: v_mfma_i32_32x32x32_i8 takes signed bytes, so
// every byte is flipped to x - 128 (x ^ 0x80) and
//     a . b = sum (a' + 128)(b' + 128) = dot' + 128 (sum a + sum b) - 16384 Dp          (sums over the padded length Dp)
//     |a - b|^2 = (|a|^2 - 256 sum a + 32768 Dp) + (|b|^2 - 256 sum b) - 2 dot'
// with every term an exact integer below 2^31.  The queries are the B operand (kept in registers for the whole
// train set), the train rows the A operand (16 bytes per lane and instruction, next tile's loads issued before this
// tile's arithmetic): the result tile has its query on the lane (column = lane & 31) and 16 train rows in the lane's
// registers, so the running top-2 of a query is lane-local; the 8 partial lists of a query (2 lane halves x 4 waves)
// meet in LDS at the end.  The train set is split over gridDim.y workgroups per block of 32 queries (63 workgroups of
// one wave per SIMD each would wait out every load: 93 us at 2000 x 2000 against 63 us for the byte-dot kernel); each
// leaves its top-2 in global memory and the one that arrives last merges them.  Whatever order the instruction assigns the 32 k-positions of a step to (lane half,
// element), A and B use the same one: the sum over k does not depend on it.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KB>   // Dp = 32 * KB bytes per row
__global__ __launch_bounds__(256) void knn2_mfma_kernel(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                        int nt, unsigned long long* __restrict__ part,
                                                        unsigned* __restrict__ arrived, int* __restrict__ best,
                                                        double* __restrict__ d2, const int* __restrict__ d_nq = nullptr,
                                                        const int* __restrict__ d_nt = nullptr) {
  constexpr int Dp = 32 * KB;
  if (d_nq) nq = min(nq, *d_nq);                       // counts that live on the device (frame pipeline): the launch is
  if (d_nt) nt = min(nt, *d_nt);                       // sized for the capacities
  if ((int)blockIdx.x * 32 >= nq || nt <= 0) {         // (every split of a block of queries leaves together)
    if (nt <= 0 && threadIdx.x < 32 && (int)blockIdx.x * 32 + (int)threadIdx.x < nq && blockIdx.y == 0) {
      const int qi = blockIdx.x * 32 + threadIdx.x;
      best[2 * qi] = best[2 * qi + 1] = -1;
      d2[2 * qi] = d2[2 * qi + 1] = 0.0;
    }
    return;
  }
  __shared__ unsigned s_t[4][32];                      // per wave: the tile's train terms
  __shared__ unsigned long long s_k[32][8][2];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32;
  // the wave's queries: B fragments + the query term of column r
  v4i bq[KB];
  unsigned qterm;
  {
    const int qi = min(q0 + r, nq - 1);
    unsigned n2 = 0, sm = 0;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const v4i w = *reinterpret_cast<const v4i*>(q + (size_t)qi * Dp + kb * 32 + 16 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        n2 = __builtin_amdgcn_udot4((unsigned)w[e], (unsigned)w[e], n2, false);
        sm = __builtin_amdgcn_udot4((unsigned)w[e], 0x01010101u, sm, false);
      }
      bq[kb] = w ^ (int)0x80808080;
    }
    n2 += __shfl_xor(n2, 32);
    sm += __shfl_xor(sm, 32);
    qterm = n2 - 256u * sm + 32768u * (unsigned)Dp;
  }
  unsigned long long k0 = ~0ull, k1 = ~0ull;
  const int ntiles_all = (nt + 31) / 32;
  const int tile_lo = (int)((long long)ntiles_all * blockIdx.y / gridDim.y);
  const int ntiles = (int)((long long)ntiles_all * (blockIdx.y + 1) / gridDim.y);     // this workgroup: tiles tile_lo .. ntiles - 1
  v4i a_next[KB];
  auto load_tile = [&](int tile) {
    const int row = min(tile * 32 + r, nt - 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) a_next[kb] = *reinterpret_cast<const v4i*>(t + (size_t)row * Dp + kb * 32 + 16 * h);
  };
  if (tile_lo + wv < ntiles) load_tile(tile_lo + wv);
  for (int tile = tile_lo + wv; tile < ntiles; tile += 4) {
    v4i a[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) a[kb] = a_next[kb];
    if (tile + 4 < ntiles) load_tile(tile + 4);
    unsigned n2 = 0, sm = 0;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        n2 = __builtin_amdgcn_udot4((unsigned)a[kb][e], (unsigned)a[kb][e], n2, false);
        sm = __builtin_amdgcn_udot4((unsigned)a[kb][e], 0x01010101u, sm, false);
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kb] ^ (int)0x80808080, bq[kb], acc, 0, 0, 0);
    }
    n2 += __shfl_xor(n2, 32);
    sm += __shfl_xor(sm, 32);
    // (a wave's LDS operations execute in order: the previous tile's reads are done)
    if (h == 0) s_t[wv][r] = n2 - 256u * sm;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int m = (g & 3) + 8 * (g >> 2) + 4 * h;             // row of register g (C/D layout of the 32x32 shapes)
      const int j = tile * 32 + m;
      const unsigned dist = qterm + s_t[wv][m] - 2u * (unsigned)acc[g];
      if (j < nt) top2_insert(k0, k1, ((unsigned long long)dist << 32) | (unsigned)j);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  s_k[r][wv * 2 + h][0] = k0;
  s_k[r][wv * 2 + h][1] = k1;
  __syncthreads();
  unsigned long long a0 = ~0ull, a1 = ~0ull;
  if (tid < 32) {
    for (int i = 0; i < 8; ++i) {
      top2_insert(a0, a1, s_k[tid][i][0]);
      top2_insert(a0, a1, s_k[tid][i][1]);
    }
  }
  if (gridDim.y > 1) {
    // this workgroup's share of the train set is done: leave it, and merge everybody's if this is the last to arrive
    unsigned long long* mine = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64;
    if (tid < 32) {
      mine[2 * tid] = a0;
      mine[2 * tid + 1] = a1;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      const unsigned n = atomicAdd(&arrived[blockIdx.x], 1u);
      s_last = n == gridDim.y - 1 ? 1 : 0;
      if (s_last) arrived[blockIdx.x] = 0u;               // ready for the next call
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid < 32) {
      a0 = a1 = ~0ull;
      const unsigned long long* all = part + (size_t)blockIdx.x * gridDim.y * 64;
      for (unsigned sidx = 0; sidx < gridDim.y; ++sidx) {
        top2_insert(a0, a1, __hip_atomic_load(&all[sidx * 64 + 2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        top2_insert(a0, a1, __hip_atomic_load(&all[sidx * 64 + 2 * tid + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
  }
  if (tid < 32 && q0 + tid < nq) {
    const int qi = q0 + tid;
    best[2 * qi] = a0 == ~0ull ? -1 : (int)(a0 & 0xffffffffu);
    best[2 * qi + 1] = a1 == ~0ull ? -1 : (int)(a1 & 0xffffffffu);
    d2[2 * qi] = a0 == ~0ull ? 0.0 : (double)(unsigned)(a0 >> 32);
    d2[2 * qi + 1] = a1 == ~0ull ? 0.0 : (double)(unsigned)(a1 >> 32);
  }
}

// general float descriptors: float64 accumulation in index order, one lane per (query, train) pair
__global__ __launch_bounds__(MT) void knn2_f32_kernel(const float* __restrict__ q, int nq, const float* __restrict__ t,
                                                      int nt, int D, int* __restrict__ best, double* __restrict__ d2) {
  __shared__ unsigned long long s_d[MT][2];
  __shared__ int s_i[MT][2];
  const int qi = blockIdx.x, tid = threadIdx.x;
  const float* a = q + (size_t)qi * D;
  double e0 = 0, e1 = 0;
  int b0 = -1, b1 = -1;
  for (int j = tid; j < nt; j += MT) {
    const float* b = t + (size_t)j * D;
    double s = 0.0;
    for (int k = 0; k < D; ++k) {
      const double d = (double)a[k] - (double)b[k];
      s += d * d;
    }
    if (b0 < 0 || s < e0) {
      b1 = b0; e1 = e0; b0 = j; e0 = s;
    } else if (b1 < 0 || s < e1) {
      b1 = j; e1 = s;
    }
  }
  // non-negative doubles order like their bit patterns; ties go to the lower train index
  s_d[tid][0] = b0 < 0 ? ~0ull : (unsigned long long)__double_as_longlong(e0);
  s_d[tid][1] = b1 < 0 ? ~0ull : (unsigned long long)__double_as_longlong(e1);
  s_i[tid][0] = b0;
  s_i[tid][1] = b1;
  __syncthreads();
  if (tid == 0) {
    unsigned long long k0 = ~0ull, k1 = ~0ull;
    int i0 = -1, i1 = -1;
    for (int i = 0; i < MT; ++i)
      for (int c = 0; c < 2; ++c) {
        const unsigned long long k = s_d[i][c];
        const int id = s_i[i][c];
        if (id < 0) continue;
        if (i0 < 0 || k < k0 || (k == k0 && id < i0)) {
          k1 = k0; i1 = i0; k0 = k; i0 = id;
        } else if (i1 < 0 || k < k1 || (k == k1 && id < i1)) {
          k1 = k; i1 = id;
        }
      }
    best[2 * qi] = i0;
    best[2 * qi + 1] = i1;
    d2[2 * qi] = i0 < 0 ? 0.0 : __longlong_as_double((long long)k0);
    d2[2 * qi + 1] = i1 < 0 ? 0.0 : __longlong_as_double((long long)k1);
  }
}

// float descriptors -> bytes, rows padded with zeros to Dp; *not_bytes is raised when a value is not a whole number
// in 0..255 (the byte kernels' result is then discarded and the float kernel runs)
__global__ __launch_bounds__(256) void pack_bytes_kernel(const float* __restrict__ in, int n, int D, int Dp,
                                                         uint8_t* __restrict__ out, unsigned* __restrict__ not_bytes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * Dp) return;
  const int row = (int)(i / Dp), k = (int)(i - (size_t)row * Dp);
  uint8_t b = 0;
  if (k < D) {
    const float f = in[(size_t)row * D + k];
    if (!(f >= 0.f && f <= 255.f) || f != (float)(int)f) atomicOr(not_bytes, 1u);
    else b = (uint8_t)f;
  }
  out[i] = b;
}

// ratio test (harris.py:250-258, sift.py:45-52): m.distance < ratio * n.distance on float32 distances, then the
// train index goes to the first query that asks for it.  One workgroup; pairs come out in query order.
constexpr int RU_T = 1024;
__global__ __launch_bounds__(RU_T) void ratio_unique_kernel(const int* __restrict__ best, const double* __restrict__ d2, int nq,
                                                            double ratio, int* __restrict__ owner /* nt, preset to INT_MAX */,
                                                            int* __restrict__ pairs, int* __restrict__ n_pairs,
                                                            const int* __restrict__ d_nq = nullptr) {
  __shared__ int s_scan[RU_T];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  if (d_nq) nq = min(nq, *d_nq);
  for (int i = tid; i < nq; i += RU_T) {
    const int b0 = best[2 * i], b1 = best[2 * i + 1];
    if (b0 < 0 || b1 < 0) continue;
    const float m = sqrtf((float)d2[2 * i]), sd = sqrtf((float)d2[2 * i + 1]);
    if ((double)m < ratio * (double)sd) atomicMin(&owner[b0], i);
  }
  if (tid == 0) s_base = 0;
  __threadfence();
  __syncthreads();
  for (int base = 0; base < nq; base += RU_T) {
    const int i = base + tid;
    int keep = 0, b0 = -1;
    if (i < nq) {
      b0 = best[2 * i];
      const int b1 = best[2 * i + 1];
      if (b0 >= 0 && b1 >= 0) keep = __hip_atomic_load(&owner[b0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i ? 1 : 0;
    }
    s_scan[tid] = keep;
    __syncthreads();
    for (int off = 1; off < RU_T; off <<= 1) {
      const int add = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += add;
      __syncthreads();
    }
    const int pos = s_base + s_scan[tid] - keep;
    if (keep) {
      pairs[2 * pos] = i;
      pairs[2 * pos + 1] = b0;
    }
    __syncthreads();
    if (tid == RU_T - 1) s_base += s_scan[tid];
    __syncthreads();
  }
  if (tid == 0) *n_pairs = s_base;
}

}  // namespace

extern "C" {

// d_best: nq*2 int32 (nearest, second nearest train index, -1 if absent); d_d2: nq*2 float64
int vo_knn2_dev(vo_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt, int D, int32_t* d_best,
                double* d_d2) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_q && d_t && d_best && d_d2 && nq >= 1 && nt >= 1 && D >= 1, "knn2: bad arguments");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    vo_prof_scope ps(ctx, VO_K_MATCH);
    hipLaunchKernelGGL(knn2_f32_kernel, dim3(nq), dim3(MT), 0, ctx->stream, d_q, nq, d_t, nt, D, d_best, d_d2);
  }
  return vo_check_launch(ctx, "knn2_f32_kernel");
}

// Frame-pipeline form (SIFT tracker mode, sift.py:38-54): 128-byte descriptor rows already on the device, the two
// counts too (d_nq, d_nt; the launches are sized for cap_q / cap_t).  Pairs (query, train) in query order -> d_pairs
// (cap_q x 2), their number -> *d_npairs.  Asynchronous on the context's stream; scratch[10..14] of the context.
int vo_match_u8_dev(vo_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, int cap_q, const uint8_t* d_t, const int32_t* d_nt,
                    int cap_t, double ratio, int32_t* d_pairs, int32_t* d_npairs, int row_bytes) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_q && d_nq && d_t && d_nt && d_pairs && d_npairs && cap_q >= 1 && cap_t >= 1, "match_u8_dev: bad arguments");
  VO_REQUIRE(ctx, row_bytes == 128 || row_bytes == 384, "match_u8_dev: rows of 128 (SIFT) or 384 (19x19 patches, padded) bytes");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  vo_buf* s = ctx->scratch;
  const int qblocks = vo_cdiv(cap_q, 32), ttiles = vo_cdiv(cap_t, 32);
  int splits = 1;
  while (qblocks * splits < 512 && ttiles / (splits * 2) >= 4) splits *= 2;
  VO_TRY(vo_ensure(ctx, s[10], (size_t)cap_q * 8));                      // best
  VO_TRY(vo_ensure(ctx, s[11], (size_t)cap_q * 16));                     // d2
  VO_TRY(vo_ensure(ctx, s[12], (size_t)cap_t * 4));                      // owner
  VO_TRY(vo_ensure(ctx, s[13], (size_t)qblocks * splits * 64 * 8));      // partial top-2 lists
  if (ctx->match_arrived.cap < (size_t)qblocks * 4) {
    VO_TRY(vo_ensure(ctx, ctx->match_arrived, (size_t)qblocks * 4));
    VO_HIP_TRY(ctx, hipMemsetAsync(ctx->match_arrived.p, 0, ctx->match_arrived.cap, st));
  }
  VO_HIP_TRY(ctx, hipMemsetAsync(s[12].p, 0x7f, (size_t)cap_t * 4, st));
  VO_HIP_TRY(ctx, hipMemsetAsync(d_npairs, 0, 4, st));
  {
    vo_prof_scope ps(ctx, VO_K_MATCH);
    if (row_bytes == 128)
      hipLaunchKernelGGL(knn2_mfma_kernel<4>, dim3(qblocks, splits), dim3(256), 0, st, d_q, cap_q, d_t, cap_t,
                         (unsigned long long*)s[13].p, (unsigned*)ctx->match_arrived.p, (int*)s[10].p, (double*)s[11].p,
                         (const int*)d_nq, (const int*)d_nt);
    else
      hipLaunchKernelGGL(knn2_mfma_kernel<12>, dim3(qblocks, splits), dim3(256), 0, st, d_q, cap_q, d_t, cap_t,
                         (unsigned long long*)s[13].p, (unsigned*)ctx->match_arrived.p, (int*)s[10].p, (double*)s[11].p,
                         (const int*)d_nq, (const int*)d_nt);
  }
  VO_TRY(vo_check_launch(ctx, "knn2_mfma_kernel"));
  hipLaunchKernelGGL(ratio_unique_kernel, dim3(1), dim3(RU_T), 0, st, (const int*)s[10].p, (const double*)s[11].p, cap_q,
                     ratio, (int*)s[12].p, (int*)d_pairs, (int*)d_npairs, (const int*)d_nq);
  return vo_check_launch(ctx, "ratio_unique_kernel");
}

int vo_match_knn2_ratio(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int D, double ratio,
                        int32_t* pairs, int32_t* n_pairs) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, pairs && n_pairs, "match_knn2_ratio: null pointer");
  *n_pairs = 0;
  VO_REQUIRE(ctx, nq >= 0 && nt >= 0 && D >= 1, "match_knn2_ratio: bad arguments");
  if (nq == 0 || nt == 0) return VO_OK;
  VO_REQUIRE(ctx, q && t, "match_knn2_ratio: null pointer");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  vo_buf* s = ctx->scratch;
  const size_t qbytes = (size_t)nq * D * 4, tbytes = (size_t)nt * D * 4;
  const bool bytes_fit = (size_t)D * 255 * 255 < (1ull << 31);
  const bool mfma = D == 128 || D == 361;               // the reference's two descriptor lengths (sift.py, harris.py)
  const int Dp = mfma ? (D + 31) & ~31 : (D + 3) & ~3;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)nq * Dp));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)nt * Dp));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)nq * 8));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)nq * 16));
  VO_TRY(vo_ensure(ctx, s[5], qbytes));
  VO_TRY(vo_ensure(ctx, s[6], tbytes));
  VO_TRY(vo_ensure(ctx, s[7], (size_t)nt * 4));           // owner of every train index
  VO_TRY(vo_ensure(ctx, s[8], (size_t)nq * 8 + 16));      // pairs, then [n_pairs, not_bytes]
  // descriptors go up as they are (through the pinned staging buffer); packing happens on the device
  VO_TRY(vo_ensure_pinned(ctx, qbytes + tbytes));
  memcpy(ctx->h_pin, q, qbytes);
  memcpy((char*)ctx->h_pin + qbytes, t, tbytes);
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[5].p, ctx->h_pin, qbytes, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[6].p, (char*)ctx->h_pin + qbytes, tbytes, hipMemcpyHostToDevice, st));
  int* d_pairs = (int*)s[8].p;
  int* d_n = d_pairs + (size_t)nq * 2;
  unsigned* d_flag = (unsigned*)(d_n + 1);
  VO_HIP_TRY(ctx, hipMemsetAsync(d_n, 0, 8, st));
  auto filter = [&]() -> int {
    VO_HIP_TRY(ctx, hipMemsetAsync(s[7].p, 0x7f, (size_t)nt * 4, st));     // 0x7f7f7f7f: above every query index
    hipLaunchKernelGGL(ratio_unique_kernel, dim3(1), dim3(RU_T), 0, st, (const int*)s[2].p, (const double*)s[3].p, nq, ratio,
                       (int*)s[7].p, d_pairs, d_n);
    return vo_check_launch(ctx, "ratio_unique_kernel");
  };
  int host[2] = {0, 0};
  bool float_path = !bytes_fit;
  if (bytes_fit) {
    hipLaunchKernelGGL(pack_bytes_kernel, dim3((unsigned)(((size_t)nq * Dp + 255) / 256)), dim3(256), 0, st,
                       (const float*)s[5].p, nq, D, Dp, (uint8_t*)s[0].p, d_flag);
    hipLaunchKernelGGL(pack_bytes_kernel, dim3((unsigned)(((size_t)nt * Dp + 255) / 256)), dim3(256), 0, st,
                       (const float*)s[6].p, nt, D, Dp, (uint8_t*)s[1].p, d_flag);
    VO_TRY(vo_check_launch(ctx, "pack_bytes_kernel"));
    const size_t lds = (size_t)QB * (Dp / 4) * 4 + (size_t)QB * MT * 2 * 8;
    // MFMA form: enough workgroups to fill the chip -- the train set is split while a share keeps >= 4 tiles of 32
    const int qblocks = vo_cdiv(nq, 32), ttiles = vo_cdiv(nt, 32);
    int splits = 1;
    while (qblocks * splits < 512 && ttiles / (splits * 2) >= 4) splits *= 2;
    if (mfma) {
      VO_TRY(vo_ensure(ctx, s[4], (size_t)qblocks * splits * 64 * 8));
      if (ctx->match_arrived.cap < (size_t)qblocks * 4) {     // arrival counters: zero once, the kernel leaves them at zero
        VO_TRY(vo_ensure(ctx, ctx->match_arrived, (size_t)qblocks * 4));
        VO_HIP_TRY(ctx, hipMemsetAsync(ctx->match_arrived.p, 0, ctx->match_arrived.cap, st));
      }
    }
    {
      vo_prof_scope ps(ctx, VO_K_MATCH);
      if (Dp == 128)
        hipLaunchKernelGGL(knn2_mfma_kernel<4>, dim3(qblocks, splits), dim3(256), 0, st, (const uint8_t*)s[0].p, nq,
                           (const uint8_t*)s[1].p, nt, (unsigned long long*)s[4].p, (unsigned*)ctx->match_arrived.p,
                           (int*)s[2].p, (double*)s[3].p);
      else if (mfma)
        hipLaunchKernelGGL(knn2_mfma_kernel<12>, dim3(qblocks, splits), dim3(256), 0, st, (const uint8_t*)s[0].p, nq,
                           (const uint8_t*)s[1].p, nt, (unsigned long long*)s[4].p, (unsigned*)ctx->match_arrived.p,
                           (int*)s[2].p, (double*)s[3].p);
      else
        hipLaunchKernelGGL(knn2_u8_kernel, dim3(vo_cdiv(nq, QB)), dim3(MT), lds, st, (const uint8_t*)s[0].p, nq,
                           (const uint8_t*)s[1].p, nt, Dp, (int*)s[2].p, (double*)s[3].p);
    }
    VO_TRY(vo_check_launch(ctx, "knn2 kernel"));
    VO_TRY(filter());
    VO_HIP_TRY(ctx, hipMemcpyAsync(host, d_n, 8, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
    float_path = host[1] != 0;                             // some value was no byte: the result above does not count
  }
  if (float_path) {
    VO_TRY(vo_knn2_dev(ctx, (const float*)s[5].p, nq, (const float*)s[6].p, nt, D, (int32_t*)s[2].p, (double*)s[3].p));
    VO_TRY(filter());
    VO_HIP_TRY(ctx, hipMemcpyAsync(host, d_n, 4, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  const int n = host[0];
  if (n > 0) {
    VO_HIP_TRY(ctx, hipMemcpyAsync(pairs, d_pairs, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  *n_pairs = n;
  return VO_OK;
}

}  // extern "C"
