// Brute-force 2-nearest-neighbour descriptor matching with ratio test for gfx950.
//
// Reference call sites: src/vo/features/harris.py:246-262 (raw 19x19 patches, D = 361,
// ratio 0.85) and src/vo/features/sift.py:38-54 (D = 128, ratio 0.8):
//   cv2.BFMatcher().knnMatch(desc1, desc2, k=2), keep m if m.distance < ratio * n.distance
//   and the train index has not been used yet (queries in order).
// Both descriptor kinds are integer-valued in 0..255, so the squared distance
//   |a|^2 + |b|^2 - 2 a.b
// is an exact integer: descriptors are packed to bytes and a.b runs on the packed
// 4-way byte dot product (v_dot4_u32_u8), 4 multiply-adds per lane per instruction.
// Any other input takes the float path (float64 accumulation in index order).  Either
// way the result equals the oracle's definition (oracle/csrc/match.c) bit for bit.
// The first-come uniqueness filter is a sequential pass over nq results (host side).
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
  if (k < k0) {
    k1 = k0;
    k0 = k;
  } else if (k < k1) {
    k1 = k;
  }
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

bool all_bytes(const float* v, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const float f = v[i];
    if (!(f >= 0.f && f <= 255.f) || f != (float)(int)f) return false;
  }
  return true;
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
  std::vector<int32_t> best((size_t)nq * 2);
  std::vector<double> d2((size_t)nq * 2);
  VO_TRY(vo_ensure(ctx, s[2], (size_t)nq * 8));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)nq * 16));
  if (all_bytes(q, (size_t)nq * D) && all_bytes(t, (size_t)nt * D) && (size_t)D * 255 * 255 < (1ull << 31)) {
    const int Dp = (D + 3) & ~3;
    std::vector<uint8_t> qb((size_t)nq * Dp, 0), tb((size_t)nt * Dp, 0);
    for (int i = 0; i < nq; ++i)
      for (int k = 0; k < D; ++k) qb[(size_t)i * Dp + k] = (uint8_t)q[(size_t)i * D + k];
    for (int i = 0; i < nt; ++i)
      for (int k = 0; k < D; ++k) tb[(size_t)i * Dp + k] = (uint8_t)t[(size_t)i * D + k];
    VO_TRY(vo_ensure(ctx, s[0], qb.size()));
    VO_TRY(vo_ensure(ctx, s[1], tb.size()));
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, qb.data(), qb.size(), hipMemcpyHostToDevice, st));
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, tb.data(), tb.size(), hipMemcpyHostToDevice, st));
    const size_t lds = (size_t)QB * (Dp / 4) * 4 + (size_t)QB * MT * 2 * 8;
    {
      vo_prof_scope ps(ctx, VO_K_MATCH);
      hipLaunchKernelGGL(knn2_u8_kernel, dim3(vo_cdiv(nq, QB)), dim3(MT), lds, st, (const uint8_t*)s[0].p, nq,
                         (const uint8_t*)s[1].p, nt, Dp, (int*)s[2].p, (double*)s[3].p);
    }
    VO_TRY(vo_check_launch(ctx, "knn2_u8_kernel"));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));   // host staging vectors go out of scope below
  } else {
    VO_TRY(vo_ensure(ctx, s[0], (size_t)nq * D * 4));
    VO_TRY(vo_ensure(ctx, s[1], (size_t)nt * D * 4));
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, q, (size_t)nq * D * 4, hipMemcpyHostToDevice, st));
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, t, (size_t)nt * D * 4, hipMemcpyHostToDevice, st));
    VO_TRY(vo_knn2_dev(ctx, (const float*)s[0].p, nq, (const float*)s[1].p, nt, D, (int32_t*)s[2].p, (double*)s[3].p));
  }
  VO_HIP_TRY(ctx, hipMemcpyAsync(best.data(), s[2].p, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(d2.data(), s[3].p, (size_t)nq * 16, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  // ratio test + first-come uniqueness, queries in order (harris.py:250-258, sift.py:45-52)
  std::vector<uint8_t> used((size_t)nt, 0);
  int n = 0;
  for (int i = 0; i < nq; ++i) {
    const int b0 = best[2 * i], b1 = best[2 * i + 1];
    if (b0 < 0 || b1 < 0) continue;
    const float m = sqrtf((float)d2[2 * i]), sd = sqrtf((float)d2[2 * i + 1]);
    if ((double)m < ratio * (double)sd && !used[b0]) {
      pairs[2 * n] = i;
      pairs[2 * n + 1] = b0;
      used[b0] = 1;
      ++n;
    }
  }
  *n_pairs = n;
  return VO_OK;
}

}  // extern "C"
