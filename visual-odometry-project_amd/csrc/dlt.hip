// Batched DLT triangulation for gfx950: one lane per point, fp64.
//
// Reference behaviour: LandmarksTriangulator._linear_triangulation
// (src/vo/landmarks/triangulation.py:352-389) builds, per point,
//     A = [ [x1]_x C1 ; [x2]_x C2 ]   (6 x 4),  x = (u, v, 1)
// takes the right singular vector of the smallest singular value (LAPACK SVD) and
// de-homogenises it (src/vo/helpers.py:18-28).  triangulate_candidates
// (triangulation.py:38-86) calls it with one C1 per point.
// Here the 6x4 SVD is a one-sided (Hestenes) Jacobi iteration held in registers:
// columns of A are rotated pairwise until mutually orthogonal, the same rotations
// accumulate V; the column of least norm gives the singular vector.  Results agree
// with LAPACK to rounding (parity by tolerance).
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

template <int P, int Q>
__device__ __forceinline__ bool rotate_pair(double (&A)[6][4], double (&V)[4][4]) {
  double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    alpha += A[r][P] * A[r][P];
    beta += A[r][Q] * A[r][Q];
    gamma += A[r][P] * A[r][Q];
  }
  if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) return false;
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + tt * tt);
  const double s = c * tt;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double ap = A[r][P], aq = A[r][Q];
    A[r][P] = c * ap - s * aq;
    A[r][Q] = s * ap + c * aq;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double vp = V[r][P], vq = V[r][Q];
    V[r][P] = c * vp - s * vq;
    V[r][Q] = s * vp + c * vq;
  }
  return true;
}

__global__ __launch_bounds__(128) void dlt_kernel(const double* __restrict__ x1, const double* __restrict__ x2, int n_arg,
                                                  const int* __restrict__ d_n,
                                                  const double* __restrict__ C1, int c1_per_point,
                                                  const double* __restrict__ C2, double* __restrict__ Xout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = d_n ? min(*d_n, n_arg) : n_arg;   // point count given, or read on the device (pipeline)
  if (i >= n) return;
  const double* c1 = C1 + (c1_per_point ? (size_t)12 * i : 0);
  double A[6][4], V[4][4];
  {
    const double u = x1[2 * i], v = x1[2 * i + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double r0 = c1[c], r1 = c1[4 + c], r2 = c1[8 + c];
      A[0][c] = v * r2 - r1;      // [x]_x rows: (0,-1,v), (1,0,-u), (-v,u,0)
      A[1][c] = r0 - u * r2;
      A[2][c] = u * r1 - v * r0;
    }
  }
  {
    const double u = x2[2 * i], v = x2[2 * i + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double r0 = C2[c], r1 = C2[4 + c], r2 = C2[8 + c];
      A[3][c] = v * r2 - r1;
      A[4][c] = r0 - u * r2;
      A[5][c] = u * r1 - v * r0;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;

  for (int sweep = 0; sweep < 30; ++sweep) {
    bool any = false;
    any |= rotate_pair<0, 1>(A, V);
    any |= rotate_pair<0, 2>(A, V);
    any |= rotate_pair<0, 3>(A, V);
    any |= rotate_pair<1, 2>(A, V);
    any |= rotate_pair<1, 3>(A, V);
    any |= rotate_pair<2, 3>(A, V);
    if (!any) break;
  }
  double nrm[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += A[r][c] * A[r][c];
    nrm[c] = s;
  }
  // column of least norm (static indexing keeps V in registers)
  double best = nrm[0];
  double p0 = V[0][0], p1 = V[1][0], p2 = V[2][0], p3 = V[3][0];
#pragma unroll
  for (int c = 1; c < 4; ++c) {
    if (nrm[c] < best) {
      best = nrm[c];
      p0 = V[0][c];
      p1 = V[1][c];
      p2 = V[2][c];
      p3 = V[3][c];
    }
  }
  Xout[3 * i] = p0 / p3;
  Xout[3 * i + 1] = p1 / p3;
  Xout[3 * i + 2] = p2 / p3;
}

}  // namespace

// Pipeline-internal form: at most n_cap points, the actual count is read from *d_n when the
// kernel runs (so it can be enqueued, or captured in a graph, before the count exists).
int vo_triangulate_dlt_ndev(vo_ctx* ctx, const double* d_x1, const double* d_x2, const int32_t* d_n, int n_cap,
                            const double* d_C1, const double* d_C2, double* d_X) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_x1 && d_x2 && d_n && d_C1 && d_C2 && d_X && n_cap >= 1, "triangulate_dlt_n: bad arguments");
  {
    vo_prof_scope ps(ctx, VO_K_DLT);
    hipLaunchKernelGGL(dlt_kernel, dim3(vo_cdiv(n_cap, 128)), dim3(128), 0, ctx->stream, d_x1, d_x2, n_cap, d_n, d_C1,
                       0, d_C2, d_X);
  }
  return vo_check_launch(ctx, "dlt_kernel");
}

extern "C" {

int vo_triangulate_dlt_dev(vo_ctx* ctx, const double* d_x1, const double* d_x2, int n, const double* d_C1,
                           int c1_per_point, const double* d_C2, double* d_X) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, n >= 0, "triangulate_dlt: bad n");
  if (n == 0) return VO_OK;
  VO_REQUIRE(ctx, d_x1 && d_x2 && d_C1 && d_C2 && d_X, "triangulate_dlt: null pointer");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    vo_prof_scope ps(ctx, VO_K_DLT);
    hipLaunchKernelGGL(dlt_kernel, dim3(vo_cdiv(n, 128)), dim3(128), 0, ctx->stream, d_x1, d_x2, n,
                       (const int*)nullptr, d_C1, c1_per_point ? 1 : 0, d_C2, d_X);
  }
  return vo_check_launch(ctx, "dlt_kernel");
}

int vo_triangulate_dlt(vo_ctx* ctx, const double* x1, const double* x2, int n, const double* C1, int c1_per_point,
                       const double* C2, double* X) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, n >= 0, "triangulate_dlt: bad n");
  if (n == 0) return VO_OK;
  VO_REQUIRE(ctx, x1 && x2 && C1 && C2 && X, "triangulate_dlt: null pointer");
  vo_buf* s = ctx->scratch;
  const size_t c1b = c1_per_point ? (size_t)n * 96 : 96;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)n * 16));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)n * 16));
  VO_TRY(vo_ensure(ctx, s[2], c1b));
  VO_TRY(vo_ensure(ctx, s[3], 96));
  VO_TRY(vo_ensure(ctx, s[4], (size_t)n * 24));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, x1, (size_t)n * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, x2, (size_t)n * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, C1, c1b, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[3].p, C2, 96, hipMemcpyHostToDevice, st));
  VO_TRY(vo_triangulate_dlt_dev(ctx, (const double*)s[0].p, (const double*)s[1].p, n, (const double*)s[2].p,
                                c1_per_point, (const double*)s[3].p, (double*)s[4].p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(X, s[4].p, (size_t)n * 24, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

}  // extern "C"
