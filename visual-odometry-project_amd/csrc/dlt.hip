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
#include "dlt_device.h"

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(128) void dlt_kernel(const double* __restrict__ x1, const double* __restrict__ x2, int n_arg,
                                                  const int* __restrict__ d_n,
                                                  const double* __restrict__ C1, int c1_per_point,
                                                  const double* __restrict__ C2, double* __restrict__ Xout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = d_n ? min(*d_n, n_arg) : n_arg;   // point count given, or read on the device (pipeline)
  if (i >= n) return;
  const double* c1 = C1 + (c1_per_point ? (size_t)12 * i : 0);
  double X[3];
  vo_dlt::triangulate_point(c1, x1[2 * i], x1[2 * i + 1], C2, x2[2 * i], x2[2 * i + 1], X);
  Xout[3 * i] = X[0];
  Xout[3 * i + 1] = X[1];
  Xout[3 * i + 2] = X[2];
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
