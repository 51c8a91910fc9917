// Per-point DLT shared by the batched kernel (dlt.hip) and the device-resident frame state
// (state.hip): one lane solves one 6x4 system, fp64, registers only.
//
// Reference: LandmarksTriangulator._linear_triangulation (src/vo/landmarks/triangulation.py:352-389):
//     A = [ [x1]_x C1 ; [x2]_x C2 ]   (6 x 4),  x = (u, v, 1)
// right singular vector of the smallest singular value, de-homogenised (src/vo/helpers.py:18-28).
// The 6x4 SVD is a one-sided (Hestenes) Jacobi iteration: columns of A are rotated pairwise until
// mutually orthogonal, the same rotations accumulate V; the column of least norm gives the vector.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace vo_dlt {

template <int P, int Q>
__device__ __forceinline__ bool rotate_pair(double (&A)[6][4], double (&V)[4][4]) {
  double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    alpha += A[r][P] * A[r][P];
    beta += A[r][Q] * A[r][Q];
    gamma += A[r][P] * A[r][Q];
  }
  // |gamma| <= 1e-15 sqrt(alpha beta), squared: no square root for the test
  if (gamma == 0.0 || gamma * gamma <= 1e-30 * (alpha * beta)) return false;
  // The rotation of t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = d / g, c = 1 / sqrt(1 + t^2), s = c t, written
  // without zeta and t:  t = sign |g| / w with w = |d| + sqrt(d^2 + g^2), hence c = w / rn and s = sign |g| / rn with
  // rn = sqrt(w^2 + g^2).  Two square roots and ONE division on the dependent chain instead of three and three (the
  // sweeps of the landmark stage are one long chain per candidate: that stage 17.4 -> 15.7 us).
  const double d = beta - alpha, g = 2.0 * gamma;
  const double w = fabs(d) + sqrt(d * d + g * g);
  const double rn = sqrt(w * w + g * g);
  const double c = w / rn;
  const double sm = fabs(g) / rn;
  const double s = (d == 0.0 || (d > 0.0) == (g > 0.0)) ? sm : -sm;
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

// c1, c2: 3x4 row-major camera matrices; (u1, v1), (u2, v2): the two observations; X: the point
__device__ __forceinline__ void triangulate_point(const double* __restrict__ c1, double u1, double v1,
                                                  const double* __restrict__ c2, double u2, double v2, double* X) {
  double A[6][4], V[4][4];
  {
    const double u = u1, v = v1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double r0 = c1[c], r1 = c1[4 + c], r2 = c1[8 + c];
      A[0][c] = v * r2 - r1;      // [x]_x rows: (0,-1,v), (1,0,-u), (-v,u,0)
      A[1][c] = r0 - u * r2;
      A[2][c] = u * r1 - v * r0;
    }
  }
  {
    const double u = u2, v = v2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double r0 = c2[c], r1 = c2[4 + c], r2 = c2[8 + c];
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
  X[0] = p0 / p3;
  X[1] = p1 / p3;
  X[2] = p2 / p3;
}

}  // namespace vo_dlt
