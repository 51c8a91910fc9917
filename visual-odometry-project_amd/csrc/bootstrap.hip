// Two-view bootstrap (reference: src/vo/landmarks/triangulation.py:110-350, src/vo/helpers.py:31-54), the step
// before the per-frame loop (src/main.py:204-230), as four kernels:
//
//   f8_hyp_kernel      one lane per RANSAC sample of 8 correspondences: Kronecker rows (triangulation.py:203-205), the
//                      null vector of the 8x9 system (np.linalg.svd(Q)[2][-1], :209-212) by a one-sided (Hestenes) Jacobi
//                      iteration on its columns -- the sample's rows in registers, the accumulated rotations in LDS --
//                      and the rank-2 projection (:214-217: the smallest singular triplet of the 3x3 removed)
//   f_score_kernel     every hypothesis against every correspondence: algebraic error (p2^T F p1)^2 (:140-145) or the
//                      squared distance to the epipolar lines in both images; strict `<` threshold; counts + mask rows
//   f8_fit_kernel      the 8-point fit over ALL (masked) correspondences (RANSAC's closing model_fn(population[inliers]),
//                      src/vo/algorithms/ransac.py:123-127; _find_fundamental_matrix :165-222): Hartley normalisation,
//                      the 9x9 normal matrix by a workgroup reduction, its smallest eigenvector by Jacobi, rank 2
//   relative_pose_kernel  E = K2^T F K1, its SVD (3x3 Jacobi), the four [R | +-T] (:245-277), the four cheirality
//                      votes over the inliers by DLT (:313-332), the winner's triangulation of all points (:334-350)
//
// fp64 throughout, no FMA contraction.  Parity: tests/golden/bootstrap.npz (the reference's own NumPy route) -- F up to
// scale / sign, the RANSAC trace's inlier mask exactly, M, landmarks to 1e-9 / 1e-6.
#include "dlt_device.h"
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int HB = 64;        // hypotheses (lanes) per workgroup of f8_hyp_kernel

// ---- 3x3 symmetric eigenproblem in registers (cyclic Jacobi, unrolled): A -> diagonal, V its eigenvectors (columns) ----
template <int P, int Q>
__device__ __forceinline__ void sym3_rotate(double (&A)[3][3], double (&V)[3][3]) {
  const double apq = A[P][Q];
  if (apq == 0.0) return;
  const double theta = (A[Q][Q] - A[P][P]) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  constexpr int R = 3 - P - Q;
  const double arp = A[R][P], arq = A[R][Q];
  A[R][P] = A[P][R] = c * arp - s * arq;
  A[R][Q] = A[Q][R] = s * arp + c * arq;
  A[P][P] = A[P][P] - t * apq;
  A[Q][Q] = A[Q][Q] + t * apq;
  A[P][Q] = A[Q][P] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double vp = V[k][P], vq = V[k][Q];
    V[k][P] = c * vp - s * vq;
    V[k][Q] = s * vp + c * vq;
  }
}

__device__ __forceinline__ void sym3_eig(double (&A)[3][3], double (&V)[3][3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) V[r][c] = r == c ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-34 * dia || off == 0.0) break;
    sym3_rotate<0, 1>(A, V);
    sym3_rotate<0, 2>(A, V);
    sym3_rotate<1, 2>(A, V);
  }
}

// F (row-major 3x3) -> F with its smallest singular value set to zero (triangulation.py:214-217):
// F' = F - (F v3) v3^T with v3 the eigenvector of F^T F to its smallest eigenvalue
__device__ __forceinline__ void rank2(double* F) {
  double G[3][3], V[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) G[a][b] = F[a] * F[b] + F[3 + a] * F[3 + b] + F[6 + a] * F[6 + b];
  sym3_eig(G, V);
  int m = 0;
  if (G[1][1] < G[m][m]) m = 1;
  if (G[2][2] < G[m][m]) m = 2;
  double v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = m == 0 ? V[k][0] : (m == 1 ? V[k][1] : V[k][2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double fv = F[3 * r] * v[0] + F[3 * r + 1] * v[1] + F[3 * r + 2] * v[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) F[3 * r + c] = F[3 * r + c] - fv * v[c];
  }
}

// Hartley normalisation of n points given their sums (helpers.py:31-54): s = sqrt(2) / sqrt(mean |p - mu|^2)
struct hartley {
  double s, tx, ty;        // x' = s x + tx, y' = s y + ty
};

// T2^T F T1 for T = [[s, 0, tx], [0, s, ty], [0, 0, 1]]
__device__ __forceinline__ void denormalise(double* F, const hartley& h1, const hartley& h2) {
  double A[9];             // F T1
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    A[3 * r] = F[3 * r] * h1.s;
    A[3 * r + 1] = F[3 * r + 1] * h1.s;
    A[3 * r + 2] = F[3 * r] * h1.tx + F[3 * r + 1] * h1.ty + F[3 * r + 2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    F[c] = h2.s * A[c];
    F[3 + c] = h2.s * A[3 + c];
    F[6 + c] = h2.tx * A[c] + h2.ty * A[3 + c] + A[6 + c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// one lane = one sample of 8 correspondences -> F (9 doubles, row-major)
__global__ __launch_bounds__(HB) void f8_hyp_kernel(const double* __restrict__ p1, const double* __restrict__ p2, int N,
                                                    const int* __restrict__ samples, int Hyp, int normalize,
                                                    double* __restrict__ Fout) {
  __shared__ double s_V[81][HB];            // accumulated rotations, [entry][lane]
  const int lane = threadIdx.x;
  const int h = blockIdx.x * HB + lane;
  if (h >= Hyp) return;
  double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = min(max(samples[8 * h + k], 0), N - 1);
    x1[k] = p1[2 * i];
    y1[k] = p1[2 * i + 1];
    x2[k] = p2[2 * i];
    y2[k] = p2[2 * i + 1];
  }
  hartley h1 = {1.0, 0.0, 0.0}, h2 = {1.0, 0.0, 0.0};
  if (normalize) {
    auto fit = [](const double* x, const double* y) {
      double mx = 0.0, my = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        mx += x[k];
        my += y[k];
      }
      mx = mx / 8.0;
      my = my / 8.0;
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += (x[k] - mx) * (x[k] - mx) + (y[k] - my) * (y[k] - my);
      const double s = sqrt(2.0) / sqrt(v / 8.0);
      hartley r = {s, -s * mx, -s * my};
      return r;
    };
    h1 = fit(x1, y1);
    h2 = fit(x2, y2);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x1[k] = h1.s * x1[k] + h1.tx;
      y1[k] = h1.s * y1[k] + h1.ty;
      x2[k] = h2.s * x2[k] + h2.tx;
      y2[k] = h2.s * y2[k] + h2.ty;
    }
  }
  // Q[k] = kron(p1_k, p2_k): entry 3a + b = p1[a] p2[b]
  double Q[8][9];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double a[3] = {x1[k], y1[k], 1.0}, b[3] = {x2[k], y2[k], 1.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Q[k][3 * i + j] = a[i] * b[j];
  }
#pragma unroll
  for (int e = 0; e < 81; ++e) s_V[e][lane] = (e / 9 == e % 9) ? 1.0 : 0.0;
  // one-sided Jacobi: rotate column pairs until mutually orthogonal; the 8x9 system has a null space, whose
  // direction ends up in the column of (near-)zero norm
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool any = false;
#pragma unroll
    for (int P = 0; P < 8; ++P) {
#pragma unroll
      for (int Qc = P + 1; Qc < 9; ++Qc) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          alpha += Q[r][P] * Q[r][P];
          beta += Q[r][Qc] * Q[r][Qc];
          gamma += Q[r][P] * Q[r][Qc];
        }
        if (gamma == 0.0 || gamma * gamma <= 1e-30 * (alpha * beta)) continue;
        any = true;
        const double d = beta - alpha, g = 2.0 * gamma;
        const double w = fabs(d) + sqrt(d * d + g * g);
        const double rn = sqrt(w * w + g * g);
        const double c = w / rn;
        const double sm = fabs(g) / rn;
        const double s = (d == 0.0 || (d > 0.0) == (g > 0.0)) ? sm : -sm;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double ap = Q[r][P], aq = Q[r][Qc];
          Q[r][P] = c * ap - s * aq;
          Q[r][Qc] = s * ap + c * aq;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const double vp = s_V[9 * r + P][lane], vq = s_V[9 * r + Qc][lane];
          s_V[9 * r + P][lane] = c * vp - s * vq;
          s_V[9 * r + Qc][lane] = s * vp + c * vq;
        }
      }
    }
    if (!any) break;
  }
  int m = 0;
  double best = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    double nrm = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) nrm += Q[r][c] * Q[r][c];
    if (c == 0 || nrm < best) {
      best = nrm;
      m = c;
    }
  }
  // F = Vh[-1].reshape(3, 3).T (stacked column-wise, triangulation.py:212): F[b][a] = f[3a + b]
  double F[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) F[3 * b + a] = s_V[9 * (3 * a + b) + m][lane];
  rank2(F);
  if (normalize) denormalise(F, h1, h2);
#pragma unroll
  for (int k = 0; k < 9; ++k) Fout[(size_t)9 * h + k] = F[k];
}

// error of one correspondence under F.  kind 0: (p2^T F p1)^2 in NumPy's order ((p2^T F) p1, triangulation.py:140-145);
// kind 1: max of the squared distances to the epipolar lines in both images
__device__ __forceinline__ double f_error(const double* F, double x1, double y1, double x2, double y2, int kind) {
  if (kind == 0) {
    const double t0 = x2 * F[0] + y2 * F[3] + F[6];
    const double t1 = x2 * F[1] + y2 * F[4] + F[7];
    const double t2 = x2 * F[2] + y2 * F[5] + F[8];
    const double r = t0 * x1 + t1 * y1 + t2;
    return r * r;
  }
  const double l2x = F[0] * x1 + F[1] * y1 + F[2], l2y = F[3] * x1 + F[4] * y1 + F[5], l2z = F[6] * x1 + F[7] * y1 + F[8];
  const double l1x = F[0] * x2 + F[3] * y2 + F[6], l1y = F[1] * x2 + F[4] * y2 + F[7];
  const double e = x2 * l2x + y2 * l2y + l2z;
  const double num = e * e;
  const double d2 = num / (l2x * l2x + l2y * l2y), d1 = num / (l1x * l1x + l1y * l1y);
  return d1 > d2 ? d1 : d2;                  // (np.maximum)
}

// grid.x = hypothesis; one workgroup walks the correspondences: count + mask row (words 64-bit words per row)
__global__ __launch_bounds__(256) void f_score_kernel(const double* __restrict__ p1, const double* __restrict__ p2, int N,
                                                      const double* __restrict__ Fs, int kind, double thr,
                                                      int* __restrict__ counts, unsigned long long* __restrict__ masks,
                                                      int words) {
  __shared__ int s_cnt[4];
  const int h = blockIdx.x, tid = threadIdx.x;
  double F[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) F[k] = Fs[(size_t)9 * h + k];
  int cnt = 0;
  for (int base = 0; base < N; base += 256) {
    const int i = base + tid;
    bool in = false;
    if (i < N) in = f_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], kind) < thr;
    const unsigned long long b = __ballot(in);
    if ((tid & 63) == 0) {
      if (masks && (base + tid) / 64 < words) masks[(size_t)h * words + (base + tid) / 64] = b;
      cnt += __popcll(b);
    }
  }
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) counts[h] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---------------------------------------------------------------------------------------------------------------
// 9x9 symmetric Jacobi in LDS (one work item): returns the eigenvector to the smallest eigenvalue in out[9]
__device__ void sym9_min_eigvec(double* A /*81*/, double* V /*81*/, double* out) {
  for (int e = 0; e < 81; ++e) V[e] = (e / 9 == e % 9) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) {
        const double v = A[9 * r + c];
        if (r == c) dia += v * v;
        else off += v * v;
      }
    if (off == 0.0 || off <= 1e-34 * dia) break;
    for (int p = 0; p < 8; ++p)
      for (int q = p + 1; q < 9; ++q) {
        const double apq = A[9 * p + q];
        if (apq == 0.0) continue;
        const double theta = (A[9 * q + q] - A[9 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 9; ++k) {
          if (k == p || k == q) continue;
          const double akp = A[9 * k + p], akq = A[9 * k + q];
          const double np_ = c * akp - s * akq, nq = s * akp + c * akq;
          A[9 * k + p] = A[9 * p + k] = np_;
          A[9 * k + q] = A[9 * q + k] = nq;
        }
        A[9 * p + p] = A[9 * p + p] - t * apq;
        A[9 * q + q] = A[9 * q + q] + t * apq;
        A[9 * p + q] = A[9 * q + p] = 0.0;
        for (int k = 0; k < 9; ++k) {
          const double vp = V[9 * k + p], vq = V[9 * k + q];
          V[9 * k + p] = c * vp - s * vq;
          V[9 * k + q] = s * vp + c * vq;
        }
      }
  }
  int m = 0;
  for (int c = 1; c < 9; ++c)
    if (A[9 * c + c] < A[9 * m + m]) m = c;
  for (int k = 0; k < 9; ++k) out[k] = V[9 * k + m];
}

__device__ __forceinline__ double block_sum(double v, double* s_red /*[4]*/, int tid) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// the 8-point fit over the points whose mask byte is set (all if mask == NULL); one workgroup
__global__ __launch_bounds__(256) void f8_fit_kernel(const double* __restrict__ p1, const double* __restrict__ p2, int N,
                                                     const uint8_t* __restrict__ mask, int normalize,
                                                     double* __restrict__ Fout, int* __restrict__ n_used) {
  __shared__ double s_red[4];
  __shared__ double s_A[81], s_V[81], s_f[9];
  const int tid = threadIdx.x;
  hartley h1 = {1.0, 0.0, 0.0}, h2 = {1.0, 0.0, 0.0};
  double cnt = 0.0;
  for (int i = tid; i < N; i += 256) cnt += (!mask || mask[i]) ? 1.0 : 0.0;
  const double n = block_sum(cnt, s_red, tid);
  if (normalize) {
    double sx1 = 0, sy1 = 0, sx2 = 0, sy2 = 0;
    for (int i = tid; i < N; i += 256)
      if (!mask || mask[i]) {
        sx1 += p1[2 * i];
        sy1 += p1[2 * i + 1];
        sx2 += p2[2 * i];
        sy2 += p2[2 * i + 1];
      }
    const double mx1 = block_sum(sx1, s_red, tid) / n, my1 = block_sum(sy1, s_red, tid) / n;
    const double mx2 = block_sum(sx2, s_red, tid) / n, my2 = block_sum(sy2, s_red, tid) / n;
    double v1 = 0, v2 = 0;
    for (int i = tid; i < N; i += 256)
      if (!mask || mask[i]) {
        const double a = p1[2 * i] - mx1, b = p1[2 * i + 1] - my1, c = p2[2 * i] - mx2, d = p2[2 * i + 1] - my2;
        v1 += a * a + b * b;
        v2 += c * c + d * d;
      }
    const double s1 = sqrt(2.0) / sqrt(block_sum(v1, s_red, tid) / n), s2 = sqrt(2.0) / sqrt(block_sum(v2, s_red, tid) / n);
    h1 = {s1, -s1 * mx1, -s1 * my1};
    h2 = {s2, -s2 * mx2, -s2 * my2};
  }
  // Q^T Q, upper triangle: 45 sums
  double acc[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
  for (int i = tid; i < N; i += 256) {
    if (mask && !mask[i]) continue;
    const double a[3] = {h1.s * p1[2 * i] + h1.tx, h1.s * p1[2 * i + 1] + h1.ty, 1.0};
    const double b[3] = {h2.s * p2[2 * i] + h2.tx, h2.s * p2[2 * i + 1] + h2.ty, 1.0};
    double q[9];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) q[3 * u + v] = a[u] * b[v];
    int k = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) acc[k++] += q[r] * q[c];
  }
  {
    int k = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) {
        const double v = block_sum(acc[k++], s_red, tid);
        if (tid == 0) s_A[9 * r + c] = s_A[9 * c + r] = v;
      }
  }
  __syncthreads();
  if (tid == 0) {
    sym9_min_eigvec(s_A, s_V, s_f);
    double F[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) F[3 * b + a] = s_f[3 * a + b];
    rank2(F);
    if (normalize) denormalise(F, h1, h2);
#pragma unroll
    for (int k = 0; k < 9; ++k) Fout[k] = F[k];
    if (n_used) *n_used = (int)n;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// E -> the four [R | +-T] (triangulation.py:245-277).  U, Vh of E's SVD from the eigenvectors of E^T E; the four
// candidates are the same set whatever signs an SVD routine picks, their ORDER may differ from LAPACK's.
__device__ void decompose_essential(const double* E, double* M4 /*4 x 12*/) {
  double G[3][3], V[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) G[a][b] = E[a] * E[b] + E[3 + a] * E[3 + b] + E[6 + a] * E[6 + b];
  sym3_eig(G, V);
  // order the eigenvalues descending: i0 >= i1 >= i2
  int i0 = 0, i2 = 0;
  double ev[3] = {G[0][0], G[1][1], G[2][2]};
  for (int k = 1; k < 3; ++k) {
    if (ev[k] > ev[i0]) i0 = k;
    if (ev[k] < ev[i2]) i2 = k;
  }
  if (i0 == i2) {            // all equal
    i0 = 0;
    i2 = 2;
  }
  const int i1 = 3 - i0 - i2;
  auto col = [&](int c, double* v) {
    for (int k = 0; k < 3; ++k) v[k] = c == 0 ? V[k][0] : (c == 1 ? V[k][1] : V[k][2]);
  };
  double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
  col(i0, v1);
  col(i1, v2);
  // v3 = v1 x v2 (a right-handed V; E's third singular value is ~0, its direction is the cross product's)
  v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
  v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
  v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
  auto unit_Ev = [&](const double* v, double* u) {
    for (int r = 0; r < 3; ++r) u[r] = E[3 * r] * v[0] + E[3 * r + 1] * v[1] + E[3 * r + 2] * v[2];
    const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int r = 0; r < 3; ++r) u[r] = u[r] / n;
  };
  unit_Ev(v1, u1);
  unit_Ev(v2, u2);
  // u2 made orthogonal to u1 (they are to rounding), u3 = u1 x u2
  {
    const double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int r = 0; r < 3; ++r) u2[r] = u2[r] - d * u1[r];
    const double n = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int r = 0; r < 3; ++r) u2[r] = u2[r] / n;
  }
  u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
  u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
  u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
  // R0 = U W Vh = u2 v1^T - u1 v2^T + u3 v3^T;  R1 = U W^T Vh = -u2 v1^T + u1 v2^T + u3 v3^T.  U and V are both
  // right-handed here, so both have determinant +1 (the reference negates the ones that come out with -1).
  for (int j = 0; j < 2; ++j) {
    const double sg = j == 0 ? 1.0 : -1.0;
    for (int i = 0; i < 2; ++i) {
      double* M = M4 + 12 * (2 * i + j);
      const double st = i == 0 ? 1.0 : -1.0;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) M[4 * r + c] = sg * (u2[r] * v1[c] - u1[r] * v2[c]) + u3[r] * v3[c];
        M[4 * r + 3] = st * u3[r];
      }
    }
  }
}

__global__ void essential_decompose_kernel(const double* __restrict__ E, double* __restrict__ M4) {
  if (threadIdx.x == 0 && blockIdx.x == 0) decompose_essential(E, M4);
}

// one workgroup: E from F, the four candidates, the cheirality votes over the inliers, the final triangulation
__global__ __launch_bounds__(256) void relative_pose_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                            int N, const uint8_t* __restrict__ inl, const double* __restrict__ F,
                                                            vo_cam2 cams, double* __restrict__ Mout, double* __restrict__ X,
                                                            uint8_t* __restrict__ mask_out, double* __restrict__ M4out) {
  __shared__ double s_M4[48];
  __shared__ int s_votes[4];
  __shared__ int s_best;
  const int tid = threadIdx.x;
  if (tid == 0) {
    // E = K2^T F K1 (triangulation.py:238-243)
    double A[9], E[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) A[3 * r + c] = F[3 * r] * cams.K1[c] + F[3 * r + 1] * cams.K1[3 + c] + F[3 * r + 2] * cams.K1[6 + c];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) E[3 * r + c] = cams.K2[r] * A[c] + cams.K2[3 + r] * A[3 + c] + cams.K2[6 + r] * A[6 + c];
    decompose_essential(E, s_M4);
    for (int m = 0; m < 4; ++m) s_votes[m] = 0;
  }
  __syncthreads();
  if (M4out && tid < 48) M4out[tid] = s_M4[tid];
  double C1[12];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) C1[4 * r + c] = cams.K1[3 * r + c];     // K1 [I | 0]
    C1[4 * r + 3] = 0.0;
  }
  auto project2 = [&](const double* M, double* C2) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) C2[4 * r + c] = cams.K2[3 * r] * M[c] + cams.K2[3 * r + 1] * M[4 + c] + cams.K2[3 * r + 2] * M[8 + c];
  };
  for (int m = 0; m < 4; ++m) {
    double M[12], C2[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = s_M4[12 * m + k];
    project2(M, C2);
    int votes = 0;
    for (int i = tid; i < N; i += 256) {
      if (inl && !inl[i]) continue;
      double P[3];
      vo_dlt::triangulate_point(C1, x1[2 * i], x1[2 * i + 1], C2, x2[2 * i], x2[2 * i + 1], P);
      const double z2 = M[8] * P[0] + M[9] * P[1] + M[10] * P[2] + M[11];
      votes += (P[2] >= 0.0 && z2 >= 0.0) ? 1 : 0;
    }
    const unsigned long long dummy = 0ull;
    (void)dummy;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) votes += __shfl_xor(votes, off);
    if ((tid & 63) == 0 && votes) atomicAdd(&s_votes[m], votes);
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int m = 1; m < 4; ++m)
      if (s_votes[m] > s_votes[best]) best = m;        // strict `>`: the first of equals (triangulation.py:326)
    s_best = best;
  }
  __syncthreads();
  double M[12], C2[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) M[k] = s_M4[12 * s_best + k];
  project2(M, C2);
  if (tid < 12) Mout[tid] = M[tid];
  for (int i = tid; i < N; i += 256) {
    double P[3];
    vo_dlt::triangulate_point(C1, x1[2 * i], x1[2 * i + 1], C2, x2[2 * i], x2[2 * i + 1], P);
    X[3 * i] = P[0];
    X[3 * i + 1] = P[1];
    X[3 * i + 2] = P[2];
    if (mask_out) {
      const double z2 = M[8] * P[0] + M[9] * P[1] + M[10] * P[2] + M[11];
      mask_out[i] = ((!inl || inl[i]) && P[2] >= 0.0 && z2 >= 0.0) ? 1 : 0;
    }
  }
}

}  // namespace

extern "C" {

int vo_fundamental_hypotheses(vo_ctx* ctx, const double* p1, const double* p2, int N, const int32_t* samples, int Hyp,
                              int normalize_samples, int error_kind, double threshold, double* F, int32_t* counts,
                              uint64_t* masks) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, p1 && p2 && samples && F && counts, "fundamental_hypotheses: null pointer");
  VO_REQUIRE(ctx, N >= 8 && Hyp >= 1 && Hyp <= (1 << 20), "fundamental_hypotheses: need N >= 8, 1 <= Hyp <= 2^20");
  VO_REQUIRE(ctx, error_kind == 0 || error_kind == 1, "fundamental_hypotheses: error_kind must be 0 or 1");
  for (size_t k = 0; k < (size_t)8 * Hyp; ++k)
    VO_REQUIRE(ctx, samples[k] >= 0 && samples[k] < N, "fundamental_hypotheses: sample index %d out of range", (int)samples[k]);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int words = vo_cdiv(N, 64);
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)Hyp * 32));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)Hyp * 72));
  VO_TRY(vo_ensure(ctx, s[4], (size_t)Hyp * 4));
  VO_TRY(vo_ensure(ctx, s[5], (size_t)Hyp * words * 8));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, p1, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, p2, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, samples, (size_t)Hyp * 32, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(f8_hyp_kernel, dim3(vo_cdiv(Hyp, HB)), dim3(HB), 0, st, (const double*)s[0].p, (const double*)s[1].p, N,
                     (const int*)s[2].p, Hyp, normalize_samples, (double*)s[3].p);
  VO_TRY(vo_check_launch(ctx, "f8_hyp_kernel"));
  hipLaunchKernelGGL(f_score_kernel, dim3(Hyp), dim3(256), 0, st, (const double*)s[0].p, (const double*)s[1].p, N,
                     (const double*)s[3].p, error_kind, threshold, (int*)s[4].p,
                     masks ? (unsigned long long*)s[5].p : nullptr, words);
  VO_TRY(vo_check_launch(ctx, "f_score_kernel"));
  VO_HIP_TRY(ctx, hipMemcpyAsync(F, s[3].p, (size_t)Hyp * 72, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(counts, s[4].p, (size_t)Hyp * 4, hipMemcpyDeviceToHost, st));
  if (masks) VO_HIP_TRY(ctx, hipMemcpyAsync(masks, s[5].p, (size_t)Hyp * words * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_fundamental_fit(vo_ctx* ctx, const double* p1, const double* p2, int N, const uint8_t* mask, int normalize,
                       double* F) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, p1 && p2 && F, "fundamental_fit: null pointer");
  VO_REQUIRE(ctx, N >= 8, "fundamental_fit: the 8-point algorithm needs 8 correspondences");
  if (mask) {
    int n = 0;
    for (int i = 0; i < N; ++i) n += mask[i] ? 1 : 0;
    VO_REQUIRE(ctx, n >= 8, "fundamental_fit: only %d correspondences are selected", n);
  }
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)N));
  VO_TRY(vo_ensure(ctx, s[3], 128));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, p1, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, p2, (size_t)N * 16, hipMemcpyHostToDevice, st));
  if (mask) VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, mask, (size_t)N, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(f8_fit_kernel, dim3(1), dim3(256), 0, st, (const double*)s[0].p, (const double*)s[1].p, N,
                     mask ? (const uint8_t*)s[2].p : nullptr, normalize, (double*)s[3].p, (int*)nullptr);
  VO_TRY(vo_check_launch(ctx, "f8_fit_kernel"));
  VO_HIP_TRY(ctx, hipMemcpyAsync(F, s[3].p, 72, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_essential_decompose(vo_ctx* ctx, const double* E, double* M4) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, E && M4, "essential_decompose: null pointer");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], 512));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, E, 72, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(essential_decompose_kernel, dim3(1), dim3(64), 0, st, (const double*)s[0].p, (double*)s[0].p + 16);
  VO_TRY(vo_check_launch(ctx, "essential_decompose_kernel"));
  VO_HIP_TRY(ctx, hipMemcpyAsync(M4, (double*)s[0].p + 16, 48 * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_relative_pose(vo_ctx* ctx, const double* x1, const double* x2, int N, const uint8_t* inliers, const double* K1,
                     const double* K2, const double* F, double* M, double* X, uint8_t* mask_out, double* M4) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, x1 && x2 && K1 && K2 && F && M && X, "relative_pose: null pointer");
  VO_REQUIRE(ctx, N >= 1, "relative_pose: no correspondences");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)N));
  VO_TRY(vo_ensure(ctx, s[3], 1024));
  VO_TRY(vo_ensure(ctx, s[4], (size_t)N * 24));
  VO_TRY(vo_ensure(ctx, s[5], (size_t)N));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, x1, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, x2, (size_t)N * 16, hipMemcpyHostToDevice, st));
  if (inliers) VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, inliers, (size_t)N, hipMemcpyHostToDevice, st));
  double* dF = (double*)s[3].p;
  VO_HIP_TRY(ctx, hipMemcpyAsync(dF, F, 72, hipMemcpyHostToDevice, st));
  vo_cam2 cams;
  memcpy(cams.K1, K1, 72);
  memcpy(cams.K2, K2, 72);
  hipLaunchKernelGGL(relative_pose_kernel, dim3(1), dim3(256), 0, st, (const double*)s[0].p, (const double*)s[1].p, N,
                     inliers ? (const uint8_t*)s[2].p : nullptr, (const double*)dF, cams, dF + 16, (double*)s[4].p,
                     (uint8_t*)s[5].p, dF + 32);
  VO_TRY(vo_check_launch(ctx, "relative_pose_kernel"));
  VO_HIP_TRY(ctx, hipMemcpyAsync(M, dF + 16, 96, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(X, s[4].p, (size_t)N * 24, hipMemcpyDeviceToHost, st));
  if (mask_out) VO_HIP_TRY(ctx, hipMemcpyAsync(mask_out, s[5].p, (size_t)N, hipMemcpyDeviceToHost, st));
  if (M4) VO_HIP_TRY(ctx, hipMemcpyAsync(M4, dF + 32, 48 * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

}  // extern "C"
