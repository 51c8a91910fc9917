// Pose refinement on the device.
//
// Reference call site: src/vo/pose_estimation/p3p.py:188-213 (_nonlinear_refinement): SciPy
// least_squares (TRF, numerical Jacobian, tolerances 1e-8) over the twist of the pose, one
// residual per inlier = its reprojection distance, started from the best RANSAC hypothesis.
// The objective  sum_i |x_i - proj(K, R X_i + t)|^2  does not depend on the parametrisation, so
// this kernel runs Gauss-Newton with the analytic Jacobian on the left-multiplied increment
// T <- [Exp(w) | v] T  (restated in oracle/refine_np.py) and converges to the minimiser itself
// in 3-4 iterations; SciPy stops within ~1e-4 of it (tests/test_oracle_refine.py).
//
// One workgroup of 512 threads: every thread keeps its (up to four) points in registers and
// accumulates their 21 + 6 + 1 terms once per iteration, the partial sums
// are added by wave shuffles and then through LDS, lane 0 solves the 6x6 system (Cholesky) and
// applies the update; all fp64.
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int RF_T = 512;
constexpr int RF_S = 28;   // 21 (upper triangle of J^T J) + 6 (J^T e) + 1 (cost)
constexpr int RF_PT = 4;   // points per thread kept in registers (N <= 2048; beyond that they are re-read)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

struct rf_point {
  double X, Y, Z, u, v;
  bool on;
};

__device__ __forceinline__ rf_point load_point(const double* __restrict__ X, const double* __restrict__ x, int N,
                                               const uint8_t* __restrict__ mask8,
                                               const unsigned long long* __restrict__ mask_bits, int i) {
  rf_point p;
  const bool in = i < N;
  const int j = in ? i : 0;
  p.X = X[3 * j];
  p.Y = X[3 * j + 1];
  p.Z = X[3 * j + 2];
  p.u = x[2 * j];
  p.v = x[2 * j + 1];
  bool on = in;
  if (mask8) on = on && mask8[j] != 0;
  if (mask_bits) on = on && ((mask_bits[j >> 6] >> (j & 63)) & 1ull) != 0;
  p.on = on;
  return p;
}

// adds one point's terms for the pose (R, t) to s[]
__device__ __forceinline__ void add_point(const rf_point& p, const double* R, const double* t, double fx, double fy,
                                          double cx, double cy, double* s) {
  if (!p.on) return;
  const double px = R[0] * p.X + R[1] * p.Y + R[2] * p.Z + t[0];
  const double py = R[3] * p.X + R[4] * p.Y + R[5] * p.Z + t[1];
  const double pz = R[6] * p.X + R[7] * p.Y + R[8] * p.Z + t[2];
  const double iz = 1.0 / pz;
  const double eu = p.u - (fx * px * iz + cx);
  const double ev = p.v - (fy * py * iz + cy);
  // rows of J = d proj / d (v, w):  d proj / d p = [[a, 0, c], [0, b, d]] with a = fx iz, c = -fx px iz^2,
  // b = fy iz, d = -fy py iz^2;  d p / d v = I, d p / d w = -[p]_x.  The two structural zeros are
  // written out (J0 = [a, 0, c, c py, a pz - c px, -a py], J1 = [0, b, d, d py - b pz, -d px, b px]):
  // their products add exact zeros, so dropping them changes no bit and saves a quarter of the flops.
  const double a = fx * iz, c = -fx * px * iz * iz;
  const double b = fy * iz, d = -fy * py * iz * iz;
  const double j03 = c * py, j04 = a * pz - c * px, j05 = -a * py;
  const double j13 = -b * pz + d * py, j14 = -d * px, j15 = b * px;
  s[0] += a * a;                       // (0,0)
  s[2] += a * c;                       // (0,2)   [(0,1) stays 0]
  s[3] += a * j03;
  s[4] += a * j04;
  s[5] += a * j05;
  s[6] += b * b;                       // (1,1)
  s[7] += b * d;
  s[8] += b * j13;
  s[9] += b * j14;
  s[10] += b * j15;
  s[11] += c * c + d * d;              // (2,2)
  s[12] += c * j03 + d * j13;
  s[13] += c * j04 + d * j14;
  s[14] += c * j05 + d * j15;
  s[15] += j03 * j03 + j13 * j13;      // (3,3)
  s[16] += j03 * j04 + j13 * j14;
  s[17] += j03 * j05 + j13 * j15;
  s[18] += j04 * j04 + j14 * j14;      // (4,4)
  s[19] += j04 * j05 + j14 * j15;
  s[20] += j05 * j05 + j15 * j15;      // (5,5)
  s[21] += a * eu;
  s[22] += b * ev;
  s[23] += c * eu + d * ev;
  s[24] += j03 * eu + j13 * ev;
  s[25] += j04 * eu + j14 * ev;
  s[26] += j05 * eu + j15 * ev;
  s[27] += eu * eu + ev * ev;
}

// sums for the pose (R, t): the thread's cached points, plus (N > RF_PT * RF_T) the rest re-read
__device__ __forceinline__ void accumulate(const rf_point* cache, const double* __restrict__ X,
                                           const double* __restrict__ x, int N, const uint8_t* __restrict__ mask8,
                                           const unsigned long long* __restrict__ mask_bits, const double* R,
                                           const double* t, double fx, double fy, double cx, double cy, double* s) {
#pragma unroll
  for (int k = 0; k < RF_S; ++k) s[k] = 0.0;
#pragma unroll
  for (int k = 0; k < RF_PT; ++k) add_point(cache[k], R, t, fx, fy, cx, cy, s);
  for (int i = RF_PT * RF_T + threadIdx.x; i < N; i += RF_T)
    add_point(load_point(X, x, N, mask8, mask_bits, i), R, t, fx, fy, cx, cy, s);
}

// block-wide sums of s[] -> s_tot[] (valid in every thread after the call)
__device__ __forceinline__ void block_sums(double* s, double (*s_w)[RF_S], double* s_tot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RF_S; ++k) {
    const double v = wave_sum_f64(s[k]);
    if (lane == 0) s_w[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < RF_S) {
    double v = 0.0;
    for (int w = 0; w < RF_T / 64; ++w) v += s_w[w][threadIdx.x];
    s_tot[threadIdx.x] = v;
  }
  __syncthreads();
}

__device__ bool cholesky_solve6(const double* s, double* d) {
  // A (upper triangle in s[0..20], row-major) d = b (s[21..26])
  double A[6][6], L[6][6];
  int q = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      A[a][b] = s[q];
      A[b][a] = s[q];
      ++q;
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = A[i][j];
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(v > 0.0)) return false;
        L[i][i] = sqrt(v);
      } else {
        L[i][j] = v / L[j][j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double v = s[21 + i];
    for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
    y[i] = v / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < 6; ++k) v -= L[k][i] * d[k];
    d[i] = v / L[i][i];
  }
  return true;
}

__device__ void exp_so3(const double* w, double* E) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double a = 1.0, b = 0.0;
  if (th >= 1e-12) {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / (th * th);
  }
  const double Wx[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double w2 = 0.0;
      for (int k = 0; k < 3; ++k) w2 += Wx[3 * r + k] * Wx[3 * k + c];
      E[3 * r + c] = (r == c ? 1.0 : 0.0) + a * Wx[3 * r + c] + b * w2;
    }
}

// Rt0: R (9, row-major) then t (3).  out: R (9), t (3), iterations, cost (14 doubles).
__global__ __launch_bounds__(RF_T) void refine_pose_kernel(const double* __restrict__ X, const double* __restrict__ x, int N,
                                                           const int* __restrict__ d_n,
                                                           const uint8_t* __restrict__ mask8,
                                                           const unsigned long long* __restrict__ mask_bits,
                                                           const double* __restrict__ Rt0, double fx, double fy, double cx,
                                                           double cy, int max_iter, double tol, double* __restrict__ out,
                                                           unsigned tag) {
  __shared__ double s_w[RF_T / 64][RF_S];
  __shared__ double s_tot[RF_S], s_new[RF_S];
  __shared__ double s_pose[12], s_try[12];
  __shared__ int s_state;   // 0 continue, 1 stop (keep s_pose)
  if (d_n) N = min(N, *d_n);
  if (threadIdx.x < 12) s_pose[threadIdx.x] = Rt0[threadIdx.x];
  __syncthreads();
  // the thread's points stay in registers for every iteration (all loads go out together, once)
  rf_point cache[RF_PT];
#pragma unroll
  for (int k = 0; k < RF_PT; ++k) cache[k] = load_point(X, x, N, mask8, mask_bits, k * RF_T + threadIdx.x);
  double s[RF_S];
  int it = 0;
  accumulate(cache, X, x, N, mask8, mask_bits, s_pose, s_pose + 9, fx, fy, cx, cy, s);
  block_sums(s, s_w, s_tot);
  double cost = s_tot[27];
  while (it < max_iter) {
    if (threadIdx.x == 0) {
      double d[6];
      s_state = 0;
      if (!cholesky_solve6(s_tot, d)) {
        s_state = 1;
      } else {
        double E[9];
        exp_so3(d + 3, E);
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            s_try[3 * r + c] = E[3 * r] * s_pose[c] + E[3 * r + 1] * s_pose[3 + c] + E[3 * r + 2] * s_pose[6 + c];
          s_try[9 + r] = E[3 * r] * s_pose[9] + E[3 * r + 1] * s_pose[10] + E[3 * r + 2] * s_pose[11] + d[r];
        }
        const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
        const double tn = sqrt(s_try[9] * s_try[9] + s_try[10] * s_try[10] + s_try[11] * s_try[11]);
        if (dn <= tol * (1.0 + tn)) s_state = 2;   // converged after this update
      }
    }
    __syncthreads();
    if (s_state == 1) break;
    accumulate(cache, X, x, N, mask8, mask_bits, s_try, s_try + 9, fx, fy, cx, cy, s);
    block_sums(s, s_w, s_new);
    const double cost_new = s_new[27];
    if (!(cost_new <= cost)) break;   // no decrease: keep the previous pose (uniform: all threads read the same sums)
    __syncthreads();
    if (threadIdx.x < 12) s_pose[threadIdx.x] = s_try[threadIdx.x];
    if (threadIdx.x < RF_S) s_tot[threadIdx.x] = s_new[threadIdx.x];
    ++it;
    const bool done = s_state == 2 || cost - cost_new <= 1e-16 * cost;
    cost = cost_new;
    __syncthreads();
    if (done) break;
  }
  if (threadIdx.x < 12) out[threadIdx.x] = s_pose[threadIdx.x];
  if (threadIdx.x == 0) {
    out[12] = (double)it;
    out[13] = cost;
  }
  if (tag) {   // out is mapped host memory the host polls: the tag goes last
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) out[14] = (double)tag;
  }
}

}  // namespace

// count read on the device when d_n != nullptr (pipeline); at most one of the masks may be given;
// tag != 0: d_out14 has a 15th slot that receives the tag once the other 14 are visible to the host
int vo_refine_pose_ndev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const int32_t* d_n, const double* K,
                        const uint8_t* d_mask8, const uint64_t* d_mask_bits, const double* d_Rt0, int max_iter,
                        double* d_out14, unsigned tag) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_X && d_x && K && d_Rt0 && d_out14, "refine_pose: null pointer");
  VO_REQUIRE(ctx, N >= 0 && max_iter >= 0 && max_iter <= 100, "refine_pose: bad arguments");
  VO_REQUIRE(ctx, !(d_mask8 && d_mask_bits), "refine_pose: give one kind of mask");
  VO_REQUIRE(ctx, K[0] != 0.0 && K[4] != 0.0, "refine_pose: singular intrinsics");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    vo_prof_scope ps(ctx, VO_K_REFINE);
    hipLaunchKernelGGL(refine_pose_kernel, dim3(1), dim3(RF_T), 0, ctx->stream, d_X, d_x, N, d_n, d_mask8,
                       (const unsigned long long*)d_mask_bits, d_Rt0, K[0], K[4], K[2], K[5], max_iter, 1e-9, d_out14, tag);
  }
  return vo_check_launch(ctx, "refine_pose_kernel");
}

extern "C" {

int vo_refine_pose_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const double* K, const uint8_t* d_mask,
                       const double* d_Rt0, int max_iter, double* d_out14) {
  return vo_refine_pose_ndev(ctx, d_X, d_x, N, nullptr, K, d_mask, nullptr, d_Rt0, max_iter, d_out14, 0u);
}

int vo_refine_pose(vo_ctx* ctx, const double* X, const double* x, int N, const double* K, const uint8_t* inlier_mask,
                   const double* R0, const double* t0, int max_iter, double* R, double* t, int32_t* iterations,
                   double* cost) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, X && x && K && R0 && t0 && R && t, "refine_pose: null pointer");
  VO_REQUIRE(ctx, N >= 0, "refine_pose: bad N");
  vo_buf* s = ctx->scratch;
  const size_t n = (size_t)(N > 0 ? N : 1);
  VO_TRY(vo_ensure(ctx, s[0], n * 24));
  VO_TRY(vo_ensure(ctx, s[1], n * 16));
  VO_TRY(vo_ensure(ctx, s[2], n));
  VO_TRY(vo_ensure(ctx, s[3], 14 * 8 + 12 * 8));
  hipStream_t st = ctx->stream;
  double h[12];
  memcpy(h, R0, 72);
  memcpy(h + 9, t0, 24);
  double* d_Rt0 = (double*)s[3].p;
  double* d_out = d_Rt0 + 12;
  if (N > 0) {
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, X, (size_t)N * 24, hipMemcpyHostToDevice, st));
    VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, x, (size_t)N * 16, hipMemcpyHostToDevice, st));
    if (inlier_mask) VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, inlier_mask, (size_t)N, hipMemcpyHostToDevice, st));
  }
  VO_HIP_TRY(ctx, hipMemcpyAsync(d_Rt0, h, 96, hipMemcpyHostToDevice, st));
  VO_TRY(vo_refine_pose_dev(ctx, (const double*)s[0].p, (const double*)s[1].p, N, K,
                            inlier_mask ? (const uint8_t*)s[2].p : nullptr, d_Rt0, max_iter, d_out));
  double o[14];
  VO_HIP_TRY(ctx, hipMemcpyAsync(o, d_out, 14 * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  memcpy(R, o, 72);
  memcpy(t, o + 9, 24);
  if (iterations) *iterations = (int32_t)o[12];
  if (cost) *cost = o[13];
  return VO_OK;
}

}  // extern "C"
