// Pose refinement on the device.
//
// Reference call site: src/vo/pose_estimation/p3p.py:188-213 (_nonlinear_refinement): SciPy
// least_squares (TRF, numerical Jacobian, tolerances 1e-8) over the twist of the pose, one
// residual per inlier = its reprojection distance, started from the best RANSAC hypothesis.
// The objective  sum_i |x_i - proj(K, R X_i + t)|^2  does not depend on the parametrisation, so
// this kernel runs Gauss-Newton with the analytic Jacobian on the left-multiplied increment
// T <- [Exp(w) | v] T  (restated in oracle/refine_np.py) until the next step would be below 1e-9
// relative, 2-3 iterations; SciPy stops within ~1e-4 of the minimiser (tests/test_oracle_refine.py).
//
// One workgroup of 512 threads: every thread keeps its (up to four) points in registers and
// accumulates their 21 + 6 + 1 terms once per iteration; a wave adds its lanes' terms with a
// halving butterfly (32 shuffles for all 28 sums), wave 0 adds the eight wave totals, solves the
// 6x6 system with one lane per row and applies the update; all fp64, two barriers per iteration.
#include <type_traits>
#include "state_device.h"
#include "dlt_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int RF_T = 512;
constexpr int RF_S = 28;   // 21 (upper triangle of J^T J) + 6 (J^T e) + 1 (cost)
constexpr int RF_PT = 4;   // points per thread kept in registers (N <= 2048; beyond that they are re-read)

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

// the value of lane (l ^ OFF): a DPP move where one exists (8: row_ror:8, 2 and 1: quad_perm), the LDS crossbar otherwise
template <int OFF>
__device__ __forceinline__ double lane_xor(double x) {
  if constexpr (OFF == 8 || OFF == 2 || OFF == 1) {
    constexpr int ctrl = OFF == 8 ? 0x128 : (OFF == 2 ? 0x4E : 0xB1);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {
    return __shfl_xor(x, OFF);
  }
}

// One butterfly step of the wave-wide sums: the lane pair (l, l ^ OFF) splits the CNT values it still
// carries between its two lanes, so the number of values halves while the number of lanes summed doubles.
// (OFF = 32 and 16: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper lanes (odd rows) of one register
//  with the lower lanes (even rows) of another -- exactly this step's "keep one half, send the other", so the two values
//  are swapped in place and added, no select and no trip through the LDS crossbar; a + b = b + a, same bits.)
template <int CNT, int OFF>
__device__ __forceinline__ void fold_step(double* v, int lane) {
  if constexpr (OFF == 32 || OFF == 16) {
#pragma unroll
    for (int k = 0; k < CNT / 2; ++k) {
      const unsigned alo = (unsigned)__double2loint(v[k]), ahi = (unsigned)__double2hiint(v[k]);
      const unsigned blo = (unsigned)__double2loint(v[k + CNT / 2]), bhi = (unsigned)__double2hiint(v[k + CNT / 2]);
      const auto rl = OFF == 32 ? __builtin_amdgcn_permlane32_swap(alo, blo, false, false)
                                : __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto rh = OFF == 32 ? __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false)
                                : __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      v[k] = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    return;
  }
  const bool hi = (lane & OFF) != 0;
#pragma unroll
  for (int k = 0; k < CNT / 2; ++k) {
    const double keep = hi ? v[k + CNT / 2] : v[k];
    const double send = hi ? v[k] : v[k + CNT / 2];
    v[k] = keep + lane_xor<OFF>(send);
  }
}

// wave-wide sums of s[0..27]: afterwards lanes 2j and 2j+1 hold the total of s[j] (32 shuffles, not 28 * 6)
__device__ __forceinline__ double wave_sums(const double* s, int lane) {
  double v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = k < RF_S ? s[k] : 0.0;
  fold_step<32, 32>(v, lane);
  fold_step<16, 16>(v, lane);
  fold_step<8, 8>(v, lane);
  fold_step<4, 4>(v, lane);
  fold_step<2, 2>(v, lane);
  return v[0] + lane_xor<1>(v[0]);
}

// sum of an int over the 64 lanes, in all of them: four DPP adds inside each row of 16, the four row sums through scalar
// registers (six dependent trips through the LDS crossbar as a __shfl_xor butterfly)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// lane K's value in every lane, through a scalar register (K is a compile-time constant: v_readlane_b32, no trip
// through the LDS crossbar as __shfl makes)
template <int K>
__device__ __forceinline__ double lane_value(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), K), hi = __builtin_amdgcn_readlane(__double2hiint(v), K);
  return __hiloint2double(hi, lo);
}

// position of (a, b), a <= b, in the row-major upper triangle s[0..20]
__device__ __forceinline__ int tri_index(int a, int b) { return a * 6 - (a * (a - 1)) / 2 + (b - a); }

// Solves the 6x6 normal equations held one total per lane (lane k of the wave: s[k]) by Gauss-Jordan
// elimination, lane i < 6 working on row i (no pivoting: the matrix is positive definite or the solve
// is refused, the same condition under which a Cholesky factorisation exists).  Returns d[i] in lane i.
__device__ __forceinline__ bool solve6_rows(double tot, int lane, double* d_out) {
  const int i = min(lane, 5);
  double row[7];
#pragma unroll
  for (int j = 0; j < 6; ++j) row[j] = __shfl(tot, tri_index(min(i, j), max(i, j)));
  row[6] = __shfl(tot, 21 + i);
  bool ok = true;
  auto pivot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double pk[7];
#pragma unroll
    for (int j = k; j < 7; ++j) pk[j] = lane_value<k>(row[j]);
    ok = ok && pk[k] > 0.0;
    const double f = row[k] / pk[k];
    if (i != k) {
#pragma unroll
      for (int j = k + 1; j < 7; ++j) row[j] -= f * pk[j];
      row[k] = 0.0;
    }
  };
  pivot(std::integral_constant<int, 0>());
  pivot(std::integral_constant<int, 1>());
  pivot(std::integral_constant<int, 2>());
  pivot(std::integral_constant<int, 3>());
  pivot(std::integral_constant<int, 4>());
  pivot(std::integral_constant<int, 5>());
  double diag = row[0];
#pragma unroll
  for (int j = 1; j < 6; ++j) diag = i == j ? row[j] : diag;
  *d_out = row[6] / diag;
  return ok;
}

// Rodrigues coefficients sin(th)/th and (1 - cos th)/th^2: their Taylor series in th^2 for the small
// rotations an update step makes (nine terms, below 1e-17 for th < 1/4), the library functions otherwise
__device__ __forceinline__ void rodrigues_coefficients(double th2, double* a, double* b) {
  if (th2 < 0.0625) {
    double sa = 1.0, sb = 1.0;
    const double ca[8] = {1.0 / 272, 1.0 / 210, 1.0 / 156, 1.0 / 110, 1.0 / 72, 1.0 / 42, 1.0 / 20, 1.0 / 6};
    const double cb[8] = {1.0 / 306, 1.0 / 240, 1.0 / 182, 1.0 / 132, 1.0 / 90, 1.0 / 56, 1.0 / 30, 1.0 / 12};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sa = 1.0 - th2 * ca[k] * sa;
      sb = 1.0 - th2 * cb[k] * sb;
    }
    *a = sa;
    *b = 0.5 * sb;
  } else {
    const double th = sqrt(th2);
    *a = sin(th) / th;
    *b = (1.0 - cos(th)) / th2;
  }
}

// Gauss-Newton on the pose in s_pose (s_try = the trial pose, both 12 doubles of LDS, initialised by the caller,
// s_state = 0), all RF_T threads.  Two barriers per iteration: every wave adds up its points for the trial pose
// and leaves 28 wave totals in LDS; wave 0 alone then adds those, accepts or rejects the trial, solves for the
// next step and writes the next trial pose while the other waves wait.  On return s_pose holds the result;
// *it_out / *cost_out are valid in wave 0.
__device__ __forceinline__ void gauss_newton(const rf_point* cache, const double* __restrict__ X,
                                             const double* __restrict__ x, int N, const uint8_t* __restrict__ mask8,
                                             const unsigned long long* __restrict__ mask_bits, double fx, double fy,
                                             double cx, double cy, int max_iter, double tol, double (*s_w)[RF_S],
                                             double* s_pose, double* s_try, int* s_state, int* it_out,
                                             double* cost_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int it = -1;              // -1: the first pass evaluates the starting pose
  double cost = 0.0;
  for (;;) {
    __syncthreads();
    if (*s_state != 0) break;
    double s[RF_S];
    accumulate(cache, X, x, N, mask8, mask_bits, s_try, s_try + 9, fx, fy, cx, cy, s);
    const double wsum = wave_sums(s, lane);
    if ((lane & 1) == 0 && (lane >> 1) < RF_S) s_w[wv][lane >> 1] = wsum;
    __syncthreads();
    if (wv != 0) continue;
    double tot = 0.0;
    if (lane < RF_S) {
#pragma unroll
      for (int w = 0; w < RF_T / 64; ++w) tot += s_w[w][lane];
    }
    const double cost_new = lane_value<27>(tot);
    bool stop = false;
    if (it < 0) {
      it = 0;
      cost = cost_new;
    } else if (!(cost_new <= cost)) {
      stop = true;           // no decrease: keep the previous pose
    } else {
      if (lane < 12) s_pose[lane] = s_try[lane];
      ++it;
      stop = cost - cost_new <= 1e-16 * cost;
      cost = cost_new;
    }
    if (!stop && it >= max_iter) stop = true;
    if (!stop) {
      double di;
      const bool ok = solve6_rows(tot, lane, &di);
      if (!ok) {
        stop = true;
      } else {
        const double d[6] = {lane_value<0>(di), lane_value<1>(di), lane_value<2>(di), lane_value<3>(di), lane_value<4>(di),
                             lane_value<5>(di)};
        double ca, cb;
        rodrigues_coefficients(d[3] * d[3] + d[4] * d[4] + d[5] * d[5], &ca, &cb);
        const double Wx[9] = {0.0, -d[5], d[4], d[5], 0.0, -d[3], -d[4], d[3], 0.0};
        double E[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            double w2 = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) w2 += Wx[3 * r + k] * Wx[3 * k + c];
            E[3 * r + c] = (r == c ? 1.0 : 0.0) + ca * Wx[3 * r + c] + cb * w2;
          }
        // every lane computes the same 12 numbers; lane 0 stores them (s_pose was written above by this wave)
        double P[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) P[q] = s_pose[q];
        double T[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int c = 0; c < 3; ++c) T[3 * r + c] = E[3 * r] * P[c] + E[3 * r + 1] * P[3 + c] + E[3 * r + 2] * P[6 + c];
          T[9 + r] = E[3 * r] * P[9] + E[3 * r + 1] * P[10] + E[3 * r + 2] * P[11] + d[r];
        }
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 12; ++q) s_try[q] = T[q];
        }
        const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
        const double tn = sqrt(T[9] * T[9] + T[10] * T[10] + T[11] * T[11]);
        if (dn <= tol * (1.0 + tn)) stop = true;   // converged: a step this small is not taken (nor evaluated)
      }
    }
    if (stop && lane == 0) *s_state = 1;
  }
  *it_out = it;
  *cost_out = cost;
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
  __shared__ double s_pose[12], s_try[12];
  __shared__ int s_state;   // 0 run the next trial, 1 finished (s_pose holds the result)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (d_n) N = min(N, *d_n);
  if (threadIdx.x < 12) {
    const double v = Rt0[threadIdx.x];
    s_pose[threadIdx.x] = v;
    s_try[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) s_state = 0;
  // the thread's points stay in registers for every iteration (all loads go out together, once)
  rf_point cache[RF_PT];
#pragma unroll
  for (int k = 0; k < RF_PT; ++k) cache[k] = load_point(X, x, N, mask8, mask_bits, k * RF_T + threadIdx.x);
  int it;
  double cost;
  gauss_newton(cache, X, x, N, mask8, mask_bits, fx, fy, cx, cy, max_iter, tol, s_w, s_pose, s_try, &s_state, &it, &cost);
  if (wv == 0) {
    if (lane < 12) out[lane] = s_pose[lane];
    if (lane == 0) {
      out[12] = (double)(it < 0 ? 0 : it);
      out[13] = cost;
    }
    if (tag) {   // out is mapped host memory the host polls: the tag goes last
      __threadfence_system();
      if (lane == 0) {
        __hip_atomic_store(&out[14], (double)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// The frame loop's pose kernel (vo_pose_job): one workgroup.
//   1. wave 0 replays the sequential RANSAC rule over the scored hypotheses (ransac.py:90-121, state_device.h)
//      while the other waves already fetch the population's coordinates;
//   2. all waves refine the accepted pose over its inliers (p3p.py:188-213), as refine_pose_kernel does;
//   3. all threads walk the new frame's features: pose, outliers, bearing-angle candidates (main.py:261-268);
//   4. (job.tail) main.py:279-286 as state_landmarks_kernel does it -- the candidates, compacted into a list, are
//      triangulated one per work item with their track's own start pose (triangulation.py:38-86), inserted as landmarks
//      (state.py:69-88), all landmarks checked for cheirality when there was a candidate (state.py:90-107) -- and the
//      step's result record (VO_FUSED_TAIL=1; round 2's form: it saves launches, but one workgroup then triangulates every
//      candidate -- the frame loop has used the many-workgroup launch since round 3, see pipeline.hip: enqueue_pose_half).
constexpr int TAIL_LIST = 4096;   // candidates the LDS list holds (more: every work item walks its own features)
__global__ __launch_bounds__(RF_T) void frame_pose_kernel(vo_pose_job job) {
  using namespace vo_state_dev;
  __shared__ double s_table[RP_TABLE_LDS];
  __shared__ double s_w[RF_T / 64][RF_S];
  __shared__ double s_pose[12], s_try[12];
  __shared__ int s_state, s_cand[RF_T / 64];
  __shared__ int s_list[TAIL_LIST];
  __shared__ int s_nc, s_tail[2];
  constexpr int MASK_LDS = 256;        // mask words the replay hands over in LDS (16384 correspondences)
  __shared__ unsigned long long s_mask[MASK_LDS];
  __shared__ int s_mask_ok;
  if (blockIdx.x != 0) {               // several sequences per launch: one workgroup per sequence
    const size_t q = blockIdx.x;
    job.ctl += q;
    job.rp.valid += q * (size_t)job.rp.hyp;
    job.rp.counts += q * (size_t)job.rp.hyp;
    job.rp.R += q * (size_t)job.rp.hyp * 9;
    job.rp.t += q * (size_t)job.rp.hyp * 3;
    job.rp.masks += q * (size_t)job.rp.hyp * job.rp.words;
    job.rp.best_mask += q * (size_t)job.rp.words;
    job.B = vo_feat_seq(job.B, q);
    if (job.res) {
      job.res += q;
      job.seq_word += q;
    }
  }
  vo_seq_ctl* ctl = job.ctl;
  if (threadIdx.x == 0) ctl->ts[3] = wall_clock64();
  replay_head head;                    // (requested with the fault word below: one round trip for both)
  if (job.do_replay && (threadIdx.x >> 6) == 0) head = replay_prefetch(ctl);
  {
    // a fault of an earlier step (sticky) or of this step's regroup, which leaves "few landmarks" in its own word
    const int entry_fault = ctl->fault | (job.do_replay ? ctl->few : 0);
    if (entry_fault) {
      if (threadIdx.x == 0) {
        ctl->fault = entry_fault;
        if (job.tail && job.res) {
          ctl->ts[4] = wall_clock64();
          write_fault_record(ctl, entry_fault, job.res, job.seq_word, job.seq);
        }
      }
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double* X = job.B.land;
  const double* x = job.B.kp64;
  const int cap = job.B.pitch;
  const bool lds_table = job.do_replay && job.rp.table_len + 1 <= RP_TABLE_LDS;
  replay_chunk first;                              // (wave 0: the replay's first reads go out before everything else)
  if (job.do_replay && wv == 0) replay_fetch(job.rp, lane, 0, first);
  if (lds_table)
    for (int k = tid; k < job.rp.table_len + 1; k += RF_T) s_table[k] = job.rp.table[k];
  // coordinates first (the population's size and the inlier mask are known only after the replay)
  rf_point cache[RF_PT];
#pragma unroll
  for (int k = 0; k < RF_PT; ++k) cache[k] = load_point(X, x, cap, nullptr, nullptr, k * RF_T + tid);
  if (tid == 0) {
    s_state = 0;
    s_mask_ok = 0;
  }
  __syncthreads();
  if (job.do_replay && wv == 0)
    replay_wave(ctl, job.rp, lane, lds_table ? s_table : job.rp.table, first, true, job.debug_fault_every, &head, s_pose, s_mask,
                MASK_LDS, &s_mask_ok);
  __syncthreads();
  if (tid == 0 && job.stamps) ctl->ts[5] = wall_clock64();
  if (ctl->fault) {                             // (raised by the replay: the host finishes this step)
    if (job.tail && job.res && tid == 0) {
      ctl->ts[4] = wall_clock64();
      write_fault_record(ctl, ctl->fault, job.res, job.seq_word, job.seq);
    }
    return;
  }
  const int N = min(ctl->n_p3p, cap);
  const unsigned long long* mask_bits = job.rp.best_mask;
  // the replay left the accepted pose and its mask row in LDS (s_pose, s_mask) -- the host path (do_replay = 0) in memory
  const bool from_lds = job.do_replay && s_mask_ok != 0;
#pragma unroll
  for (int k = 0; k < RF_PT; ++k) {
    const int i = k * RF_T + tid;
    const int w = min(i, cap - 1) >> 6;
    const unsigned long long word = from_lds ? s_mask[min(w, MASK_LDS - 1)] : mask_bits[w];
    cache[k].on = i < N && ((word >> (i & 63)) & 1ull) != 0;
  }
  if (tid < 12) {
    const double v = job.do_replay ? s_pose[tid] : ctl->best_pose[tid];
    s_pose[tid] = v;
    s_try[tid] = v;
  }
  int it;
  double cost;
  gauss_newton(cache, X, x, N, nullptr, mask_bits, job.cam.K[0], job.cam.K[4], job.cam.K[2], job.cam.K[5], job.max_iter,
               1e-9, s_w, s_pose, s_try, &s_state, &it, &cost);
  if (wv == 0) {
    if (lane < 12) ctl->refined[lane] = s_pose[lane];
    if (lane == 0) {
      ctl->refined[12] = (double)(it < 0 ? 0 : it);
      ctl->refined[13] = cost;
      if (job.stamps) ctl->ts[7] = wall_clock64();
    }
  }
  if (!job.walk) {          // the walk's own launch goes on from here: its counters start at 0
    if (tid == 0) {
      ctl->n_cand = 0;
      ctl->n_pend = 0;
    }
    return;
  }
  // ---- the new pose, both ways (update_with_world_pose, state.py:38-50), then every feature ----
  double Tcw[12], Twc[12];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Tcw[4 * r] = s_pose[3 * r];
    Tcw[4 * r + 1] = s_pose[3 * r + 1];
    Tcw[4 * r + 2] = s_pose[3 * r + 2];
    Tcw[4 * r + 3] = s_pose[9 + r];
  }
  rigid_inverse_3x4(Tcw, Twc);
  const int n2 = ctl->n2, n_tri = ctl->n_tri;
  if (tid == 0) {
    s_nc = 0;
    s_tail[0] = s_tail[1] = 0;
  }
  __syncthreads();
  int count = 0;
  for (int i = tid; i < n2; i += RF_T) {
    const int c = candidate_feature(job.B, i, n_tri, mask_bits, job.cam, Twc, job.bearing_thr);
    count += c;
    if (c && job.tail) {
      const int slot = atomicAdd(&s_nc, 1);
      if (slot < TAIL_LIST) s_list[slot] = i;
    }
  }
  count = wave_sum_i32(count);
  if (lane == 0) s_cand[wv] = count;
  __syncthreads();
  int n_cand = 0;
  for (int w = 0; w < RF_T / 64; ++w) n_cand += s_cand[w];
  if (tid == 0) {
    ctl->n_cand = n_cand;
    ctl->n = n2;                                 // the new frame is the current one from here on
  }
  if (tid < 12) {
    double ncw = 0.0, nwc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      ncw = tid == k ? Tcw[k] : ncw;
      nwc = tid == k ? Twc[k] : nwc;
    }
    commit_pose(ctl, tid, ncw, nwc);
  }
  if (!job.tail) return;
  // ---- 4. triangulate_candidates -> update_with_world_landmarks -> _check_landmarks, then the record ----
  __syncthreads();                               // (poses committed, candidate flags and restarted tracks in place)
  unsigned long long ts4 = 0ull;
  if (tid == 0) {
    ts4 = wall_clock64();
    ctl->ts[4] = ts4;
  }
  vo_feat B = job.B;
  {
    double Tp[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tp[k] = ctl->T_cw_prev[k];
    int dropped = 0, land = 0;
    // _check_landmarks for one feature whose landmark is X (state.py:90-107): behind either camera -> the track restarts here
    auto check = [&](int i, int st, double X0, double X1, double X2) {
      if (n_cand > 0) {
        const double zc = Tcw[8] * X0 + Tcw[9] * X1 + Tcw[10] * X2 + Tcw[11];
        const double zp = Tp[8] * X0 + Tp[9] * X1 + Tp[10] * X2 + Tp[11];
        if (zc < 0.0 || zp < 0.0) {                            // (NaN landmarks compare false)
          const double nan = dnan();
          B.land[3 * i] = B.land[3 * i + 1] = B.land[3 * i + 2] = nan;
          B.state[i] = 0;
          st = 0;
          B.track[2 * i] = B.kp64[2 * i];
          B.track[2 * i + 1] = B.kp64[2 * i + 1];
#pragma unroll
          for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + i] = Twc[k];
          ++dropped;
        }
      }
      land += st == 2 ? 1 : 0;
    };
    // features that are no candidates: nothing to wait for -- four per work item and pass, their loads in flight together
    for (int base = tid; base < n2; base += 4 * RF_T) {
      int st[4];
      double X[4][3];
      bool go[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * RF_T, ic = min(i, n2 - 1);
        go[u] = i < n2 && !B.cand[ic];
        st[u] = B.state[ic];
        X[u][0] = B.land[3 * ic];
        X[u][1] = B.land[3 * ic + 1];
        X[u][2] = B.land[3 * ic + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (go[u]) check(base + u * RF_T, st[u], X[u][0], X[u][1], X[u][2]);
    }
    // the candidates: triangulated with their track's own start pose, inserted, checked by the same work item
    double C2[12];
    k_times(job.cam.K, Tcw, C2);                 // proj2 = K inv(current_pose)[:3] (triangulation.py:53-57)
    auto triangulate = [&](int i) {
      double Ts[12], Ti[12], C1[12], X[3];
#pragma unroll
      for (int q = 0; q < 12; ++q) Ts[q] = B.pose[(size_t)q * B.pitch + i];
      rigid_inverse_3x4(Ts, Ti);
      k_times(job.cam.K, Ti, C1);                // proj1 = K inv(pose_start)[:3]
      vo_dlt::triangulate_point(C1, B.track[2 * i], B.track[2 * i + 1], C2, B.kp64[2 * i], B.kp64[2 * i + 1], X);
      B.land[3 * i] = X[0];
      B.land[3 * i + 1] = X[1];
      B.land[3 * i + 2] = X[2];
      B.state[i] = 2;
      check(i, 2, X[0], X[1], X[2]);
    };
    if (n_cand <= TAIL_LIST) {
      for (int k = tid; k < n_cand; k += RF_T) triangulate(s_list[k]);
    } else {
      for (int i = tid; i < n2; i += RF_T)
        if (B.cand[i]) triangulate(i);
    }
    dropped = wave_sum_i32(dropped);
    land = wave_sum_i32(land);
    if (lane == 0) {
      if (dropped) atomicAdd(&s_tail[0], dropped);
      if (land) atomicAdd(&s_tail[1], land);
    }
  }
  __syncthreads();
  const int n_dropped = s_tail[0], n_land = s_tail[1];
  if (tid == 0) {
    ctl->n_dropped = n_dropped;
    ctl->n_land = n_land;
    ctl->step += 1;
  }
  if (!job.res) return;
  ts4 = __shfl(ts4, 0);                          // (only wave 0's first lane writes the record's scalar fields)
  write_step_record(ctl, tid, job.max_iter > 0 ? 1 : 0, n2, n_cand, n_dropped, n_land, ts4, job.res, job.seq_word, job.seq);
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

int vo_frame_pose(vo_ctx* ctx, const vo_pose_job& job, int S) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, job.ctl && job.B.land && job.B.kp64 && job.rp.best_mask, "frame_pose: null pointer");
  VO_REQUIRE(ctx, job.max_iter >= 0 && job.max_iter <= 100, "frame_pose: bad iteration limit");
  {
    vo_prof_scope ps(ctx, VO_K_REFINE);
    hipLaunchKernelGGL(frame_pose_kernel, dim3(S), dim3(RF_T), 0, ctx->stream, job);
  }
  return vo_check_launch(ctx, "frame_pose_kernel");
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
