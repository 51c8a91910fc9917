// Device-resident Features / State / RANSAC bookkeeping of the per-frame loop (gfx950).
//
// The reference keeps these in Python objects and runs O(N) NumPy code on them between the
// heavy calls of a frame (src/main.py:248-286):
//   KLTTracker.track_features   src/vo/features/klt.py:191-280   (re-detect, filter, identity matches)
//   Matches.__init__            src/vo/primitives/matches.py:26-212 (4-group regroup + propagation)
//   RANSAC.find_best_model      src/vo/algorithms/ransac.py:90-121 (sequential accept / adapt rule)
//   State.update_with_world_pose / reset_outliers / compute_candidates / _calculate_bearing_angle /
//   update_with_world_landmarks / _check_landmarks      src/vo/primitives/state.py:38-219
//   LandmarksTriangulator.triangulate_candidates        src/vo/landmarks/triangulation.py:38-86
// Here each of them is a kernel over arrays that never leave HBM, so a frame is one chain of
// launches without a host turn.  Every kernel is one workgroup per sequence (blockIdx.y = sequence
// when several sequences are batched): the arrays are a few thousand elements, the work is
// latency-bound, and a single workgroup needs no inter-workgroup ordering.
#include <cmath>

#include "dlt_device.h"
#include "vo_state.h"

#pragma clang fp contract(off)

namespace {

constexpr int RG_T = 1024;   // regroup: threads
constexpr int UP_T = 512;    // update: threads

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000ll); }

// ---------------------------------------------------------------------------------------------
// klt.py:207-230 -> update_features (klt.py:117-189)
__global__ __launch_bounds__(256) void state_append_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat F,
                                                           const double* __restrict__ det_kp, int n_det, double frac,
                                                           int cap, int debug_fault_every, int pose_mode) {
  const int n = ctl->n;
  int fault = ctl->fault;
  if (debug_fault_every > 0 && (ctl->step % debug_fault_every) == debug_fault_every - 1) fault |= VO_FAULT_FORCED;
  // `length < self._num_features * 0.8` (klt.py:208-212)
  const bool redetect = !fault && (double)n < (double)ctl->num_features * frac;
  if (redetect && n + n_det > cap) fault |= VO_FAULT_CAPACITY;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (redetect && !fault && i < n_det) {
    const int j = n + i;
    const double x = det_kp[2 * i], y = det_kp[2 * i + 1];
    F.kp[2 * j] = (float)x;
    F.kp[2 * j + 1] = (float)y;
    F.kp64[2 * j] = (double)(float)x;
    F.kp64[2 * j + 1] = (double)(float)y;
    F.state[j] = 0;
    F.cand[j] = 0;
    F.land[3 * j] = F.land[3 * j + 1] = F.land[3 * j + 2] = dnan();
    F.track[2 * j] = (double)(float)x;          // "tracks are extended with the new keypoints"
    F.track[2 * j + 1] = (double)(float)y;
#pragma unroll
    for (int k = 0; k < 12; ++k)   // np.eye(4) (klt.py:148-153), or the pose of the frame the keypoints were found on
      F.pose[12 * j + k] = pose_mode ? ctl->T_wc[k] : ((k == 0 || k == 5 || k == 10) ? 1.0 : 0.0);
  }
  if (i == 0) {
    // (n and num_features are only read by this kernel; what it decides goes to words of its own)
    ctl->n_in = fault ? 0 : (redetect ? n + n_det : n);
    ctl->redetected = (redetect && !fault) ? 1 : 0;
    ctl->fault = fault;
  }
}

// ---------------------------------------------------------------------------------------------
// Block-wide exclusive scan of three counters packed into one 64-bit word (21 bits each).
__device__ __forceinline__ unsigned long long block_scan_excl(unsigned long long v, unsigned long long* s_wave,
                                                              unsigned long long* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    unsigned long long w = lane < nw ? s_wave[lane] : 0ull;
    unsigned long long winc = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(winc, off);
      if (lane >= off) winc += o;
    }
    if (lane < nw) s_wave[lane] = winc - w;
    if (lane == nw - 1) s_wave[nw] = winc;
  }
  __syncthreads();
  *total = s_wave[nw];
  const unsigned long long r = s_wave[wv] + inc - v;
  __syncthreads();
  return r;
}

__device__ __forceinline__ void write_group(const vo_feat& A, const vo_feat& B, int g, int src, int dst, double x,
                                            double y) {
  B.kp[2 * dst] = (float)x;
  B.kp[2 * dst + 1] = (float)y;
  B.kp64[2 * dst] = x;
  B.kp64[2 * dst + 1] = y;
  B.cand[dst] = 0;
  const double nan = dnan();
  if (g == 0) {            // triangulated: the landmark travels, the track data is over (matches.py:146-201)
    B.state[dst] = 2;
    B.land[3 * dst] = A.land[3 * src];
    B.land[3 * dst + 1] = A.land[3 * src + 1];
    B.land[3 * dst + 2] = A.land[3 * src + 2];
    B.track[2 * dst] = B.track[2 * dst + 1] = nan;
#pragma unroll
    for (int k = 0; k < 12; ++k) B.pose[12 * dst + k] = nan;
  } else {
    B.state[dst] = 1;
    B.land[3 * dst] = B.land[3 * dst + 1] = B.land[3 * dst + 2] = nan;
    if (g == 1) {          // matched before: the track goes on
      B.track[2 * dst] = A.track[2 * src];
      B.track[2 * dst + 1] = A.track[2 * src + 1];
    } else {               // newly matched: the track starts at frame 1's keypoint
      B.track[2 * dst] = A.kp64[2 * src];
      B.track[2 * dst + 1] = A.kp64[2 * src + 1];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) B.pose[12 * dst + k] = A.pose[12 * src + k];
  }
}

constexpr int RG_MAX_PER = 32;   // items per thread: capacity 32768

// PAIRS == false: items are frame 1's features 0 .. n_in-1, kept when status & err < thr, matched to themselves
//                 (klt.py:244-278): the new frame holds the kept ones only.
// PAIRS == true : items are the rows of an explicit (index in frame 1, index among the new keypoints) list;
//                 new keypoints without a match follow as the unmatched group.
template <bool PAIRS>
__global__ __launch_bounds__(RG_T) void state_regroup_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat A, vo_feat B,
                                                             const float* __restrict__ next_xy,
                                                             const uint8_t* __restrict__ status,
                                                             const float* __restrict__ err, float err_thr,
                                                             const int* __restrict__ pairs, int M,
                                                             const double* __restrict__ new_kp, int n2_in, int cap) {
  __shared__ unsigned long long s_wave[RG_T / 64 + 1];
  __shared__ unsigned s_matched[PAIRS ? 1024 : 1];   // bit per new keypoint (capacity 32768)
  if (ctl->fault) return;
  const int tid = threadIdx.x;
  const int n_items = PAIRS ? M : ctl->n_in;
  const int per = (n_items + RG_T - 1) / RG_T;
  const int j0 = tid * per, j1 = min(j0 + per, n_items);
  if (PAIRS) {
    for (int k = tid; k < 1024; k += RG_T) s_matched[k] = 0u;
    __syncthreads();
  }
  unsigned long long keys = 0ull;      // 2 bits per item: 0 triangulated, 1 matched, 2 newly matched, 3 dropped
  unsigned long long cnt = 0ull;
  for (int j = j0; j < j1; ++j) {
    int key;
    if (PAIRS) {
      const int i1 = pairs[2 * j], i2 = pairs[2 * j + 1];
      const int st = A.state[i1];
      key = st == 2 ? 0 : (st == 1 ? 1 : 2);
      atomicOr(&s_matched[i2 >> 5], 1u << (i2 & 31));
    } else {
      const bool keep = status[j] != 0 && err[j] < err_thr;
      const int st = A.state[j];
      key = keep ? (st == 2 ? 0 : (st == 1 ? 1 : 2)) : 3;
    }
    keys |= (unsigned long long)key << (2 * (j - j0));
    if (key < 3) cnt += 1ull << (21 * key);
  }
  unsigned long long total;
  const unsigned long long pre = block_scan_excl(cnt, s_wave, &total);
  const int T0 = (int)(total & 0x1fffff), T1 = (int)((total >> 21) & 0x1fffff), T2 = (int)((total >> 42) & 0x1fffff);
  int pos[3] = {(int)(pre & 0x1fffff), T0 + (int)((pre >> 21) & 0x1fffff), T0 + T1 + (int)((pre >> 42) & 0x1fffff)};
  for (int j = j0; j < j1; ++j) {
    const int key = (int)((keys >> (2 * (j - j0))) & 3ull);
    if (key == 3) continue;
    int src;
    double x, y;
    if (PAIRS) {
      src = pairs[2 * j];
      const int i2 = pairs[2 * j + 1];
      x = new_kp[2 * i2];
      y = new_kp[2 * i2 + 1];
    } else {
      src = j;
      x = (double)next_xy[2 * j];
      y = (double)next_xy[2 * j + 1];
    }
    write_group(A, B, key, src, pos[key]++, x, y);
  }
  int n2 = T0 + T1 + T2;
  if (PAIRS) {
    // unmatched new keypoints, ascending (np.delete(arange, matched), matches.py:33-37): a new track starts here
    const int per2 = (n2_in + RG_T - 1) / RG_T;
    const int k0 = tid * per2, k1 = min(k0 + per2, n2_in);
    unsigned long long c = 0ull;
    for (int k = k0; k < k1; ++k) c += ((s_matched[k >> 5] >> (k & 31)) & 1u) ? 0ull : 1ull;
    unsigned long long tot2;
    int p = n2 + (int)block_scan_excl(c, s_wave, &tot2);
    const double nan = dnan();
    for (int k = k0; k < k1; ++k) {
      if ((s_matched[k >> 5] >> (k & 31)) & 1u) continue;
      const double x = new_kp[2 * k], y = new_kp[2 * k + 1];
      B.kp[2 * p] = (float)x;
      B.kp[2 * p + 1] = (float)y;
      B.kp64[2 * p] = x;
      B.kp64[2 * p + 1] = y;
      B.state[p] = 0;
      B.cand[p] = 0;
      B.land[3 * p] = B.land[3 * p + 1] = B.land[3 * p + 2] = nan;
      B.track[2 * p] = x;
      B.track[2 * p + 1] = y;
#pragma unroll
      for (int q = 0; q < 12; ++q) B.pose[12 * p + q] = nan;
      ++p;
    }
    n2 += (int)tot2;
  }
  if (tid == 0) {
    ctl->n2 = n2;
    ctl->n_tri = T0;
    ctl->n_mat = T1;
    ctl->n_new = T2;
    int fault = 0;
    if (!PAIRS && T0 < 8) fault = VO_FAULT_FEW_LANDMARKS;   // (population below what the device-side sampler handles)
    ctl->n_p3p = fault ? 0 : T0;
    if (fault) ctl->fault = fault;
  }
}

// ---------------------------------------------------------------------------------------------
// n_iterations for an outlier ratio: k_min + #{thresholds <= ratio}; table[0] = k_min, table[1..len] thresholds
// (pipeline.hip builds it from the host's libm by bisection, so the device needs neither log nor pow and
// returns exactly what ransac.py:58-67 returns on the host).
__host__ __device__ inline long long table_lookup(const double* table, int len, long long max_it, double orat) {
  int lo = 0, hi = len;                 // number of thresholds <= orat
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (table[1 + mid] <= orat) lo = mid + 1;
    else hi = mid;
  }
  long long k = lo == len ? 0x7fffffffffffffffll : (long long)table[0] + lo;
  return (max_it >= 0 && max_it < k) ? max_it : k;
}

// One wave walks the batch of hypotheses through the reference's loop (ransac.py:90-121):
//     while n < n_iterations: draw; model None -> continue; count; strictly better -> keep, adapt; n += 1
// 64 hypotheses per round: a prefix maximum finds the hypotheses that improve on everything before them,
// and between two such events n_iterations is constant, so the place where the loop ends is a ballot.
__global__ __launch_bounds__(64) void ransac_replay_kernel(vo_seq_ctl* __restrict__ ctl,
                                                           const uint8_t* __restrict__ valid,
                                                           const int* __restrict__ counts,
                                                           const double* __restrict__ Rall,
                                                           const double* __restrict__ tall,
                                                           const unsigned long long* __restrict__ masks, int words,
                                                           int hyp, const double* __restrict__ table, int table_len,
                                                           long long max_it,
                                                           unsigned long long* __restrict__ best_mask) {
  if (ctl->fault) return;
  const int lane = threadIdx.x;
  const int N = ctl->n_p3p;
  long long n_it = ctl->n_iterations;
  double orat = ctl->outlier_ratio;
  long long n = 0;
  int best = -1, best_idx = -1, consumed = -1, hyp_valid = 0;
  bool risky_seen = false;
  int next_base = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int base = 0; base < hyp && consumed < 0; base += 64) {
    next_base = base + 64;
    const int h = base + lane;
    const int vb = h < hyp ? valid[h] : 0;
    const bool v = (vb & 1) != 0;
    const int c = v ? counts[h] : -1;
    const unsigned long long vmask = __ballot(v), rmask = __ballot((vb & 2) != 0);
    int pm = c;                                   // inclusive prefix maximum
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(pm, off);
      if (lane >= off) pm = max(pm, o);
    }
    int epm = __shfl_up(pm, 1);
    if (lane == 0) epm = -1;
    epm = max(epm, best);
    const unsigned long long imask = __ballot(v && c > epm);
    const long long n_here = n + __popcll(vmask & lt);       // iterations counted before this lane's draw
    int cur = 0;
    for (;;) {
      const unsigned long long ge = cur >= 64 ? 0ull : ~((1ull << cur) - 1ull);
      const unsigned long long smask = __ballot(n_here >= n_it) & ge;   // the `while` test fails before this draw
      const unsigned long long emask = imask & ge;
      const int sp = smask ? __ffsll((long long)smask) - 1 : 64;
      const int ep = emask ? __ffsll((long long)emask) - 1 : 64;
      if (sp <= ep && sp < 64) {
        consumed = base + sp;
        n = __shfl(n_here, sp);
        risky_seen |= (rmask & ((1ull << sp) - 1ull)) != 0ull;
        break;
      }
      if (ep == 64) {
        n += __popcll(vmask);
        risky_seen |= rmask != 0ull;
        break;
      }
      best = __shfl(c, ep);
      best_idx = base + ep;
      {   // ransac.py:113-120
        double o = 1.0 - (double)best / (double)N;
        o = fmin(fmax(o, 0.01), 0.99);
        orat = o;
        n_it = table_lookup(table, table_len, max_it, o);
      }
      cur = ep + 1;
    }
    hyp_valid += __popcll(vmask);
  }
  int fault = 0;
  if (consumed < 0) {
    if (n >= n_it) consumed = hyp;                // the loop ends exactly behind the last sample of the batch
    else fault |= VO_FAULT_UNFINISHED;            // (also: no hypothesis had a solution)
  }
  // (valid hypotheses behind the point where the loop ended are not part of hyp_valid's meaning for the
  //  reference; the count over the whole batch is reported for diagnostics only)
  for (int base = next_base; base < hyp; base += 64) {
    const int h = base + lane;
    hyp_valid += __popcll(__ballot(h < hyp && (valid[h] & 1)));
  }
  if (risky_seen) fault |= VO_FAULT_RISKY_DRAW;
  if (!fault && best_idx < 0) fault |= VO_FAULT_UNFINISHED;
  if (fault) {
    if (lane == 0) {
      ctl->fault = fault;
      ctl->n_p3p = 0;
    }
    return;
  }
  if (lane < 9) ctl->best_pose[lane] = Rall[9 * best_idx + lane];
  if (lane < 3) ctl->best_pose[9 + lane] = tall[3 * best_idx + lane];
  const int wn = (N + 63) >> 6;
  for (int w = lane; w < wn; w += 64) best_mask[w] = masks[(size_t)best_idx * words + w];
  if (lane == 0) {
    ctl->n_iterations = n_it;
    ctl->outlier_ratio = orat;
    ctl->raw_pos += 7ull * (unsigned long long)consumed;
    ctl->best_idx = best_idx;
    ctl->best_count = best;
    ctl->consumed = consumed;
    ctl->hyp_valid = hyp_valid;
    ctl->n_done = n;
  }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rigid_inverse_3x4(const double* T, double* Ti) {
  // [R t] -> [R^T  -R^T t]
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) Ti[4 * r + c] = T[4 * c + r];
    Ti[4 * r + 3] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
  }
}

__device__ __forceinline__ void k_times(const double* K, const double* T, double* C) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) C[4 * r + c] = K[3 * r] * T[c] + K[3 * r + 1] * T[4 + c] + K[3 * r + 2] * T[8 + c];
}

// main.py:261-286 on the new frame's features (B), one workgroup:
//   phase bit 0: outliers[triangulate_inliers] = ~inliers; update_with_world_pose; reset_outliers;
//                compute_candidates (bearing angle >= threshold among state == 1)
//   phase bit 1: triangulate_candidates (one start pose per track) -> update_with_world_landmarks ->
//                _check_landmarks; step bookkeeping and the result record
__global__ __launch_bounds__(UP_T) void state_update_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat B,
                                                            const unsigned long long* __restrict__ best_mask,
                                                            vo_cam cam, double bearing_thr, int use_refined, int phases,
                                                            vo_step_result* __restrict__ res,
                                                            unsigned* __restrict__ seq_word, unsigned seq) {
  __shared__ int s_list[UP_T];
  __shared__ int s_n, s_cnt[4];
  const int tid = threadIdx.x;
  const int fault = ctl->fault;
  if (fault) {
    if (res && tid == 0) {
      res->fault = fault;
      res->n_features_in = ctl->n_in;
      res->n_tracked = ctl->n2;
      res->n_triangulated = ctl->n_tri;
      res->raw_pos = ctl->raw_pos;
      __threadfence_system();
      *seq_word = seq;
    }
    return;
  }
  const int n2 = ctl->n2, n_tri = ctl->n_tri;
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) s_n = 0;
  double Tcw[12], Twc[12], Tp[12];   // pose of the new frame both ways; world->camera pose of the frame being left
  if (phases & 1) {
    // update_from_matches (state.py:17-22): what was current becomes previous
#pragma unroll
    for (int k = 0; k < 12; ++k) Tp[k] = ctl->T_cw[k];
    if (use_refined < 0) {          // given by the host (bookkeeping entry point), both directions
#pragma unroll
      for (int k = 0; k < 12; ++k) Tcw[k] = ctl->T_in_cw[k];
#pragma unroll
      for (int k = 0; k < 12; ++k) Twc[k] = ctl->T_in_wc[k];
    } else {
      // world -> camera as estimated, camera -> world = its inverse (update_with_world_pose, state.py:38-50)
      const double* src = use_refined ? ctl->refined : ctl->best_pose;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Tcw[4 * r] = src[3 * r];
        Tcw[4 * r + 1] = src[3 * r + 1];
        Tcw[4 * r + 2] = src[3 * r + 2];
        Tcw[4 * r + 3] = src[9 + r];
      }
      rigid_inverse_3x4(Tcw, Twc);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) Tcw[k] = ctl->T_cw[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) Twc[k] = ctl->T_wc[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tp[k] = ctl->T_cw_prev[k];
  }
  __syncthreads();
  if (phases & 1) {
    for (int i = tid; i < n2; i += UP_T) {
      int st = B.state[i];
      const double u = B.kp64[2 * i], v = B.kp64[2 * i + 1];
      bool reset = false;
      if (st == 0) reset = true;                              // set_pose_for_new_tracks (features.py:224-237)
      if (i < n_tri && !((best_mask[i >> 6] >> (i & 63)) & 1ull)) {   // P3P outlier (main.py:261-262)
        st = 0;
        reset = true;
      }
      int cand = 0;
      if (reset) {
        if (st == 0 && B.state[i] != 0) {                     // reset_outliers (state.py:162-172)
          B.track[2 * i] = u;
          B.track[2 * i + 1] = v;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) B.pose[12 * i + k] = Twc[k];
        B.state[i] = 0;
      } else if (st == 1) {
        // _calculate_bearing_angle (state.py:174-219): rays through the track's first and last keypoint
        const double* P = B.pose + 12 * i;
        const double a = B.track[2 * i], b = B.track[2 * i + 1];
        const double* Ki = cam.Kinv;
        const double n1x = Ki[0] * a + Ki[1] * b + Ki[2], n1y = Ki[3] * a + Ki[4] * b + Ki[5],
                     n1z = Ki[6] * a + Ki[7] * b + Ki[8];
        const double n2x = Ki[0] * u + Ki[1] * v + Ki[2], n2y = Ki[3] * u + Ki[4] * v + Ki[5],
                     n2z = Ki[6] * u + Ki[7] * v + Ki[8];
        const double r1x = P[0] * n1x + P[1] * n1y + P[2] * n1z, r1y = P[4] * n1x + P[5] * n1y + P[6] * n1z,
                     r1z = P[8] * n1x + P[9] * n1y + P[10] * n1z;
        const double r2x = Twc[0] * n2x + Twc[1] * n2y + Twc[2] * n2z, r2y = Twc[4] * n2x + Twc[5] * n2y + Twc[6] * n2z,
                     r2z = Twc[8] * n2x + Twc[9] * n2y + Twc[10] * n2z;
        const double dot = r1x * r2x + r1y * r2y + r1z * r2z;
        const double l1 = sqrt(r1x * r1x + r1y * r1y + r1z * r1z), l2 = sqrt(r2x * r2x + r2y * r2y + r2z * r2z);
        const double ang = acos(dot / (l1 * l2));
        cand = ang >= bearing_thr ? 1 : 0;                    // (NaN compares false, as in NumPy)
      }
      B.cand[i] = (uint8_t)cand;
    }
    __syncthreads();                 // (every thread has read the old pose)
    if (tid < 12) {
      ctl->T_cw_prev[tid] = Tp[tid];
      ctl->T_wc_prev[tid] = ctl->T_wc[tid];
    }
    __syncthreads();
    if (tid < 12) {
      ctl->T_cw[tid] = Tcw[tid];
      ctl->T_wc[tid] = Twc[tid];
    }
    if (tid == 0) ctl->n = n2;       // the new frame is the current one from here on
  }
  if (!(phases & 2)) return;
  __syncthreads();
  // ---- candidates -> list (any order: every candidate writes its own slot) ----
  double C2[12];
  k_times(cam.K, Tcw, C2);
  int n_cand_total = 0;
  for (int base = 0; base < n2; base += UP_T) {
    const int i = base + tid;
    const bool c = i < n2 && B.cand[i] != 0;
    if (c) s_list[atomicAdd(&s_n, 1)] = i;
    __syncthreads();
    const int m = s_n;
    n_cand_total += m;
    if (tid < m) {
      const int k = s_list[tid];
      // proj1 = K inv(pose_start)[:3] (triangulation.py:53-56)
      double Ts[12], Ti[12], C1[12], X[3];
#pragma unroll
      for (int q = 0; q < 12; ++q) Ts[q] = B.pose[12 * k + q];
      rigid_inverse_3x4(Ts, Ti);
      k_times(cam.K, Ti, C1);
      vo_dlt::triangulate_point(C1, B.track[2 * k], B.track[2 * k + 1], C2, B.kp64[2 * k], B.kp64[2 * k + 1], X);
      B.land[3 * k] = X[0];                                    // update_with_world_landmarks (state.py:69-88)
      B.land[3 * k + 1] = X[1];
      B.land[3 * k + 2] = X[2];
      B.state[k] = 2;
    }
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
  }
  // ---- _check_landmarks (state.py:90-107) runs inside update_with_world_landmarks, i.e. only when there
  //      was a candidate (main.py:279-284) ----
  int dropped = 0, nland = 0;
  for (int i = tid; i < n2; i += UP_T) {
    int st = B.state[i];
    if (n_cand_total > 0) {
      const double x = B.land[3 * i], y = B.land[3 * i + 1], z = B.land[3 * i + 2];
      const double zc = Tcw[8] * x + Tcw[9] * y + Tcw[10] * z + Tcw[11];
      const double zp = Tp[8] * x + Tp[9] * y + Tp[10] * z + Tp[11];
      if (zc < 0.0 || zp < 0.0) {                              // (NaN landmarks compare false)
        const double nan = dnan();
        B.land[3 * i] = B.land[3 * i + 1] = B.land[3 * i + 2] = nan;
        B.state[i] = 0;
        st = 0;
        B.track[2 * i] = B.kp64[2 * i];
        B.track[2 * i + 1] = B.kp64[2 * i + 1];
#pragma unroll
        for (int k = 0; k < 12; ++k) B.pose[12 * i + k] = Twc[k];
        ++dropped;
      }
    }
    nland += st == 2 ? 1 : 0;
  }
  atomicAdd(&s_cnt[0], dropped);
  atomicAdd(&s_cnt[1], nland);
  __syncthreads();
  if (tid == 0) {
    ctl->n = n2;
    ctl->n_cand = n_cand_total;
    ctl->n_dropped = s_cnt[0];
    ctl->n_land = s_cnt[1];
    ctl->step += 1;
  }
  if (res) {
    if (tid < 9) {
      res->R[tid] = ctl->best_pose[tid];
      res->R_refined[tid] = use_refined > 0 ? ctl->refined[tid] : ctl->best_pose[tid];
    }
    if (tid < 3) {
      res->t[tid] = ctl->best_pose[9 + tid];
      res->t_refined[tid] = use_refined > 0 ? ctl->refined[9 + tid] : ctl->best_pose[9 + tid];
    }
    if (tid < 12) res->T_wc[tid] = Twc[tid];
    if (tid == 0) {
      res->n_tracked = n2;
      res->n_inliers = ctl->best_count;
      res->best_index = ctl->best_idx;
      res->hyp_valid = ctl->hyp_valid;
      res->ransac_iterations = ctl->n_done;
      res->draws_consumed = ctl->consumed;
      res->refine_iterations = use_refined > 0 ? (int)ctl->refined[12] : -1;
      res->refine_cost = use_refined > 0 ? ctl->refined[13] : 0.0;
      res->n_features_in = ctl->n_in;
      res->redetected = ctl->redetected;
      res->n_triangulated = n_tri;
      res->n_candidates = n_cand_total;
      res->n_dropped = s_cnt[0];
      res->n_landmarks = s_cnt[1];
      res->fault = 0;
      res->recovered = 0;
      res->raw_pos = ctl->raw_pos;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      *seq_word = seq;
    }
  }
}

}  // namespace

int vo_state_append(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat F, const double* d_det_kp, int n_det, double frac, int cap,
                    int debug_fault_every, int pose_mode) {
  {
    vo_prof_scope ps(ctx, VO_K_STATE_APPEND);
    hipLaunchKernelGGL(state_append_kernel, dim3(vo_cdiv(n_det > 0 ? n_det : 1, 256)), dim3(256), 0, ctx->stream, ctl, F,
                       d_det_kp, n_det, frac, cap, debug_fault_every, pose_mode);
  }
  return vo_check_launch(ctx, "state_append_kernel");
}

int vo_state_regroup_klt(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const float* d_next_xy,
                         const uint8_t* d_status, const float* d_err, float err_thr, int cap) {
  VO_REQUIRE(ctx, cap <= RG_T * RG_MAX_PER, "state_regroup: capacity %d above %d", cap, RG_T * RG_MAX_PER);
  {
    vo_prof_scope ps(ctx, VO_K_STATE_REGROUP);
    hipLaunchKernelGGL(state_regroup_kernel<false>, dim3(1), dim3(RG_T), 0, ctx->stream, ctl, A, B, d_next_xy, d_status,
                       d_err, err_thr, (const int*)nullptr, 0, (const double*)nullptr, 0, cap);
  }
  return vo_check_launch(ctx, "state_regroup_kernel");
}

int vo_state_regroup_pairs(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const int32_t* d_pairs, int M,
                           const double* d_new_kp, int n2_in, int cap) {
  VO_REQUIRE(ctx, cap <= RG_T * RG_MAX_PER && n2_in <= cap && M <= cap, "state_regroup: capacity exceeded");
  {
    vo_prof_scope ps(ctx, VO_K_STATE_REGROUP);
    hipLaunchKernelGGL(state_regroup_kernel<true>, dim3(1), dim3(RG_T), 0, ctx->stream, ctl, A, B, (const float*)nullptr,
                       (const uint8_t*)nullptr, (const float*)nullptr, 0.f, d_pairs, M, d_new_kp, n2_in, cap);
  }
  return vo_check_launch(ctx, "state_regroup_kernel");
}

int vo_state_ransac_replay(vo_ctx* ctx, vo_seq_ctl* ctl, const uint8_t* d_valid, const int32_t* d_counts,
                           const double* d_R, const double* d_t, const uint64_t* d_masks, int words, int hyp,
                           const double* d_thr_table, int table_len, int64_t max_iterations, uint64_t* d_best_mask) {
  {
    vo_prof_scope ps(ctx, VO_K_RANSAC_REPLAY);
    hipLaunchKernelGGL(ransac_replay_kernel, dim3(1), dim3(64), 0, ctx->stream, ctl, d_valid, d_counts, d_R, d_t,
                       (const unsigned long long*)d_masks, words, hyp, d_thr_table, table_len, (long long)max_iterations,
                       (unsigned long long*)d_best_mask);
  }
  return vo_check_launch(ctx, "ransac_replay_kernel");
}

int vo_state_update(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, const uint64_t* d_best_mask, vo_cam cam,
                    double bearing_thr, int use_refined, int phases, int cap, vo_step_result* m_result,
                    unsigned* m_seq, unsigned seq) {
  {
    vo_prof_scope ps(ctx, VO_K_STATE_UPDATE);
    hipLaunchKernelGGL(state_update_kernel, dim3(1), dim3(UP_T), 0, ctx->stream, ctl, B,
                       (const unsigned long long*)d_best_mask, cam, bearing_thr, use_refined, phases, m_result, m_seq,
                       seq);
  }
  return vo_check_launch(ctx, "state_update_kernel");
}

int64_t vo_ransac_table_lookup(const double* table, int table_len, int64_t max_iterations, double outlier_ratio) {
  return (int64_t)table_lookup(table, table_len, (long long)max_iterations, outlier_ratio);
}

// table[0] = f(0.01) (the clip's lower end), table[1 + j] = smallest outlier ratio in [0.01, 0.99] with
// f >= table[0] + j + 1, where f = ransac.py:58-67 evaluated with this host's libm; +inf when f never gets there.
void vo_ransac_build_table(double confidence, int s, int table_len, double* table) {
  auto f = [&](double o) { return (double)vo_ransac_num_iterations(confidence, o, s); };
  const double kmin = f(0.01);
  table[0] = kmin;
  const double fmax_ = f(0.99);
  double lo_start = 0.01;
  for (int j = 0; j < table_len; ++j) {
    const double target = kmin + j + 1;
    if (fmax_ < target) {
      table[1 + j] = INFINITY;
      continue;
    }
    // bisection on the bit patterns (positive doubles are ordered like their bits): f(lo) < target <= f(hi)
    uint64_t lo, hi;
    double a = lo_start, b = 0.99;
    memcpy(&lo, &a, 8);
    memcpy(&hi, &b, 8);
    if (f(a) >= target) {
      table[1 + j] = a;
      continue;
    }
    while (hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      double m;
      memcpy(&m, &mid, 8);
      if (f(m) >= target) hi = mid;
      else lo = mid;
    }
    double r;
    memcpy(&r, &hi, 8);
    table[1 + j] = r;
    memcpy(&lo_start, &lo, 8);
  }
}
