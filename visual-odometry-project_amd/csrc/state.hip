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
#include "state_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int RG_T = 1024;   // regroup: threads


using namespace vo_state_dev;

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
    for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + dst] = nan;
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
    for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + dst] = A.pose[(size_t)k * A.pitch + src];
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
                                                             const double* __restrict__ new_kp, int n2_in, int cap,
                                                             const int* __restrict__ d_M = nullptr,
                                                             const int* __restrict__ d_n2 = nullptr,
                                                             int* __restrict__ src_row = nullptr) {
  __shared__ unsigned long long s_wave[RG_T / 64 + 1];
  __shared__ unsigned s_matched[PAIRS ? 1024 : 1];   // bit per new keypoint (capacity 32768)
  if (ctl->fault) return;
  const int tid = threadIdx.x;
  if (PAIRS && d_M) {                  // counts that live on the device (SIFT tracker mode of the frame pipeline)
    M = *d_M;
    n2_in = min(*d_n2, cap);
  }
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
    if (PAIRS && src_row) src_row[pos[key]] = pairs[2 * j + 1];      // (which new keypoint ended up at this place)
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
      for (int q = 0; q < 12; ++q) B.pose[(size_t)q * B.pitch + p] = nan;
      if (src_row) src_row[p] = k;
      ++p;
    }
    n2 += (int)tot2;
  }
  if (tid == 0) {
    ctl->n2 = n2;
    ctl->n_tri = T0;
    ctl->n_mat = T1;
    ctl->n_new = T2;
    if (PAIRS && d_M) {
      ctl->n_in = n2_in;               // (the record's "features in": the new frame's keypoints)
      ctl->redetected = 0;
      ctl->det_ran = 0;
    }
    int fault = 0;
    if (!PAIRS && T0 < 8) fault = VO_FAULT_FEW_LANDMARKS;   // (population below what the device-side sampler handles)
    ctl->n_p3p = fault ? 0 : T0;
    if (fault) ctl->fault = fault;
  }
}


// (the body of state_regroup_klt_kernel: it returns early on several paths, the kernel's gates sit around it)
template <bool BYP>
__device__ __forceinline__ void regroup_klt_body(vo_seq_ctl* __restrict__ ctl, vo_feat A, vo_feat B,
                                                 const float* __restrict__ next_xy, const uint8_t* __restrict__ status,
                                                 const float* __restrict__ err, float err_thr, vo_append ap, int cap,
                                                 unsigned long long (&s_red)[2][4], int (&s_wcnt)[3][4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int RG_U = 16;
  // gate form 2 (vo_internal.h): the tracker's outputs are read, the positions and the count the next tracker reads are written
  // with agent-scope accesses
  constexpr bool byp = BYP;    // (a template parameter: as a run-time flag it cost the ungated kernel 1.7 us -- its sixteen-at-a-time
                               //  loads no longer went out together)
  auto ld_status = [&](int j) -> int { return byp ? (int)vo_ld_agent(&status[j]) : (int)status[j]; };
  auto ld_err = [&](int j) -> float { return byp ? vo_ld_agent(&err[j]) : err[j]; };
  const int start = blockIdx.x * 256;
  const int entry_fault = ctl->fault;
  int fault = entry_fault;
  const int n = ctl->n;
  if (ap.debug_fault_every > 0 && (ctl->step % ap.debug_fault_every) == ap.debug_fault_every - 1) fault |= VO_FAULT_FORCED;
  // `length < self._num_features * 0.8` (klt.py:208-212)
  const bool redetect = !fault && (double)n < (double)ctl->num_features * ap.frac;
  if (redetect && n + ap.n_det > cap) fault |= VO_FAULT_CAPACITY;
  if (redetect && ap.det_go && !ap.det_go[blockIdx.y]) fault |= VO_FAULT_NO_DETECTION;
  if (fault) {
    if (blockIdx.x == 0 && tid == 0) {
      if (!entry_fault) {               // raised here.  (A fault of an EARLIER step, still open: that step's counts stay --
        ctl->fault = fault;             //  a step waiting for its next batch of hypotheses, VO_FAULT_CONTINUE, reports them
        ctl->n_in = 0;                  //  when it closes.)
        ctl->redetected = 0;
      }
      ctl->few = 0;
      ctl->n_p3p = 0;                   // this step's hypothesis kernel has nothing to do
    }
    return;
  }
  const int n_in = n + (redetect ? ap.n_det : 0);
  if (start >= n_in && blockIdx.x != 0) return;
  // The flags of the first 16 x 256 items and this work item's own feature are requested together, ahead of the counting:
  // this kernel sits on the main chain AND in front of the tracker, and each dependent round trip here (there were up to
  // six: flags four items at a time, then the own feature behind the counts) is ~2 us of both cycles.  (Requesting them even
  // before the control block is read, at indices clamped by the capacity, measured no faster.)
  const int capm = max(n_in - 1, 0), pitm = max(n - 1, 0);
  int r_s8[RG_U], r_st[RG_U];
  float r_e[RG_U];
#pragma unroll
  for (int u = 0; u < RG_U; ++u) {
    const int j = u * 256 + tid;
    r_s8[u] = ld_status(min(j, capm));
    r_e[u] = ld_err(min(j, capm));
    r_st[u] = A.state[min(j, pitm)];
  }
  const int jo = start + tid, jq = min(jo, pitm), jx = min(jo, capm);
  double o_land[3], o_track[2], o_kp[2], o_pose[12];
  const float o_nx = byp ? vo_ld_agent(&next_xy[2 * jx]) : next_xy[2 * jx];
  const float o_ny = byp ? vo_ld_agent(&next_xy[2 * jx + 1]) : next_xy[2 * jx + 1];
#pragma unroll
  for (int k = 0; k < 3; ++k) o_land[k] = A.land[3 * jq + k];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    o_track[k] = A.track[2 * jq + k];
    o_kp[k] = A.kp64[2 * jq + k];
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) o_pose[k] = A.pose[(size_t)k * A.pitch + jq];
  auto key_of = [&](int j) -> int {      // 0 triangulated, 1 matched, 2 newly matched, 3 dropped
    // (unconditional loads at clamped indices, no short-circuit: the three requests of an item, and those of the
    //  following items, go out together instead of one dependent round trip after the other)
    const int jc = min(j, n_in - 1), js = min(j, max(n - 1, 0));
    const int s8 = ld_status(jc);
    const float e = ld_err(jc);
    const int st_raw = A.state[js];
    const int keep = (int)(j < n_in) & (int)(s8 != 0) & (int)(e < err_thr);
    const int st = j < n ? st_raw : 0;
    return keep ? (st == 2 ? 0 : (st == 1 ? 1 : 2)) : 3;
  };
  unsigned long long c_all = 0ull, c_before = 0ull;
  int my_key = 3;
  // (sixteen items' flags per work item requested together: the forward stream hands 2600-4600 items to this kernel,
  //  and four at a time were three to five dependent round trips -- on the main chain AND in front of the tracker)
  for (int base0 = 0; base0 < n_in; base0 += 256 * RG_U) {
    int keys[RG_U];
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const int j = base0 + u * 256 + tid;
      if (base0 == 0) {                        // (from the values requested at the top)
        const int keep = (int)(j < n_in) & (int)(r_s8[u] != 0) & (int)(r_e[u] < err_thr);
        const int st = j < n ? r_st[u] : 0;
        keys[u] = keep ? (st == 2 ? 0 : (st == 1 ? 1 : 2)) : 3;
      } else {
        keys[u] = key_of(j);
      }
    }
#pragma unroll
    for (int u = 0; u < RG_U; ++u) {
      const int base = base0 + u * 256;
      const int key = keys[u];                    // (3 beyond n_in)
      if (base == start) my_key = key;
      if (key < 3) {
        const unsigned long long one = 1ull << (21 * key);
        c_all += one;
        if (base < start) c_before += one;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    c_all += __shfl_xor(c_all, off);
    c_before += __shfl_xor(c_before, off);
  }
  // ranks inside the workgroup: one ballot per group
  int rank = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long m = __ballot(my_key == k);
    if (my_key == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[k][wv] = __popcll(m);
  }
  if (lane == 0) {
    s_red[0][wv] = c_all;
    s_red[1][wv] = c_before;
  }
  __syncthreads();
  const unsigned long long total = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
  const unsigned long long before = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  const int T0 = (int)(total & 0x1fffff), T1 = (int)((total >> 21) & 0x1fffff), T2 = (int)((total >> 42) & 0x1fffff);
  const int j = start + tid;
  if (my_key < 3) {
    int dst = (int)((before >> (21 * my_key)) & 0x1fffff) + rank;
    for (int w = 0; w < wv; ++w) dst += s_wcnt[my_key][w];
    dst += my_key == 0 ? 0 : (my_key == 1 ? T0 : T0 + T1);
    const double x = (double)o_nx, y = (double)o_ny;
    if (j < n) {
      // (write_group, from the registers filled above)
      const double nan = dnan();
      if (byp) {
        vo_st_agent(&B.kp[2 * dst], (float)x);
        vo_st_agent(&B.kp[2 * dst + 1], (float)y);
      } else {
        B.kp[2 * dst] = (float)x;
        B.kp[2 * dst + 1] = (float)y;
      }
      B.kp64[2 * dst] = x;
      B.kp64[2 * dst + 1] = y;
      B.cand[dst] = 0;
      if (my_key == 0) {       // triangulated: the landmark travels, the track data is over (matches.py:146-201)
        B.state[dst] = 2;
#pragma unroll
        for (int k = 0; k < 3; ++k) B.land[3 * dst + k] = o_land[k];
        B.track[2 * dst] = B.track[2 * dst + 1] = nan;
#pragma unroll
        for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + dst] = nan;
      } else {
        B.state[dst] = 1;
        B.land[3 * dst] = B.land[3 * dst + 1] = B.land[3 * dst + 2] = nan;
        B.track[2 * dst] = my_key == 1 ? o_track[0] : o_kp[0];     // matched before: the track goes on; newly matched: it
        B.track[2 * dst + 1] = my_key == 1 ? o_track[1] : o_kp[1]; // starts at frame 1's keypoint
#pragma unroll
        for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + dst] = o_pose[k];
      }
    } else {
      // a keypoint the detector found on the old frame, tracked: "newly matched" (matches.py:62-110)
      const int d = j - n;
      if (byp) {
        vo_st_agent(&B.kp[2 * dst], (float)x);
        vo_st_agent(&B.kp[2 * dst + 1], (float)y);
      } else {
        B.kp[2 * dst] = (float)x;
        B.kp[2 * dst + 1] = (float)y;
      }
      B.kp64[2 * dst] = x;
      B.kp64[2 * dst + 1] = y;
      B.cand[dst] = 0;
      B.state[dst] = 1;
      B.land[3 * dst] = B.land[3 * dst + 1] = B.land[3 * dst + 2] = dnan();
      B.track[2 * dst] = (double)(float)ap.det_kp[2 * d];
      B.track[2 * dst + 1] = (double)(float)ap.det_kp[2 * d + 1];
#pragma unroll
      for (int k = 0; k < 12; ++k)   // np.eye(4) (klt.py:148-153), or the pose of the frame the keypoints were found on
        B.pose[(size_t)k * B.pitch + dst] = ap.pose_mode ? ctl->T_wc[k] : ((k == 0 || k == 5 || k == 10) ? 1.0 : 0.0);
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    ctl->n_in = n_in;
    ctl->redetected = redetect ? 1 : 0;
    ctl->det_ran = ap.det_go ? ap.det_go[blockIdx.y] : 1;
    if (byp) vo_st_agent(&ctl->n2, T0 + T1 + T2);
    else ctl->n2 = T0 + T1 + T2;
    ctl->n_tri = T0;
    ctl->n_mat = T1;
    ctl->n_new = T2;
    const int few = T0 < 8 ? VO_FAULT_FEW_LANDMARKS : 0;   // (population below what the device-side sampler handles)
    ctl->n_p3p = few ? 0 : T0;
    ctl->few = few;                    // (not into ctl->fault: this launch's other workgroups read that word on entry)
  }
}

// ---------------------------------------------------------------------------------------------
// klt.py:191-280 behind the tracker + Matches.__init__ (matches.py:26-212) for identity matches, many workgroups,
// none of which waits for another: every workgroup of 256 items counts the group sizes of ALL items itself (a few
// KB of flags, coalesced, from L2) and the sizes before its own first item, so an item's place in the new frame
//     base(group) + #(same group before the workgroup) + #(same group before it inside the workgroup)
// needs no communication.  The re-detect branch (klt.py:207-230 -> update_features, klt.py:117-189) is not a
// copy: when fewer than frac * _num_features features are left, items n .. n + n_det - 1 ARE the detector's
// keypoints of the old frame (state 0, landmark NaN, track start = the keypoint, start pose np.eye(4)) -- the
// tracker kernel read its points the same way (vo_klt_source).
template <bool BYP>
__global__ __launch_bounds__(256) void state_regroup_klt_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat A, vo_feat B,
                                                                const float* __restrict__ next_xy,
                                                                const uint8_t* __restrict__ status,
                                                                const float* __restrict__ err, float err_thr,
                                                                vo_append ap, int cap) {
  __shared__ unsigned long long s_red[2][4];
  __shared__ int s_wcnt[3][4];
  const int tid = threadIdx.x;
  if (blockIdx.y != 0) {               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    ctl += q;
    A = vo_feat_seq(A, q);
    B = vo_feat_seq(B, q);
    next_xy += q * (size_t)cap * 2;
    status += q * (size_t)cap;
    err += q * (size_t)cap;
    ap.det_kp += q * ap.det_stride;
  }
  // device-side gate: this flight's tracker has published its end (instead of a stream event; vo_seq_ctl)
  if (ap.gate_klt_want) {
    if (!vo_gate_wait(&ctl->gate_klt, ap.gate_klt_want, ap.gate_mode != 2) && tid == 0) atomicOr(&ctl->fault, (int)VO_FAULT_GATE);
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) {
    ctl->ts[1] = wall_clock64();
    ctl->ts[6] = ctl->ts[0];          // this step's tracker start (the next step's tracker overwrites ts[0] meanwhile)
  }
  regroup_klt_body<BYP>(ctl, A, B, next_xy, status, err, err_thr, ap, cap, s_red, s_wcnt);
  if (ap.gate_regroup_set) {          // every workgroup arrives, whatever it did: the last one opens the tracker's gate
    if (ap.gate_mode == 2) {
      vo_stores_done();
      __syncthreads();
    } else {
      __syncthreads();
      __threadfence();
    }
    if (tid == 0) vo_gate_arrive(&ctl->gate_regroup_cnt, gridDim.x, &ctl->gate_regroup, ap.gate_regroup_set, ap.gate_mode != 2);
  }
}


// ---------------------------------------------------------------------------------------------
// main.py:261-268 as a kernel of its own (vo_pipeline_bookkeeping and the host recovery path; inside the frame loop
// the same per-feature function runs at the end of the pose kernel, refine.hip).
// use_refined: 1 the refinement's pose, 0 the accepted hypothesis, -1 the pose the host put into ctl->T_in_*.
__global__ __launch_bounds__(256) void state_candidates_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat B,
                                                               const unsigned long long* __restrict__ best_mask,
                                                               vo_cam cam, double bearing_thr, int use_refined, int words) {
  if (blockIdx.y != 0) {               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    ctl += q;
    B = vo_feat_seq(B, q);
    best_mask += q * (size_t)words;
  }
  if (ctl->fault) return;
  const int tid = threadIdx.x;
  const int n2 = ctl->n2, n_tri = ctl->n_tri;
  double Tcw[12], Twc[12];
  if (use_refined < 0) {
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
  const int i = blockIdx.x * 256 + tid;
  const int cand = i < n2 ? candidate_feature(B, i, n_tri, best_mask, cam, Twc, bearing_thr) : 0;
  const unsigned long long cm = __ballot(cand != 0);
  if ((tid & 63) == 0 && cm) atomicAdd(&ctl->n_cand, __popcll(cm));
  if (blockIdx.x == 0) {
    if (tid < 12) {
      double ncw = 0.0, nwc = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        ncw = tid == k ? Tcw[k] : ncw;
        nwc = tid == k ? Twc[k] : nwc;
      }
      commit_pose(ctl, tid, ncw, nwc);
    }
    if (tid == 0) ctl->n = n2;       // the new frame is the current one from here on
  }
}

// main.py:279-286: triangulate_candidates (triangulation.py:38-86, one start pose per track) ->
// update_with_world_landmarks (state.py:69-88) -> _check_landmarks (state.py:90-107, only when there was a
// candidate); the workgroup that finishes last closes the step and writes its result record.
__global__ __launch_bounds__(256) void state_landmarks_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat B, vo_cam cam,
                                                              int use_refined, vo_step_result* __restrict__ res,
                                                              unsigned* __restrict__ seq_word, unsigned seq) {
  __shared__ int s_cnt[2];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  if (blockIdx.y != 0) {               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    ctl += q;
    B = vo_feat_seq(B, q);
    if (res) {
      res += q;
      seq_word += q;
    }
  }
  if (blockIdx.x == 0 && tid == 0) ctl->ts[4] = wall_clock64();
  const int fault = ctl->fault;
  if (fault) {
    if (res && blockIdx.x == 0 && tid == 0) write_fault_record(ctl, fault, res, seq_word, seq);
    return;
  }
  const int n2 = ctl->n2, n_cand = ctl->n_cand;
  if (tid < 2) s_cnt[tid] = 0;
  double Tcw[12], Twc[12], Tp[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tcw[k] = ctl->T_cw[k];
#pragma unroll
  for (int k = 0; k < 12; ++k) Twc[k] = ctl->T_wc[k];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tp[k] = ctl->T_cw_prev[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  int dropped = 0, land = 0;
  if (i < n2) {
    int st = B.state[i];
    double X[3] = {B.land[3 * i], B.land[3 * i + 1], B.land[3 * i + 2]};
    if (B.cand[i]) {
      // proj1 = K inv(pose_start)[:3], proj2 = K inv(current_pose)[:3] (triangulation.py:53-57)
      double Ts[12], Ti[12], C1[12], C2[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) Ts[q] = B.pose[(size_t)q * B.pitch + i];
      rigid_inverse_3x4(Ts, Ti);
      k_times(cam.K, Ti, C1);
      k_times(cam.K, Tcw, C2);
      vo_dlt::triangulate_point(C1, B.track[2 * i], B.track[2 * i + 1], C2, B.kp64[2 * i], B.kp64[2 * i + 1], X);
      B.land[3 * i] = X[0];
      B.land[3 * i + 1] = X[1];
      B.land[3 * i + 2] = X[2];
      st = 2;
      B.state[i] = 2;
    }
    if (n_cand > 0) {
      const double zc = Tcw[8] * X[0] + Tcw[9] * X[1] + Tcw[10] * X[2] + Tcw[11];
      const double zp = Tp[8] * X[0] + Tp[9] * X[1] + Tp[10] * X[2] + Tp[11];
      if (zc < 0.0 || zp < 0.0) {                              // behind a camera (NaN landmarks compare false)
        const double nan = dnan();
        B.land[3 * i] = B.land[3 * i + 1] = B.land[3 * i + 2] = nan;
        B.state[i] = 0;
        st = 0;
        B.track[2 * i] = B.kp64[2 * i];
        B.track[2 * i + 1] = B.kp64[2 * i + 1];
#pragma unroll
        for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + i] = Twc[k];
        dropped = 1;
      }
    }
    land = st == 2 ? 1 : 0;
  }
  {
    const int d = __popcll(__ballot(dropped != 0)), l = __popcll(__ballot(land != 0));
    if ((tid & 63) == 0) {
      if (d) atomicAdd(&s_cnt[0], d);
      if (l) atomicAdd(&s_cnt[1], l);
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_cnt[0]) atomicAdd(&ctl->n_dropped, s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&ctl->n_land, s_cnt[1]);
    __threadfence();
    s_last = atomicAdd(&ctl->done, 1) == (int)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: every other one's counts are in ----
  const int n_dropped = atomicAdd(&ctl->n_dropped, 0), n_land = atomicAdd(&ctl->n_land, 0);
  if (tid == 0) ctl->step += 1;
  if (!res) return;
  write_step_record(ctl, tid, use_refined, n2, n_cand, n_dropped, n_land, atomicAdd(&ctl->ts[4], 0ull), res, seq_word, seq);
}


// The walk (main.py:261-268) and the landmark stage (main.py:279-286) of one frame in ONE launch -- the frame loop's form;
// state_candidates_kernel + state_landmarks_kernel stay for vo_pipeline_bookkeeping and the host path.  A feature's walk, its
// triangulation and its cheirality test use nothing of any other feature, with one exception: _check_landmarks runs only
// when the FRAME has a candidate (state.py:90-107 is called from the `if candidates` branch, main.py:279).  A workgroup
// that holds a candidate knows that; one that does not (the features come grouped, the triangulated ones first: the first
// workgroups never hold one) leaves the features it would drop in `pend`, and the workgroup that arrives last -- it knows
// the frame's count -- drops them.  The same workgroup commits the pose (every other one has read the previous pose by
// then), closes the step and writes its record.  One launch less on the step's dependent chain (measured: neutral against the
// walk as its own launch -- a same-stream kernel boundary is 1-2 us and the fused kernel is that much longer -- and 5.6 us
// better than the walk on the pose kernel's one compute unit).
__global__ __launch_bounds__(256) void state_walk_landmarks_kernel(vo_seq_ctl* __restrict__ ctl, vo_feat B,
                                                                   const unsigned long long* __restrict__ best_mask, int words,
                                                                   vo_cam cam, double bearing_thr, int use_refined,
                                                                   int* __restrict__ pend, vo_step_result* __restrict__ res,
                                                                   unsigned* __restrict__ seq_word, unsigned seq, int rec_fence) {
  __shared__ int s_cnt[3];             // dropped, landmarks, candidates of this workgroup
  __shared__ int s_last;
  const int tid = threadIdx.x;
  if (blockIdx.y != 0) {               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    ctl += q;
    pend += q * (size_t)B.pitch;
    B = vo_feat_seq(B, q);
    best_mask += q * (size_t)words;
    if (res) {
      res += q;
      seq_word += q;
    }
  }
  if (blockIdx.x == 0 && tid == 0) ctl->ts[4] = wall_clock64();
  const int fault = ctl->fault;
  if (fault) {
    if (res && blockIdx.x == 0 && tid == 0) write_fault_record(ctl, fault, res, seq_word, seq);
    return;
  }
  const int n2 = ctl->n2, n_tri = ctl->n_tri;
  if (tid < 3) s_cnt[tid] = 0;
  // world -> camera as the pose kernel left it, camera -> world = its inverse (update_with_world_pose, state.py:38-50);
  // Tp: the pose that is still the current one -- the previous one once this frame's is committed
  double Tcw[12], Twc[12], Tp[12];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Tcw[4 * r] = ctl->refined[3 * r];
    Tcw[4 * r + 1] = ctl->refined[3 * r + 1];
    Tcw[4 * r + 2] = ctl->refined[3 * r + 2];
    Tcw[4 * r + 3] = ctl->refined[9 + r];
  }
  rigid_inverse_3x4(Tcw, Twc);
#pragma unroll
  for (int k = 0; k < 12; ++k) Tp[k] = ctl->T_cw[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  int cand = 0, st = 0;
  double X[3] = {0.0, 0.0, 0.0};
  if (i < n2) {
    // Everything the feature's walk, triangulation and test read, requested together: ONE round trip.  (candidate_feature
    // followed by the triangulation as state_landmarks_kernel has it reads the start pose twice, the second time behind the
    // walk's stores: two more dependent round trips on the step's chain.)
    const int st0 = B.state[i];
    const double u = B.kp64[2 * i], v = B.kp64[2 * i + 1];
    double Ts[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) Ts[q] = B.pose[(size_t)q * B.pitch + i];
    const double ta = B.track[2 * i], tb = B.track[2 * i + 1];
    X[0] = B.land[3 * i];
    X[1] = B.land[3 * i + 1];
    X[2] = B.land[3 * i + 2];
    const bool p3p_out = i < n_tri && !((best_mask[i >> 6] >> (i & 63)) & 1ull);
    // ---- candidate_feature (state_device.h), from the registers: main.py:261-268 ----
    st = st0;
    bool reset = st0 == 0;
    if (p3p_out) {                     // P3P outlier (main.py:261-262)
      st = 0;
      reset = true;
    }
    if (reset) {
      if (st0 != 0) {
        B.track[2 * i] = u;
        B.track[2 * i + 1] = v;
        B.state[i] = 0;
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + i] = Twc[k];
      st = 0;
    } else if (st == 1) {
      const double* Ki = cam.Kinv;
      const double n1x = Ki[0] * ta + Ki[1] * tb + Ki[2], n1y = Ki[3] * ta + Ki[4] * tb + Ki[5],
                   n1z = Ki[6] * ta + Ki[7] * tb + Ki[8];
      const double n2x = Ki[0] * u + Ki[1] * v + Ki[2], n2y = Ki[3] * u + Ki[4] * v + Ki[5],
                   n2z = Ki[6] * u + Ki[7] * v + Ki[8];
      const double r1x = Ts[0] * n1x + Ts[1] * n1y + Ts[2] * n1z, r1y = Ts[4] * n1x + Ts[5] * n1y + Ts[6] * n1z,
                   r1z = Ts[8] * n1x + Ts[9] * n1y + Ts[10] * n1z;
      const double r2x = Twc[0] * n2x + Twc[1] * n2y + Twc[2] * n2z, r2y = Twc[4] * n2x + Twc[5] * n2y + Twc[6] * n2z,
                   r2z = Twc[8] * n2x + Twc[9] * n2y + Twc[10] * n2z;
      const double dot = r1x * r2x + r1y * r2y + r1z * r2z;
      const double l1 = sqrt(r1x * r1x + r1y * r1y + r1z * r1z), l2 = sqrt(r2x * r2x + r2y * r2y + r2z * r2z);
      const double ang = acos(dot / (l1 * l2));
      cand = ang >= bearing_thr ? 1 : 0;                      // (NaN compares false, as in NumPy)
    }
    B.cand[i] = (uint8_t)cand;
    if (cand) {
      // proj1 = K inv(pose_start)[:3], proj2 = K inv(current_pose)[:3] (triangulation.py:53-57)
      double Ti[12], C1[12], C2[12];
      rigid_inverse_3x4(Ts, Ti);
      k_times(cam.K, Ti, C1);
      k_times(cam.K, Tcw, C2);
      vo_dlt::triangulate_point(C1, ta, tb, C2, u, v, X);
      B.land[3 * i] = X[0];
      B.land[3 * i + 1] = X[1];
      B.land[3 * i + 2] = X[2];
      st = 2;
      B.state[i] = 2;
    }
  }
  {
    const unsigned long long cm = __ballot(cand != 0);
    if ((tid & 63) == 0 && cm) atomicAdd(&s_cnt[2], __popcll(cm));
  }
  __syncthreads();
  const int local_cand = s_cnt[2];
  // behind a camera (NaN landmarks compare false): the track restarts here -- _check_landmarks, state.py:90-107
  auto drop = [&](int f) {
    const double nan = dnan();
    B.land[3 * f] = B.land[3 * f + 1] = B.land[3 * f + 2] = nan;
    B.state[f] = 0;
    B.track[2 * f] = B.kp64[2 * f];
    B.track[2 * f + 1] = B.kp64[2 * f + 1];
#pragma unroll
    for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + f] = Twc[k];
  };
  int dropped = 0;
  if (i < n2) {
    const double zc = Tcw[8] * X[0] + Tcw[9] * X[1] + Tcw[10] * X[2] + Tcw[11];
    const double zp = Tp[8] * X[0] + Tp[9] * X[1] + Tp[10] * X[2] + Tp[11];
    if (zc < 0.0 || zp < 0.0) {
      if (local_cand > 0) {
        drop(i);
        st = 0;
        dropped = 1;
      } else {
        pend[atomicAdd(&ctl->n_pend, 1)] = i;      // (still counted as what it is; the closing workgroup corrects the counts)
      }
    }
  }
  {
    const int d = __popcll(__ballot(dropped != 0)), l = __popcll(__ballot(i < n2 && st == 2));
    if ((tid & 63) == 0) {
      if (d) atomicAdd(&s_cnt[0], d);
      if (l) atomicAdd(&s_cnt[1], l);
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_cnt[0]) atomicAdd(&ctl->n_dropped, s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&ctl->n_land, s_cnt[1]);
    if (local_cand) atomicAdd(&ctl->n_cand, local_cand);
    __threadfence();
    s_last = atomicAdd(&ctl->done, 1) == (int)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: every other one's counts, features and pending verdicts are in ----
  __threadfence();
  const int n_cand = atomicAdd(&ctl->n_cand, 0), n_pend = atomicAdd(&ctl->n_pend, 0);
  int n_dropped = atomicAdd(&ctl->n_dropped, 0), n_land = atomicAdd(&ctl->n_land, 0);
  if (n_cand > 0 && n_pend > 0) {
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    for (int k = tid; k < n_pend; k += 256) {
      const int f = pend[k];
      if (B.state[f] == 2) atomicAdd(&s_cnt[1], 1);
      drop(f);
      atomicAdd(&s_cnt[0], 1);
    }
    __syncthreads();
    n_dropped += s_cnt[0];
    n_land -= s_cnt[1];
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
  if (tid == 0) {
    ctl->n = n2;                       // the new frame is the current one from here on
    ctl->step += 1;
  }
  if (!res) return;
  __syncthreads();                     // (the record reads the committed pose)
  write_step_record(ctl, tid, use_refined, n2, n_cand, n_dropped, n_land, atomicAdd(&ctl->ts[4], 0ull), res, seq_word, seq,
                    rec_fence != 0);
}

}  // namespace

int vo_state_regroup_klt(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const float* d_next_xy,
                         const uint8_t* d_status, const float* d_err, float err_thr, vo_append ap, int cap, int S) {
  {
    vo_prof_scope ps(ctx, VO_K_STATE_REGROUP);
    if (ap.gate_mode == 2 && ap.gate_klt_want != 0u)
      vo_launch_stop(ctx, state_regroup_klt_kernel<true>, dim3(vo_cdiv(cap, 256), S), dim3(256), 0, ctx->stream, ctl, A, B,
                     d_next_xy, d_status, d_err, err_thr, ap, cap);
    else
      vo_launch_stop(ctx, state_regroup_klt_kernel<false>, dim3(vo_cdiv(cap, 256), S), dim3(256), 0, ctx->stream, ctl, A, B,
                     d_next_xy, d_status, d_err, err_thr, ap, cap);
  }
  return vo_check_launch(ctx, "state_regroup_klt_kernel");
}

int vo_state_regroup_pairs(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const int32_t* d_pairs, int M,
                           const double* d_new_kp, int n2_in, int cap, const int32_t* d_M, const int32_t* d_n2,
                           int32_t* d_src_row) {
  VO_REQUIRE(ctx, cap <= RG_T * RG_MAX_PER && n2_in <= cap && M <= cap, "state_regroup: capacity exceeded");
  {
    vo_prof_scope ps(ctx, VO_K_STATE_REGROUP);
    hipLaunchKernelGGL(state_regroup_kernel<true>, dim3(1), dim3(RG_T), 0, ctx->stream, ctl, A, B, (const float*)nullptr,
                       (const uint8_t*)nullptr, (const float*)nullptr, 0.f, d_pairs, M, d_new_kp, n2_in, cap,
                       (const int*)d_M, (const int*)d_n2, (int*)d_src_row);
  }
  return vo_check_launch(ctx, "state_regroup_kernel");
}

int vo_state_candidates(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, const uint64_t* d_best_mask, vo_cam cam,
                        double bearing_thr, int use_refined, int cap, int S, int words) {
  {
    vo_prof_scope ps(ctx, VO_K_STATE_CANDIDATES);
    hipLaunchKernelGGL(state_candidates_kernel, dim3(vo_cdiv(cap, 256), S), dim3(256), 0, ctx->stream, ctl, B,
                       (const unsigned long long*)d_best_mask, cam, bearing_thr, use_refined, words);
  }
  return vo_check_launch(ctx, "state_candidates_kernel");
}

int vo_state_walk_landmarks(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, const uint64_t* d_best_mask, int words, vo_cam cam,
                            double bearing_thr, int use_refined, int cap, int32_t* d_pend, vo_step_result* m_result,
                            unsigned* m_seq, unsigned seq, int S) {
  VO_REQUIRE(ctx, d_pend != nullptr && cap <= B.pitch, "state_walk_landmarks: bad arguments");
  {
    vo_prof_scope ps(ctx, VO_K_STATE_LANDMARKS);
    // VO_RECORD_FENCE=1: system-scope fences between the record's lines and its closing words (see write_step_record)
    static const int rec_fence = getenv("VO_RECORD_FENCE") ? atoi(getenv("VO_RECORD_FENCE")) : 0;
    hipLaunchKernelGGL(state_walk_landmarks_kernel, dim3(vo_cdiv(cap, 256), S), dim3(256), 0, ctx->stream, ctl, B,
                       (const unsigned long long*)d_best_mask, words, cam, bearing_thr, use_refined, (int*)d_pend, m_result,
                       m_seq, seq, rec_fence);
  }
  return vo_check_launch(ctx, "state_walk_landmarks_kernel");
}

int vo_state_landmarks(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, vo_cam cam, int use_refined, int cap,
                       vo_step_result* m_result, unsigned* m_seq, unsigned seq, int S) {
  {
    vo_prof_scope ps(ctx, VO_K_STATE_LANDMARKS);
    hipLaunchKernelGGL(state_landmarks_kernel, dim3(vo_cdiv(cap, 256), S), dim3(256), 0, ctx->stream, ctl, B, cam,
                       use_refined, m_result, m_seq, seq);
  }
  return vo_check_launch(ctx, "state_landmarks_kernel");
}

int64_t vo_ransac_table_lookup(const double* table, int table_len, int64_t max_iterations, double outlier_ratio) {
  return (int64_t)vo_state_dev::table_lookup(table, table_len, (long long)max_iterations, outlier_ratio);
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
