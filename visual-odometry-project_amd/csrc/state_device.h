// Device functions of the frame state shared by state.hip (stand-alone kernels) and refine.hip (the frame
// loop's pose kernel, which runs the RANSAC replay in front of the refinement and the candidate selection
// behind it so that three launches of the dependent chain become one).
#pragma once
#include <cstddef>
#include "vo_state.h"

#pragma clang fp contract(off)

namespace vo_state_dev {

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000ll); }

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

// The same count by all 64 lanes of a wave together (orat uniform): 64 blocks of 64 thresholds, one probe per lane and
// level -- two dependent reads instead of twelve on the replay's serial path.  Tables above 4096 entries: bisection.
__device__ __forceinline__ long long table_lookup_wave(const double* table, int len, long long max_it, double orat, int lane) {
  if (len > 4096) return table_lookup(table, len, max_it, orat);
  const int nb = (len + 63) >> 6;
  const bool whole = lane < nb && table[1 + min(lane * 64 + 63, len - 1)] <= orat;   // every threshold of block `lane`
  const int full = __popcll(__ballot(whole));                                         // (sorted: a prefix of the blocks)
  const int idx = full * 64 + lane;
  const bool one = idx < len && table[1 + idx] <= orat;
  const int lo = min(full * 64, len) + __popcll(__ballot(one));
  const long long k = lo == len ? 0x7fffffffffffffffll : (long long)table[0] + lo;
  return (max_it >= 0 && max_it < k) ? max_it : k;
}

using replay_args = ::vo_replay_args;

constexpr int RP_TABLE_LDS = 4097;

// ONE WAVE walks the batch of hypotheses through the reference's loop (ransac.py:90-121):
//     while n < n_iterations: draw; model None -> continue; count; strictly better -> keep, adapt; n += 1
// 64 hypotheses per round: a prefix maximum finds the hypotheses that improve on everything before them,
// and between two such events n_iterations is constant, so the place where the loop ends is a ballot.
// The flags and counts of 16 rounds are requested together; tb: the threshold table (table_len + 1 doubles, in
// LDS when the caller staged it there).  Leaves the accepted pose in ctl->best_pose, its mask row in a.best_mask, the loop's bookkeeping in ctl; on a
// step the device cannot finish alone, ctl->fault.  Must be called by all 64 lanes of the wave.
// flags and counts of the 1024 hypotheses from sbase on, 16 per lane, all requests in flight together
struct replay_chunk {
  int vbs[16], cs[16];
};
__device__ __forceinline__ void replay_fetch(const replay_args& a, int lane, int sbase, replay_chunk& ch) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int h = min(sbase + 64 * q + lane, a.hyp - 1);
    ch.vbs[q] = a.valid[h];
    ch.cs[q] = a.counts[h];
  }
}

// the control-block fields the replay starts from, requested together (by the pose kernel in front of its first barrier:
// read inside replay_wave they were one more dependent round trip at the head of the step's longest kernel)
struct replay_head {
  int n_p3p, cont, consumed, best_idx, best_count, hyp_valid, step;
  long long n_iterations, n_done;
  double outlier_ratio;
};
__device__ __forceinline__ replay_head replay_prefetch(const vo_seq_ctl* __restrict__ ctl) {
  replay_head h;
  h.n_p3p = ctl->n_p3p;
  h.cont = ctl->cont;
  h.consumed = ctl->consumed;
  h.best_idx = ctl->best_idx;
  h.best_count = ctl->best_count;
  h.hyp_valid = ctl->hyp_valid;
  h.step = ctl->step;
  h.n_iterations = ctl->n_iterations;
  h.n_done = ctl->n_done;
  h.outlier_ratio = ctl->outlier_ratio;
  return h;
}

// ch: working storage; fetched0: it already holds the chunk of sbase = 0 (the pose kernel requests it in front of its
// staging of the table and the coordinates, so that the replay does not start with a memory round trip of its own)
// s_best (12 doubles) / s_mask (s_mask_words words), optional, LDS: the accepted pose and its mask row are left there as well
// (the refinement behind the caller's barrier starts from them without a trip through memory); *s_mask_ok = 1 when s_mask holds the row
__device__ __forceinline__ void replay_wave(vo_seq_ctl* __restrict__ ctl, const replay_args& a, int lane,
                                            const double* tb, replay_chunk& ch, bool fetched0, int debug_fault_every = 0,
                                            const replay_head* head = nullptr, double* s_best = nullptr,
                                            unsigned long long* s_mask = nullptr, int s_mask_words = 0, int* s_mask_ok = nullptr) {
  if (lane == 0) {            // counters the bookkeeping kernels of this step add to
    ctl->n_cand = 0;
    ctl->n_dropped = 0;
    ctl->n_land = 0;
    ctl->done = 0;
  }
  const replay_head hd = head ? *head : replay_prefetch(ctl);
  const int hyp = a.hyp;
  const int N = hd.n_p3p;
  long long n_it = hd.n_iterations;
  double orat = hd.outlier_ratio;
  // cont > 0: the loop has walked `cont` batches of `hyp` samples already (VO_FAULT_CONTINUE) and goes on where it stopped
  const int cont = hd.cont;
  const int consumed_before = cont ? hd.consumed : 0;
  const int best_idx_before = cont ? hd.best_idx : -1;
  long long n = cont ? hd.n_done : 0;
  int best = cont ? hd.best_count : -1, best_idx = -1, consumed = -1, hyp_valid = cont ? hd.hyp_valid : 0;
  if (!cont && lane == 0) {
    ctl->n_iterations0 = n_it;
    ctl->outlier_ratio0 = orat;
  }
  bool risky_seen = false;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int sbase = 0; sbase < hyp; sbase += 1024) {
    if (!(fetched0 && sbase == 0)) replay_fetch(a, lane, sbase, ch);
    const int (&vbs)[16] = ch.vbs;
    const int (&cs)[16] = ch.cs;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int base = sbase + 64 * q;
      if (base >= hyp) break;
      const int h = base + lane;
      const int vb = h < hyp ? vbs[q] : 0;
      const bool v = (vb & 1) != 0;
      const unsigned long long vmask = __ballot(v);
      hyp_valid += __popcll(vmask);
      if (consumed >= 0) continue;                // (the loop has ended; only the statistics go on)
      const int c = v ? cs[q] : -1;
      const unsigned long long rmask = __ballot((vb & 2) != 0);
      const long long n_here = n + __popcll(vmask & lt);       // iterations counted before this lane's draw
      int cur = 0;
      for (;;) {
        const unsigned long long ge = cur >= 64 ? 0ull : ~((1ull << cur) - 1ull);
        const unsigned long long smask = __ballot(n_here >= n_it) & ge;   // the `while` test fails before this draw
        // The hypotheses that improve on everything before them, one at a time: the first lane from `cur` on whose count
        // beats the running best is one (every lane before it does not), and it becomes the running best.  (A prefix
        // maximum over the 64 counts said the same for all lanes at once, at six dependent cross-lane steps per round
        // whether or not the round held an improvement; most do not.)
        const unsigned long long emask = __ballot(c > best) & ge;
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
          n_it = table_lookup_wave(tb, a.table_len, a.max_it, o, lane);
        }
        cur = ep + 1;
      }
    }
  }
  int fault = 0;
  bool more = false;                              // the rule wants samples beyond this batch
  if (consumed < 0) {
    if (n >= n_it) consumed = hyp;                // the loop ends exactly behind the last sample of the batch
    else more = true;
  }
  if (risky_seen) fault |= VO_FAULT_RISKY_DRAW;
  if (debug_fault_every > 0 && !more && (hd.step % debug_fault_every) == debug_fault_every - 1) fault |= VO_FAULT_FORCED;
  // a bound beyond the table (an unbounded max_iterations and a ratio whose bound the table does not hold): the host's loop
  if (!fault && more && n_it == 0x7fffffffffffffffll) fault |= VO_FAULT_UNFINISHED;
  if (!fault && !more && best_idx < 0 && best_idx_before < 0) fault |= VO_FAULT_UNFINISHED;   // (no hypothesis had a solution)
  if (fault) {
    if (lane == 0) {
      ctl->fault = fault;
      ctl->n_p3p = 0;
    }
    return;
  }
  const int wn = (N + 63) >> 6;
  const bool lds_mask = s_mask != nullptr && wn <= s_mask_words;
  if (best_idx >= 0) {                            // the best so far lives in this batch: its pose and mask row
    double pv = 0.0;
    if (lane < 12) {
      pv = lane < 9 ? a.R[9 * best_idx + lane] : a.t[3 * best_idx + (lane - 9)];
      ctl->best_pose[lane] = pv;
      if (s_best) s_best[lane] = pv;
    }
    for (int w = lane; w < wn; w += 64) {
      const unsigned long long m = a.masks[(size_t)best_idx * a.words + w];
      a.best_mask[w] = m;
      if (lds_mask) s_mask[w] = m;
    }
  } else if (!more) {                             // (it was found by an earlier batch of this step: from the control block)
    if (s_best && lane < 12) s_best[lane] = ctl->best_pose[lane];
    if (lds_mask)
      for (int w = lane; w < wn; w += 64) s_mask[w] = a.best_mask[w];
  }
  if (s_mask_ok && lane == 0) *s_mask_ok = lds_mask ? 1 : 0;
  if (lane == 0) {
    if (more) consumed = hyp;
    ctl->n_iterations = n_it;
    ctl->outlier_ratio = orat;
    ctl->raw_pos += 7ull * (unsigned long long)consumed;
    ctl->best_idx = best_idx >= 0 ? cont * hyp + best_idx : best_idx_before;
    ctl->best_count = best;
    ctl->consumed = consumed_before + consumed;
    ctl->hyp_valid = hyp_valid;
    ctl->n_done = n;
    ctl->cont = more ? cont + 1 : 0;
    if (more) ctl->fault = VO_FAULT_CONTINUE;     // (n_p3p stays: the next batch draws from the same population)
  }
}

// main.py:261-268 for feature i of the new frame, given the new pose both ways (Tcw world -> camera, Twc its
// inverse):  outliers[triangulate_inliers] = ~inliers; set_pose_for_new_tracks (features.py:224-237);
// reset_outliers (state.py:162-172); compute_candidates (state.py:135-160, 174-219: bearing angle between the rays
// through the track's first and last keypoint >= threshold, among state == 1).  Returns the candidate flag; st_out: the
// feature's state code as this leaves it.
__device__ __forceinline__ int candidate_feature(const vo_feat& B, int i, int n_tri,
                                                 const unsigned long long* __restrict__ best_mask, const vo_cam& cam,
                                                 const double* Twc, double bearing_thr, int* st_out = nullptr) {
  int st = B.state[i];
  const int st0 = st;
  const double u = B.kp64[2 * i], v = B.kp64[2 * i + 1];
  bool reset = st == 0;
  if (i < n_tri && !((best_mask[i >> 6] >> (i & 63)) & 1ull)) {   // P3P outlier (main.py:261-262)
    st = 0;
    reset = true;
  }
  int cand = 0;
  if (reset) {
    if (st0 != 0) {
      B.track[2 * i] = u;
      B.track[2 * i + 1] = v;
      B.state[i] = 0;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) B.pose[(size_t)k * B.pitch + i] = Twc[k];
  } else if (st == 1) {
    double P[11];           // rotation part of the track's start pose (rows of the 3x4)
#pragma unroll
    for (int k = 0; k < 11; ++k) P[k] = B.pose[(size_t)k * B.pitch + i];
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
    cand = ang >= bearing_thr ? 1 : 0;                      // (NaN compares false, as in NumPy)
  }
  B.cand[i] = (uint8_t)cand;
  if (st_out) *st_out = reset ? 0 : st;
  return cand;
}

// update_from_matches + update_with_world_pose (state.py:17-50) on the control block: what was current becomes
// previous, the new pose current.  Call with tid = 0..11 (one matrix entry each) from ONE workgroup.
__device__ __forceinline__ void commit_pose(vo_seq_ctl* __restrict__ ctl, int tid, double ncw, double nwc) {
  const double ocw = ctl->T_cw[tid], owc = ctl->T_wc[tid];
  ctl->T_cw_prev[tid] = ocw;
  ctl->T_wc_prev[tid] = owc;
  ctl->T_cw[tid] = ncw;
  ctl->T_wc[tid] = nwc;
}

// ---- result records in mapped host memory ----
// The host polls a sequence word and then copies the record.  Stores to host memory are not seen in the order they were
// made, fences notwithstanding (the record's 64-byte lines travel as separate writes: a record whose last line had
// arrived and whose first lines had not was seen about once in 10^4 steps, tests/pipeline_fuzz.py), so a record proves
// itself: all of it is written every time (staged in LDS first), seq_head = the step's number XOR all other dwords,
// seq_tail = the number + a position-weighted sum of them (record_mix): 64 bits of check.  The host takes a copy whose
// sums fit (pipeline.hip:wait_record, vo_record_check) and copies again otherwise.
constexpr int REC_DW = (int)(sizeof(vo_step_result) / 4);   // seq_head and seq_tail are the last two dwords
static_assert(sizeof(vo_step_result) % 4 == 0 && REC_DW <= 128 + 2, "record: two dwords per lane of one wave");
static_assert(offsetof(vo_step_result, seq_head) == sizeof(vo_step_result) - 8 &&
              offsetof(vo_step_result, seq_tail) == sizeof(vo_step_result) - 4, "record: the closing words come last");

// The record's second check word: the step's number plus the sum over the data dwords of (dword + golden * (k + 1)) *
// (2k + 1), mod 2^32 -- position-dependent, so that lines of two records spliced together would have to collide in this
// sum AND in the XOR: one in 2^64.
__host__ __device__ inline unsigned record_mix(unsigned v, int k) {
  return (v + 0x9e3779b9u * (unsigned)(k + 1)) * (unsigned)(2 * k + 1);
}

// The record of a step that raised a fault: what the host needs to redo it.  One work item.
__device__ __forceinline__ void write_fault_record(const vo_seq_ctl* __restrict__ ctl, int fault, vo_step_result* __restrict__ res,
                                                   unsigned* __restrict__ seq_word, unsigned seq) {
  // (no local copy of the record: it would live in scratch memory, and a kernel that uses scratch pays for it at every
  //  launch -- this is the pose kernel of the dependent chain)
  const unsigned long long rp = ctl->raw_pos;
  const unsigned f_fault = (unsigned)fault, f_in = (unsigned)ctl->n_in, f_n2 = (unsigned)ctl->n2, f_tri = (unsigned)ctl->n_tri;
  unsigned x = 0u, y = 0u;
  unsigned* dst = reinterpret_cast<unsigned*>(res);
  for (int k = 0; k < REC_DW - 2; ++k) {
    unsigned v = 0u;
    if (k == (int)(offsetof(vo_step_result, fault) / 4)) v = f_fault;
    if (k == (int)(offsetof(vo_step_result, n_features_in) / 4)) v = f_in;
    if (k == (int)(offsetof(vo_step_result, n_tracked) / 4)) v = f_n2;
    if (k == (int)(offsetof(vo_step_result, n_triangulated) / 4)) v = f_tri;
    if (k == (int)(offsetof(vo_step_result, raw_pos) / 4)) v = (unsigned)rp;
    if (k == (int)(offsetof(vo_step_result, raw_pos) / 4) + 1) v = (unsigned)(rp >> 32);
    x ^= v;
    y += record_mix(v, k);
    dst[k] = v;
  }
  dst[REC_DW - 2] = seq ^ x;
  __threadfence_system();
  dst[REC_DW - 1] = seq + y;
  __threadfence_system();
  *seq_word = seq;
}

// The record of a finished step, by the workgroup that closes it (every count is final, the workgroup is in step).
// ts4: the device clock at the start of the landmark stage.
__device__ __forceinline__ void write_step_record(vo_seq_ctl* __restrict__ ctl, int tid, int use_refined, int n2, int n_cand,
                                                  int n_dropped, int n_land, unsigned long long ts4,
                                                  vo_step_result* __restrict__ res, unsigned* __restrict__ seq_word,
                                                  unsigned seq, bool fences = true) {
  __shared__ __align__(8) unsigned s_rec[REC_DW];
  vo_step_result* r = reinterpret_cast<vo_step_result*>(s_rec);
  if (tid < REC_DW) s_rec[tid] = 0u;
  __syncthreads();
  if (tid < 9) {
    r->R[tid] = ctl->best_pose[tid];
    r->R_refined[tid] = use_refined > 0 ? ctl->refined[tid] : ctl->best_pose[tid];
  }
  if (tid < 3) {
    r->t[tid] = ctl->best_pose[9 + tid];
    r->t_refined[tid] = use_refined > 0 ? ctl->refined[9 + tid] : ctl->best_pose[9 + tid];
  }
  if (tid >= 32 && tid < 44) r->T_wc[tid - 32] = ctl->T_wc[tid - 32];
  if (tid == 63) {
    r->n_tracked = n2;
    r->n_inliers = ctl->best_count;
    r->best_index = ctl->best_idx;
    r->hyp_valid = ctl->hyp_valid;
    r->ransac_iterations = ctl->n_done;
    r->draws_consumed = ctl->consumed;
    r->refine_iterations = use_refined > 0 ? (int)ctl->refined[12] : -1;
    r->refine_cost = use_refined > 0 ? ctl->refined[13] : 0.0;
    r->n_features_in = ctl->n_in;
    r->redetected = ctl->redetected;
    r->detector_ran = ctl->det_ran;
    r->n_triangulated = ctl->n_tri;
    r->n_candidates = n_cand;
    r->n_dropped = n_dropped;
    r->n_landmarks = n_land;
    r->raw_pos = ctl->raw_pos;
    r->ts[0] = ctl->ts[6];
    for (int k = 1; k < 4; ++k) r->ts[k] = ctl->ts[k];
    r->ts[4] = ts4;
    r->ts[6] = ctl->ts[5];     // inside the pose kernel: the RANSAC replay is done ...
    r->ts[7] = ctl->ts[7];     // ... the refinement is done
    r->ts[5] = wall_clock64();
  }
  __syncthreads();
  if (tid < 64) {              // one wave: two dwords per lane, the sum by a butterfly, then the two closing words
    unsigned* dst = reinterpret_cast<unsigned*>(res);
    const bool second = tid + 64 < REC_DW - 2;
    const unsigned v0 = s_rec[tid], v1 = second ? s_rec[tid + 64] : 0u;
    unsigned x = v0 ^ v1;
    unsigned y = record_mix(v0, tid) + (second ? record_mix(v1, tid + 64) : 0u);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      x ^= __shfl_xor(x, off);
      y += __shfl_xor(y, off);
    }
    dst[tid] = v0;
    if (second) dst[tid + 64] = v1;
    if (tid == 0) dst[REC_DW - 2] = seq ^ x;
    // fences = false: no system-scope fence between the lines, the closing words and the sequence word.  The host accepts a
    // copy only when both check words fit (the lines arrive in no order WITH the fences too, see above), so the fences add
    // nothing to what it can rely on -- and each is a write-back of the L2 plus a round trip to host memory at the end of
    // the step's dependent chain.
    if (fences) __threadfence_system();
    if (tid == 0) {
      dst[REC_DW - 1] = seq + y;
      if (fences) __threadfence_system();
      *seq_word = seq;
    }
  }
}

}  // namespace vo_state_dev
