// Device-resident frame state shared by state.hip (kernels) and pipeline.hip (orchestration).
//
// One sequence keeps, in HBM, what the reference keeps in Python objects between frames:
//   Features  (src/vo/primitives/features.py:4-54): keypoints, state codes, landmarks, track starts,
//             track start poses, candidate mask -- as structure-of-arrays with a fixed capacity;
//   State     (src/vo/primitives/state.py:9-15): current / previous pose;
//   RANSAC    (src/vo/algorithms/ransac.py:42-56): the fields that persist on the estimator object
//             between find_best_model calls (n_iterations, outlier_ratio) and the position in the
//             generator's output stream;
//   KLTTracker._num_features (src/vo/features/klt.py:113).
#pragma once
#include "vo_internal.h"

struct vo_feat {
  float* kp;         // cap x 2, float32 as cv2.calcOpticalFlowPyrLK returns them (klt.py:233-241)
  double* kp64;      // cap x 2, the same values as float64: what P3P / DLT / the bearing test read
  uint8_t* state;    // cap: 0 unmatched, 1 matched, 2 triangulated (features.py:41-43)
  uint8_t* cand;     // cap: candidate_mask (features.py:54)
  double* land;      // cap x 3, NaN = unknown
  double* track;     // cap x 2, keypoint at which the track started
  double* pose;      // 12 x pitch (component-major: entry k of feature i at pose[k * pitch + i]): rows 0..2 of the
                     // 4x4 camera-to-world pose at the track's start (NaN = none)
  int pitch;         // = capacity
};

// Several sequences per launch: every array of a vo_feat holds S consecutive per-sequence blocks of `pitch`
// features; block q of each array:
__host__ __device__ inline vo_feat vo_feat_seq(vo_feat F, size_t q) {
  const size_t n = q * (size_t)F.pitch;
  F.kp += 2 * n;
  F.kp64 += 2 * n;
  F.state += n;
  F.cand += n;
  F.land += 3 * n;
  F.track += 2 * n;
  F.pose += 12 * n;
  return F;
}

enum {
  VO_FAULT_FEW_LANDMARKS = 1,   // fewer than 8 triangulated tracks: the device-side sampler does not apply
  VO_FAULT_RISKY_DRAW = 2,      // a bounded draw inside the consumed prefix could have been rejected by NumPy
  VO_FAULT_UNFINISHED = 4,      // the sequential rule is not done after `hyp` samples
  VO_FAULT_CAPACITY = 8,        // appending the detector's keypoints would exceed the feature capacity
  VO_FAULT_FORCED = 16,         // test hook (vo_pipeline_config.debug_fault_every)
  VO_FAULT_NO_DETECTION = 32,   // the tracks fell below the re-detect limit on a frame whose detection was skipped
  VO_FAULT_GATE = 128,          // a device-side gate was not opened within two seconds (a kernel it waits for never ran)
  VO_FAULT_CONTINUE = 64        // not an error: the sequential rule is not done after this launch's `hyp` samples; the loop's
                                // state is in the control block and the host launches the next batch (hypotheses + pose kernel)
};

struct vo_seq_ctl {
  // ---- Features / tracker bookkeeping ----
  int32_t n;               // features of the current frame
  int32_t num_features;    // KLTTracker._num_features
  int32_t n_in;            // features handed to the tracker (after a possible re-detect)
  int32_t redetected;
  int32_t det_ran;         // the detector ran on this step's `prev` frame (its keypoints exist whether or not they were needed)
  int32_t n2, n_tri, n_mat, n_new;   // new frame: total, and the sizes of its first three groups
  int32_t n_p3p;           // population the hypothesis kernels see: n_tri, or 0 when the step must not run
  int32_t fault;           // sticky: every later kernel of this and the following steps leaves the state alone
  int32_t step;            // steps completed
  int32_t solve_flag;      // raised by the solve kernel (population < 8)
  int32_t few;             // VO_FAULT_FEW_LANDMARKS found by THIS step's regroup; the pose kernel ORs it into `fault`.  (Not
                           // written to `fault` by the regroup itself: its other workgroups read that word on entry, and one
                           // dispatched after block 0 had retired would skip its features.)
  // Device-side gates between the tracker's stream and the main stream (one or two sequences; pipeline.hip): the regroup
  // of flight j publishes gate_regroup = j + 1 when all its workgroups have written, the tracker of flight j publishes
  // gate_klt = j + 1 likewise; the consumers poll these words instead of waiting for a stream event (17-19 us each way).
  uint32_t gate_regroup, gate_regroup_cnt, gate_klt, gate_klt_cnt;
  int32_t cont;            // batches of `hyp` samples this step's RANSAC loop has already walked (VO_FAULT_CONTINUE); 0 = none
  // ---- RANSAC: persists across frames like the reference's estimator object ----
  int64_t n_iterations;
  double outlier_ratio;
  uint64_t raw_pos;        // absolute index of the next unread 32-bit generator output
  int64_t n_iterations0;   // n_iterations / outlier_ratio as the step found them (a step that continues over several batches
  double outlier_ratio0;   // and then meets a draw NumPy might have rejected is redone from its start by the host path)
  // ---- this step ----
  int32_t best_idx, best_count, consumed, hyp_valid;
  int64_t n_done;
  int32_t n_cand, n_dropped, n_land, done;   // counters the bookkeeping kernels add to (zeroed by the replay kernel)
  int32_t n_pend, pad_pend;                  // state_walk_landmarks_kernel: features whose cheirality verdict waits for the frame's
                                             // candidate count (zeroed by the pose kernel)
  double best_pose[12];    // R (9, row-major) then t (3) of the accepted hypothesis
  double refined[16];      // refine_pose_kernel's output: R, t, accepted steps, cost (+ tag)
  // 3x4 row-major, world->camera and camera->world: State.curr_pose / State.prev_pose (state.py:9-15), and a
  // pose handed in by the host (vo_pipeline_bookkeeping)
  double T_cw[12], T_wc[12], T_cw_prev[12], T_wc_prev[12], T_in_cw[12], T_in_wc[12];
  // wall_clock64() (100 MHz) when the first work item of each kernel of the chain started: tracker, regroup,
  // hypotheses, pose, landmarks, and when the landmarks kernel's last workgroup wrote the record
  unsigned long long ts[8];
};

struct vo_cam {
  double K[9], Kinv[9];
};

// The re-detect branch (klt.py:207-230) without a copy: when fewer than frac * _num_features features are left,
// the tracker and the regroup kernel treat the detector's keypoints of the old frame as features n .. n+n_det-1.
struct vo_append {
  const double* det_kp;     // n_det x 2 (device)
  size_t det_stride;        // doubles between the sequences' detector lists (several sequences per launch)
  int n_det;
  double frac;
  int pose_mode;            // vo_pipeline_config.redetect_start_pose
  int debug_fault_every;
  const int* det_go;        // per sequence: 1 = the detector ran on the frame det_kp belongs to (NULL: it always does)
  uint32_t gate_klt_want;   // != 0: wait until ctl->gate_klt reaches it (the tracker of this flight is done) ...
  uint32_t gate_regroup_set; // ... and publish ctl->gate_regroup = this when every workgroup has written
  int gate_mode;            // 1 fences, 2 agent-scope accesses to the handed-over arrays (vo_internal.h, vo_gate_wait)
};

struct vo_replay_args {
  const uint8_t* valid;                 // bit 0: the hypothesis has a pose; bit 1: one of its draws could have been rejected
  const int32_t* counts;
  const double *R, *t;                  // hyp x 9, hyp x 3
  const unsigned long long* masks;      // hyp rows of `words` 64-bit words
  int words, hyp;
  const double* table;                  // table_len + 1 doubles (vo_ransac_build_table)
  int table_len;
  long long max_it;
  unsigned long long* best_mask;        // receives the accepted hypothesis' mask row
};

// One launch for the middle of the dependent chain (refine.hip, frame_pose_kernel): the sequential RANSAC rule
// over the scored hypotheses, the refinement of the accepted pose over its inliers, then pose / outliers /
// bearing-angle candidates of every feature.
struct vo_pose_job {
  vo_seq_ctl* ctl;
  vo_replay_args rp;
  int do_replay;        // 0: ctl->best_pose / best_mask are given (host recovery path)
  vo_feat B;            // the new frame's features: land / kp64 are the P3P population
  vo_cam cam;
  double bearing_thr;
  int max_iter;         // Gauss-Newton steps allowed; 0 = refinement off
  // tail != 0: the same workgroup goes on with what state_landmarks_kernel does (candidate triangulation, landmark
  // insertion, cheirality check) and writes the step's result record -- one launch and one kernel boundary less on the
  // dependent chain of a step
  int tail;
  vo_step_result* res;  // mapped host memory (sequence q: res + q), may be NULL
  unsigned* seq_word;
  unsigned seq;
  int stamps;           // debug: device-clock stamps after the replay and after the refinement (record ts[6], ts[7])
  int walk = 1;         // 0: the kernel stops behind the refinement; vo_state_candidates (a workgroup per 256 features) walks
                        // the features, commits the pose and counts the candidates instead of this kernel's one workgroup
  int debug_fault_every; // test hook: every n-th step the replay raises VO_FAULT_FORCED where its loop ends (a fault from the
                        // POSE kernel, after the step's regroup has run and -- over several batches -- after the loop's
                        // state has moved: what a draw NumPy might have rejected does, about once in 10^3 steps)
};
int vo_frame_pose(vo_ctx* ctx, const vo_pose_job& job, int S = 1);   // S > 1: sequence q uses block q of every array

// ---- launches (state.hip); all asynchronous on ctx->stream ----
// klt.py:207-230 + 244-278 + matches.py:26-212: (virtual) re-detect append, keep status & err < thr, then the
// 4-group regroup of the new frame
int vo_state_regroup_klt(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const float* d_next_xy,
                         const uint8_t* d_status, const float* d_err, float err_thr, vo_append ap, int cap, int S = 1);
// matches.py:26-212 for an explicit match list (harris / sift trackers, tests)
// d_M / d_n2 (optional): the pair count and the new frame's keypoint count read on the device (M and n2_in are then the
// capacities); d_src_row (optional, cap ints): for every feature written to B, the new keypoint it is
int vo_state_regroup_pairs(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat A, vo_feat B, const int32_t* d_pairs, int M,
                           const double* d_new_kp, int n2_in, int cap, const int32_t* d_M = nullptr,
                           const int32_t* d_n2 = nullptr, int32_t* d_src_row = nullptr);
// main.py:261-268 + state.py:17-50, 135-219: pose, outliers, bearing-angle candidates
// S > 1: sequence q uses block q of every array (mask rows of `words` 64-bit words)
int vo_state_candidates(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, const uint64_t* d_best_mask, vo_cam cam,
                        double bearing_thr, int use_refined, int cap, int S = 1, int words = 0);
// main.py:279-286 + triangulation.py:38-86 + state.py:69-107: candidate triangulation, landmark insertion,
// cheirality check, step bookkeeping and the result record
// vo_state_candidates and vo_state_landmarks in ONE launch (the frame loop's form): a feature's walk, its triangulation and its
// cheirality test need nothing of another feature but the frame's candidate count being > 0.  d_pend: S x cap int32 of scratch.
int vo_state_walk_landmarks(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, const uint64_t* d_best_mask, int words, vo_cam cam,
                            double bearing_thr, int use_refined, int cap, int32_t* d_pend, vo_step_result* m_result,
                            unsigned* m_seq, unsigned seq, int S);
int vo_state_landmarks(vo_ctx* ctx, vo_seq_ctl* ctl, vo_feat B, vo_cam cam, int use_refined, int cap,
                       vo_step_result* m_result, unsigned* m_seq, unsigned seq, int S = 1);
// n_iterations for an outlier ratio through the threshold table (host copy of the device lookup; tests)
int64_t vo_ransac_table_lookup(const double* table, int table_len, int64_t max_iterations, double outlier_ratio);
void vo_ransac_build_table(double confidence, int s, int table_len, double* table);
