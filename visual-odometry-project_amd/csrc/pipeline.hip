// Device-resident per-frame loop (vo_pipeline_*): the steady state of the reference driver
// (src/main.py:248-286, KLT tracker mode) as one chain of launches per frame, with the Features /
// State / RANSAC bookkeeping living in HBM (state.hip).  See include/vo_hip.h for the stage list.
//
// Streams of one step (frame k-1 -> k):
//   main   : regroup -> hypotheses+counts -> replay+refine+candidates+landmarks+record
//   tracker: pyramid(k) -> KLT(k)           needs regroup(k-1) only: runs beside the pose estimation of step k-1
//   detect : Harris response + NMS on k     (enqueued by a worker thread; consumed by the NEXT step's re-detect.
//            The reference runs its detector only when fewer than 80 % of the tracks are left, klt.py:207-230; whether
//            that will be so for frame k is known one step too late for a launch without a host turn, so the chain is
//            launched for every frame and each sequence sits it out unless its track count is within `detect_margin`
//            of the limit.  A sequence that falls through the margin in one frame finds no keypoints: fault, host path.)
// Nothing on the main stream waits for the host: counts, the generator position, the accepted pose and
// the inlier mask are words in HBM that the next kernel reads.  The host only enqueues (at most two
// steps ahead: frame buffers rotate over three slots) and reads each step's result record from mapped
// memory.  The rare step the device cannot finish alone (a bounded draw NumPy might have rejected, fewer
// than 8 landmarks, the sequential rule not done after `hyp` samples) raises a sticky fault word: every
// later kernel leaves that sequence's state untouched, and vo_pipeline_collect redoes the step with the
// sequential host sampler (recover_step) before re-enqueueing what was behind it.
//
// Several sequences per GPU (vo_pipeline_config.sequences = S): S independent streams advance in lock
// step through the SAME launches -- every per-sequence buffer is S consecutive blocks, the sequence is
// the grid's extra dimension of every kernel.  The chain is latency-bound at one sequence
// (single-workgroup kernels, 127 us per step with the chip almost empty); S sequences cost about the same
// wall time per step until the image-wide kernels fill the chip.
#include <linux/futex.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <string>
#include <mutex>
#include <thread>

#include "vo_state.h"
#include "state_device.h"

#pragma clang fp contract(off)

struct vo_pipeline {
  vo_ctx* ctx = nullptr;
  vo_ctx* det = nullptr;             // detection stream (+ the NMS workspace of all sequences)
  vo_ctx* trk = nullptr;             // tracker stream: pyramid and KLT of step k+1 run beside the pose estimation of step k
  hipEvent_t evKlt[2] = {nullptr, nullptr}, evRegroup[2] = {nullptr, nullptr};
  vo_pipeline_config cfg;
  vo_cam cam;
  int n_levels = 1, cap = 0, words = 0, S = 1;
  size_t px = 0, pyr_bytes = 0;
  // ---- per-sequence buffers: S consecutive blocks each ----
  uint8_t* d_img = nullptr;          // [S][n_frames][px]
  uint8_t* d_pyr = nullptr;          // [S][3][pyr_bytes]      (frame count mod 3)
  double* d_kp = nullptr;            // [S][3][N * 2]          detector output per frame slot
  double* d_scores[2] = {nullptr, nullptr};   // [S][px] each, alternating between consecutive detections
  int* d_det_go = nullptr;           // [3][S]: 1 = the detector ran for that sequence on the frame in keypoint slot s
  double detect_limit = 0.0;         // detect when n < detect_limit * num_features (< 0: always)
  double detect_losses = 2.5;        // ... with n extrapolated by this many times the last step's loss
  hipEvent_t evPyr[3] = {nullptr, nullptr, nullptr}, evDet[3] = {nullptr, nullptr, nullptr};
  // frame upload: pinned staging per (sequence, frame slot), allocated on first use; evImg[idx]: slot idx is in HBM
  std::vector<uint8_t*> h_img;
  std::vector<hipEvent_t> evImg;
  // vo_pipeline_set_frame_pinned: DMA straight from the caller's pinned buffer on a stream of its own (beside the kernels,
  // not in front of the pyramid); side[idx]: slot idx was last filled that way -- the tracker's stream waits for evImg too
  hipStream_t up_stream = nullptr;
  std::vector<char> side;
  // vo_pipeline_prepare: the pyramid of frame slot prepared_idx sits in pyramid slot prepared_slot, built behind the
  // previous tracker -- the next submit whose `next` is that frame does not build it again (-1: none)
  int prepared_idx = -1, prepared_slot = -1;
  int slot = 0, det_flip = 0, prev_frame = -1;
  // Features double buffer: a step reads F[cur] (frame k-1) and writes F[1 - cur] (frame k)
  vo_feat F[2];
  void* feat_mem = nullptr;
  int cur = 0;
  vo_seq_ctl* d_ctl = nullptr;       // [S]
  float *d_next = nullptr, *d_err = nullptr;   // [S][cap * 2], [S][cap]
  uint8_t* d_status = nullptr;
  double *d_R = nullptr, *d_t = nullptr;       // [S][hyp * 9], [S][hyp * 3]
  uint8_t* d_valid = nullptr;
  int32_t *d_counts = nullptr, *d_samples = nullptr, *d_pend = nullptr;   // d_pend: [S][cap], state_walk_landmarks_kernel's scratch
  uint64_t *d_masks = nullptr, *d_best_mask = nullptr;
  double* d_table = nullptr;
  std::vector<double> table;
  int table_len = 0;
  // generator outputs: one power-of-two ring per sequence in HBM, kept filled ahead of the device by the host
  uint32_t* d_raws = nullptr;        // [S][ring_len]
  uint32_t ring_len = 0;
  uint32_t* h_stage = nullptr;
  size_t stage_cap = 0;
  std::vector<uint64_t> gen_upto, pos_known, pos_dev;   // generated up to / the estimator's position after the last closed
                                                        // step / the device's position (ahead of it while a step continues)
  std::vector<vo_pcg64> raw_gen, rng;
  hipEvent_t evRaw = nullptr;
  bool raw_pending = false, seeded = false, have_state = false, primed = false;
  // results: records in mapped host memory, [4 slots][S]
  vo_step_result *h_res = nullptr, *m_res = nullptr;
  volatile unsigned* h_seq = nullptr;
  unsigned* m_seq = nullptr;
  unsigned seq = 0;
  struct flight_t { int prev_idx, next_idx, a, b, fcur, rslot; unsigned seq; long k; unsigned sift_job; };
  flight_t flight[2];
  int n_flight = 0;
  long steps_submitted = 0;
  std::vector<unsigned> slot_seq;    // [4][S]: the number sequence q's record in result slot r will carry
  std::vector<char> seq_state;       // [S]: a state was handed over before (the RANSAC object persists, ransac.py:47-56)
  int last_fbuf = 0;
  hipEvent_t evA = nullptr, evB = nullptr;
  double* d_newkp = nullptr;         // scratch of the bookkeeping entry point
  // SIFT tracker mode (vo_pipeline_config.tracker_mode = 1; src/vo/features/tracker.py:60-61, sift.py:23-56): the frame's
  // keypoints and descriptors are made by the SIFT kernels on the tracker's stream, matched against the descriptors the
  // current Features carry (bytes, regrouped with them: matches.py:51-58, 134-141) on the matrix cores, and regrouped
  // from the explicit pair list -- nothing of it leaves HBM.  One sequence per pipeline in this mode.
  // Harris tracker mode (tracker_mode = 2; tracker.py:58-59, harris.py:50-84): the same with the detector's N keypoints
  // (every frame), their 19x19 raw patches as 384-byte rows, ratio 0.85.
  int sift_cap = 0;
  int desc_row = 128;                // bytes per descriptor row: 128 (SIFT) or 384 (361 patch bytes, padded)
  float* d_skp = nullptr;            // [3][sift_cap * 6]   keypoint rows of the frame in slot s
  uint8_t* d_sdesc = nullptr;        // [3][sift_cap * 128] its descriptors
  int32_t* d_sn = nullptr;           // [3] its keypoint count; [3]: pairs of the step being enqueued
  uint8_t* d_fdesc = nullptr;        // [2][cap * 128]      descriptors of the Features buffers F[0], F[1]
  int32_t* d_srcrow = nullptr;       // [cap]               new keypoint behind every regrouped feature
  uint8_t* d_ckpt_fdesc = nullptr;
  // vo_pipeline_checkpoint / _rewind: a copy of one Features buffer (all sequences) and of the control blocks
  char* d_ckpt_feat = nullptr;
  vo_seq_ctl* d_ckpt_ctl = nullptr;
  size_t feat_block = 0;             // bytes of one Features buffer (F[0] and F[1] are consecutive blocks of feat_mem)
  int ckpt_frame = -1;
  int32_t* d_pairs = nullptr;
  long n_recovered = 0, n_continued = 0;
  bool pose_fault_hook = true;       // debug_fault_every < 0 applies to submitted steps, not to what recover_step re-enqueues
  // Detection worker: a second host thread enqueues the detection of every step (6 launches) while the caller's
  // thread enqueues pyramid, tracker and the main-stream chain (6 launches): a dozen launches and half a dozen event
  // calls per step cost one thread 70-150 us on a loaded host, more than the GPU needs for the step.
  // Host threads: this pipeline's caller and (budget 2) the detection worker.  Neither spins for long: a wait first polls
  // for spin_us microseconds (a one-sequence step is ~120 us, the common waits are shorter), then blocks -- the worker on a
  // futex until a job is posted, the caller in 20 us sleeps between looks at the mapped record.  VO_HOST_THREADS_BUDGET=1:
  // no worker (the caller enqueues the detection itself, behind the step's chain) and no spinning at all -- for many ranks
  // on few cores (a job's CPU quota); VO_HOST_SPIN_US overrides the polling window.
  std::thread worker;
  std::atomic<unsigned> job_posted{0}, job_done{0};
  std::atomic<int> worker_asleep{0};
  std::atomic<bool> quit{false};
  int threads_budget = 2;
  double spin_s = 150e-6;
  // Device-side gates instead of stream events between the tracker's stream and the main stream (vo_seq_ctl): used for
  // the launches of vo_pipeline_submit when the side streams keep off some compute units (one or two sequences), so
  // that a polling kernel can never keep the kernel it waits for from running.  After anything was enqueued again for
  // one sequence (host path, a continuing RANSAC loop, a rewind) the next submit also waits for the events.
  std::string det_key, trk_key;      // what the side contexts' streams were created with (side_pool)
  bool gates = false, gate_resync = false;
  int gate_mode = 1;                 // vo_internal.h, vo_gate_wait
  bool ext_events = true;            // see enqueue_tracker
  int sift_chain_pending = 0;        // SIFT mode: flights whose main-stream chain is not enqueued yet (their SIFT launches are
                                     // being made by the worker; the chain follows at the next submit or at collect)
  flight_t jobs[4];
  int worker_rc = 0;
  char worker_err[256] = {0};
  double dbg_part[4] = {0, 0, 0, 0};   // VO_DEBUG_TIMING: submit split into worker wait / tracker / raws / chain
  double dbg_submit = 0, dbg_wait = 0;
  long dbg_steps = 0;

  // block q of the per-sequence arrays
  uint8_t* img(int q, int idx) const { return d_img + ((size_t)q * cfg.n_frames + idx) * px; }
  size_t img_stride() const { return (size_t)cfg.n_frames * px; }
  uint8_t* pyr(int q, int s) const { return d_pyr + ((size_t)q * 3 + s) * pyr_bytes; }
  size_t pyr_stride() const { return 3 * pyr_bytes; }
  double* kp(int q, int s) const { return d_kp + ((size_t)q * 3 + s) * cfg.n_keypoints * 2; }
  size_t det_stride() const { return (size_t)3 * cfg.n_keypoints * 2; }
  vo_step_result* res_h(int rslot, int q) const { return h_res + (size_t)rslot * S + q; }
  volatile unsigned* seq_h(int rslot, int q) const { return h_seq + (size_t)rslot * S + q; }
};

namespace {

struct pose17 {
  double v[17];
};

// record = [T_cw 4x4 row-major | n | landmarks cap x 3]: what one rank contributes to the shared map
__global__ __launch_bounds__(256) void export_state_kernel(pose17 head, const double* __restrict__ land, int n, int cap,
                                                           double* __restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 17) rec[i] = head.v[i];
  const int m = min(n, cap) * 3;
  if (i < m) rec[17 + i] = land[i];
}

// Does sequence q need the detector on the frame being submitted?  The count that decides is known one step later;
// what is known now is the count of the frame before (or already this frame's, when the step's regroup has run) and
// how many tracks the last step lost: the detector runs when the count, extrapolated by `losses` such losses, is below
// (redetect_fraction + detect_margin) * num_features.  (Round 2: four losses and a margin of 0.02 -- the detector then ran
// on 26 % of the forward stream's frames for the 4 % that re-detect; 2.5 and 0.01: 15-18 %, still no frame caught without
// its keypoints in ~4000 sequence-steps; 2 and 0.005: 12-14 % and one such frame.)  (The fields are read while a regroup may be writing them: any
// mix of old and new values is a usable guess, and a wrong guess is caught by the step that needs the keypoints.)
__global__ __launch_bounds__(64) void detect_decide_kernel(const vo_seq_ctl* __restrict__ ctl, int S, double limit, int n_det,
                                                           int force, int* __restrict__ go, double losses) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= S) return;
  const int n2 = ctl[q].n2;
  const int lost = max(ctl[q].n_in - (ctl[q].redetected ? n_det : 0) - n2, 0);
  go[q] = (force || limit < 0.0 || (limit > 0.0 && (double)n2 - losses * (double)lost < (double)ctl[q].num_features * limit)) ? 1 : 0;
}

// vo_pipeline_rewind: the control block as it was at the checkpoint, except what lives on the reference's estimator
// object (RANSAC.n_iterations / outlier_ratio, ransac.py:47-56) and the generator position, which go on
__global__ __launch_bounds__(64) void ctl_rewind_kernel(vo_seq_ctl* __restrict__ ctl, const vo_seq_ctl* __restrict__ saved, int S) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= S) return;
  vo_seq_ctl c = saved[q];
  c.n_iterations = ctl[q].n_iterations;
  c.outlier_ratio = ctl[q].outlier_ratio;
  c.raw_pos = ctl[q].raw_pos;
  c.step = ctl[q].step;
  c.gate_regroup = ctl[q].gate_regroup;
  c.gate_regroup_cnt = ctl[q].gate_regroup_cnt;
  c.gate_klt = ctl[q].gate_klt;
  c.gate_klt_cnt = ctl[q].gate_klt_cnt;
  ctl[q] = c;
}

// a step whose RANSAC loop wants another batch of hypotheses (VO_FAULT_CONTINUE) goes on: the fault word is cleared and the
// population is what the step's regroup counted (a later step's regroup, enqueued behind the open step, has zeroed n_p3p)
__global__ void ctl_resume_kernel(vo_seq_ctl* __restrict__ ctl) {
  ctl->fault = 0;
  ctl->n_p3p = ctl->n_tri;
}

// SIFT tracker mode: the new frame's keypoint rows (x, y, size, angle, response, octave; float) as the float64 pairs the
// regroup takes (sift.py:18 keeps kp.pt only)
__global__ __launch_bounds__(256) void sift_kp_f64_kernel(const float* __restrict__ rows, const int* __restrict__ n, int cap,
                                                          double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(*n, cap)) return;
  out[2 * i] = (double)rows[6 * i];
  out[2 * i + 1] = (double)rows[6 * i + 1];
}

// ... and the descriptors of the regrouped frame: row dst of the new Features = the new keypoint src_row[dst]'s
__global__ __launch_bounds__(256) void desc_gather_kernel(const uint8_t* __restrict__ src, const int* __restrict__ src_row,
                                                          const vo_seq_ctl* __restrict__ ctl, int cap, uint8_t* __restrict__ dst,
                                                          int row_words) {
  if (ctl->fault) return;
  const int w = blockIdx.x * 256 + threadIdx.x;        // one 4-byte word of one row
  const int row = w / row_words, k = w - row * row_words;
  if (row >= min(ctl->n2, cap)) return;
  reinterpret_cast<unsigned*>(dst)[(size_t)row * row_words + k] =
      reinterpret_cast<const unsigned*>(src)[(size_t)src_row[row] * row_words + k];
}

template <typename T>
int dev_alloc(vo_ctx* ctx, T** p, size_t count) {
  hipError_t e = hipMalloc((void**)p, count * sizeof(T) ? count * sizeof(T) : 256);
  if (e != hipSuccess) return vo_set_error(ctx, VO_ENOMEM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
  return VO_OK;
}

template <typename T>
int pin_alloc(vo_ctx* ctx, T** p, size_t count) {
  hipError_t e = hipHostMalloc((void**)p, count * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return vo_set_error(ctx, VO_ENOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
  return VO_OK;
}

// Blocking copies on the pipeline's own main stream: hipMemcpy would go through the null stream, one more stream
// competing for the four hardware queues the pipeline's streams are spread over.
hipError_t mcpy(hipStream_t st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

hipError_t mset(hipStream_t st, void* dst, int v, size_t bytes) {
  hipError_t e = hipMemsetAsync(dst, v, bytes, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

long futex_wait(std::atomic<unsigned>* a, unsigned expect) {
  return syscall(SYS_futex, reinterpret_cast<unsigned*>(a), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}

long futex_wake(std::atomic<unsigned>* a) {
  return syscall(SYS_futex, reinterpret_cast<unsigned*>(a), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
}

// a short sleep between two looks at something another agent writes (the kernel's default timer slack would round
// 20 us up to 70: one microsecond of slack for this thread, set once)
void nap(long ns) {
  static thread_local bool slack_set = false;
  if (!slack_set) {
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    slack_set = true;
  }
  timespec ts{0, ns};
  nanosleep(&ts, nullptr);
}

// polls `done` for at most spin_s seconds, then between naps
template <typename F>
void wait_until(double spin_s, F done) {
  const double t_end = now_s() + spin_s;
  for (unsigned it = 0;; ++it) {
    if (done()) return;
    if ((it & 15) != 15 || now_s() < t_end) __builtin_ia32_pause();
    else nap(5000);
  }
}

uint32_t next_pow2(uint64_t v) {
  uint32_t r = 1;
  while (r < v) r <<= 1;
  return r;
}

void expand_pose(const double* p12, double* p16) {
  memcpy(p16, p12, 96);
  if (std::isnan(p12[0])) {
    for (int k = 12; k < 16; ++k) p16[k] = NAN;     // the reference's NaN poses are NaN in all 16 entries
  } else {
    p16[12] = p16[13] = p16[14] = 0.0;
    p16[15] = 1.0;
  }
}

// one Features buffer for S sequences of `cap` features: every array S * cap entries
vo_feat carve(char*& q, int cap, int S) {
  vo_feat f;
  auto take = [&](size_t bytes) {
    void* r = q;
    q += (bytes * S + 255) & ~size_t(255);
    return r;
  };
  f.kp = (float*)take((size_t)cap * 8);
  f.kp64 = (double*)take((size_t)cap * 16);
  f.state = (uint8_t*)take((size_t)cap);
  f.cand = (uint8_t*)take((size_t)cap);
  f.land = (double*)take((size_t)cap * 24);
  f.track = (double*)take((size_t)cap * 16);
  f.pose = (double*)take((size_t)cap * 96);
  f.pitch = cap;
  return f;
}

size_t feat_bytes(int cap, int S) {
  char* q = nullptr;
  carve(q, cap, S);
  return (size_t)(q - (char*)nullptr);
}

void sync_prof(vo_pipeline* p) {
  for (vo_ctx* q : {p->det, p->trk}) {
    q->prof_on = p->ctx->prof_on;
    q->prof_kernel = p->ctx->prof_kernel;
    q->prof_every = p->ctx->prof_every;
  }
}

}  // namespace

// ---- side contexts (the tracker's and the detector's stream + workspace) are kept, not destroyed ----
// Destroying a pipeline's side streams after every pipeline and creating the next one's hung inside the runtime about once
// in 400 create / destroy cycles (tools/dev/soak_fault_timing.py with VO_DEBUG_STAGES=1: the process sat in vo_destroy of a
// side context -- hipStreamDestroy / hipFree -- with every stream idle).  A closed pipeline's side contexts go to a pool
// keyed by what their streams were created with (compute-unit mask, priority); the next pipeline with the same keys takes
// them -- streams, workspace and all.  Contexts of OTHER keys are destroyed when a pipeline is created (live CU-masked
// queues slow every other queue of the process), so at most one configuration's contexts stay alive.
// VO_SIDE_POOL=0: off.
struct side_entry {
  int device;
  std::string key;
  vo_ctx* c;
};
static std::mutex g_side_mu;
static std::vector<side_entry>& side_pool() {
  static std::vector<side_entry>* v = new std::vector<side_entry>();
  return *v;
}
static bool side_pool_on() {
  static const bool on = !(getenv("VO_SIDE_POOL") && getenv("VO_SIDE_POOL")[0] == '0');
  return on;
}
static bool side_pool_destroy_at_exit() {
  if (const char* e = getenv("VO_SIDE_POOL_ATEXIT")) return e[0] != '0';
  const char* pre = getenv("LD_PRELOAD");
  return (pre && strstr(pre, "rocprof")) || getenv("ROCP_TOOL_LIBRARIES") != nullptr;
}
// VO_SIDE_POOL_EVICT=0: contexts of other configurations stay alive beside the new pipeline's (a test suite that switches
// configuration from test to test and does not care about the speed of its queues: no stream is destroyed before the process ends)
static bool side_pool_evicts() {
  static const bool on = !(getenv("VO_SIDE_POOL_EVICT") && getenv("VO_SIDE_POOL_EVICT")[0] == '0');
  return on;
}
static std::string side_key() {      // of a context created NOW (vo_create reads the same variables)
  const char* cus = getenv("VO_STREAM_CUS");
  const char* pr = getenv("VO_STREAM_PRIORITY");
  return std::string("cus=") + (cus ? cus : "") + ";prio=" + (pr ? pr : "");
}
static void side_evict_except(int device, const std::string& ka, const std::string& kb) {
  std::vector<vo_ctx*> gone;
  {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto& v = side_pool();
    for (size_t i = 0; i < v.size();) {
      if (device < 0 || (v[i].device == device && v[i].key != ka && v[i].key != kb)) {
        gone.push_back(v[i].c);
        v.erase(v.begin() + (long)i);
      } else {
        ++i;
      }
    }
  }
  for (vo_ctx* c : gone) vo_destroy(c);
}
static int side_take(int device, const std::string& key, vo_ctx** out) {
  if (side_pool_on()) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto& v = side_pool();
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i].device == device && v[i].key == key) {
        *out = v[i].c;
        v.erase(v.begin() + (long)i);
        return VO_OK;
      }
  }
  return vo_create(device, nullptr, out);
}
static void side_give(vo_ctx* c, const std::string& key) {
  if (!c) return;
  if (side_pool_on() && c->own_stream) {
    c->prof_on = false;
    c->prof_kernel = -1;
    c->prof_every = 1;
    c->next_stop = nullptr;
    c->nms_kp_f32 = nullptr;
    c->err[0] = 0;
    // Under rocprofv3 what is still kept when the process ends is destroyed while the runtime is alive (streams left to the
    // runtime's own teardown crashed it there); registered on first use, i.e. after the runtime's own exit handlers.  Without
    // the profiler they are left alone: a process about to end gains nothing from a call that stalls once in a few hundred.
    // VO_SIDE_POOL_ATEXIT=1 / 0 overrides.
    static const bool at_exit = (std::atexit([] {
                                   if (side_pool_destroy_at_exit()) side_evict_except(-1, "", "");
                                 }),
                                 true);
    (void)at_exit;
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto& v = side_pool();
    size_t same = 0;
    for (const side_entry& e : v) same += e.device == c->device && e.key == key ? 1 : 0;
    if (same < 4) {
      v.push_back({c->device, key, c});
      return;
    }
  }
  vo_destroy(c);
}

// VO_DEBUG_STAGES=1: a line on stderr at every stage of create / destroy (which runtime call a stall sits in)
static void dbg_stage(const char* what) {
  static const bool on = getenv("VO_DEBUG_STAGES") != nullptr;
  if (on) {
    fprintf(stderr, "[stage] %s\n", what);
    fflush(stderr);
  }
}

static void worker_main(vo_pipeline* p);

static int enqueue_pyramid(vo_pipeline* p, int frame, int s);

extern "C" {

int vo_klt_num_levels(int H, int W, int win, int max_level);
size_t vo_pyramid_bytes(int H, int W, int n_levels);

void vo_pipeline_destroy(vo_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  dbg_stage("destroy: enter");
  if (p->worker.joinable()) {          // the worker first: it enqueues on the detection stream
    p->quit.store(true, std::memory_order_seq_cst);
    futex_wake(&p->job_posted);
    p->worker.join();
  }
  dbg_stage("destroy: worker joined");
  // every stream next: nothing may still read what is freed below
  (void)hipStreamSynchronize(p->ctx->stream);
  for (vo_ctx* q : {p->det, p->trk})
    if (q) (void)hipStreamSynchronize(q->stream);
  dbg_stage("destroy: streams idle");
  void* dev[] = {p->d_det_go, p->d_img, p->d_pyr, p->d_kp, p->d_scores[0], p->d_scores[1], p->feat_mem, p->d_ctl, p->d_next, p->d_err,
                 p->d_status, p->d_R, p->d_t, p->d_valid, p->d_counts, p->d_samples, p->d_masks, p->d_best_mask, p->d_table,
                 p->d_raws, p->d_newkp, p->d_pairs, p->d_ckpt_feat, p->d_ckpt_ctl, p->d_skp, p->d_sdesc, p->d_sn, p->d_fdesc,
                 p->d_srcrow, p->d_ckpt_fdesc, p->d_pend};
  for (void* q : dev)
    if (q) (void)hipFree(q);
  void* pin[] = {p->h_stage, p->h_res, (void*)p->h_seq};
  for (void* q : pin)
    if (q) (void)hipHostFree(q);
  for (uint8_t* q : p->h_img)
    if (q) (void)hipHostFree(q);
  dbg_stage("destroy: memory freed");
  if (p->up_stream) (void)hipStreamDestroy(p->up_stream);
  for (hipEvent_t e : p->evImg)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {p->evPyr[0], p->evPyr[1], p->evPyr[2], p->evDet[0], p->evDet[1], p->evDet[2], p->evRaw, p->evA, p->evB,
                       p->evKlt[0], p->evKlt[1], p->evRegroup[0], p->evRegroup[1]})
    if (e) (void)hipEventDestroy(e);
  dbg_stage("destroy: events destroyed");
  if (p->det) (p->det->own_stream ? side_give(p->det, p->det_key) : vo_destroy(p->det));
  if (p->trk) (p->trk->own_stream ? side_give(p->trk, p->trk_key) : vo_destroy(p->trk));
  dbg_stage("destroy: side contexts handed back");
  if (getenv("VO_DEBUG_TIMING") && p->dbg_steps > 0)
    fprintf(stderr, "[vo_pipeline] %ld steps x %d sequence(s): host %.1f us enqueueing (worker wait %.1f, tracker %.1f, raws %.1f, "
            "chain %.1f), %.1f us waiting per step; %ld finished through the host path, %ld further batches of hypotheses\n",
            p->dbg_steps, p->S, 1e6 * p->dbg_submit / p->dbg_steps, 1e6 * p->dbg_part[0] / p->dbg_steps,
            1e6 * p->dbg_part[1] / p->dbg_steps, 1e6 * p->dbg_part[2] / p->dbg_steps, 1e6 * p->dbg_part[3] / p->dbg_steps,
            1e6 * p->dbg_wait / p->dbg_steps, p->n_recovered, p->n_continued);
  delete p;
}

int vo_pipeline_create(vo_ctx* ctx, const vo_pipeline_config* cfg, vo_pipeline** out) {
  dbg_stage("create: enter");
  if (!ctx || !cfg || !out) return VO_EINVAL;
  *out = nullptr;
  VO_REQUIRE(ctx, cfg->H > 0 && cfg->W > 0 && cfg->n_frames >= 2, "pipeline: bad stream shape");
  VO_REQUIRE(ctx, cfg->n_keypoints >= 8 && cfg->n_keypoints <= 16384, "pipeline: n_keypoints must be in 8..16384");
  VO_REQUIRE(ctx, cfg->hyp >= 1 && cfg->hyp <= (1 << 20), "pipeline: hyp must be in 1..2^20");
  VO_REQUIRE(ctx, cfg->K[0] != 0.0 && cfg->K[4] != 0.0, "pipeline: singular intrinsics");
  VO_REQUIRE(ctx, cfg->refine_iters >= 0 && cfg->refine_iters <= 100, "pipeline: refine_iters must be in 0..100");
  VO_REQUIRE(ctx, cfg->sequences >= 0 && cfg->sequences <= 256, "pipeline: sequences must be in 1..256");
  VO_REQUIRE(ctx, cfg->tracker_mode >= 0 && cfg->tracker_mode <= 2, "pipeline: tracker_mode must be 0 (klt), 1 (sift) or 2 (harris)");
  VO_REQUIRE(ctx, cfg->tracker_mode == 0 || cfg->sequences <= 1, "pipeline: the descriptor tracker modes run one sequence per pipeline");
  const int cap = cfg->feature_cap > 0 ? cfg->feature_cap : 2 * cfg->n_keypoints;
  VO_REQUIRE(ctx, cap >= cfg->n_keypoints && cap <= 32768, "pipeline: feature_cap must be in n_keypoints..32768");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_pipeline* p = new (std::nothrow) vo_pipeline();
  if (!p) return VO_ENOMEM;
  p->ctx = ctx;
  p->cfg = *cfg;
  p->cap = cap;
  p->words = vo_cdiv(cap, 64);
  p->S = cfg->sequences > 0 ? cfg->sequences : 1;
  p->cfg.sequences = p->S;
  const int S = p->S;
  if (p->cfg.bearing_threshold == 0.0) p->cfg.bearing_threshold = 0.0075;    // state.py:8
  if (p->cfg.redetect_fraction == 0.0) p->cfg.redetect_fraction = 0.8;       // klt.py:212
  if (p->cfg.detect_margin == 0.0) p->cfg.detect_margin = 0.01;
  if (p->cfg.detect_losses <= 0.0) p->cfg.detect_losses = 2.5;
  p->detect_losses = p->cfg.detect_losses;
  p->detect_limit = p->cfg.detect_margin < 0.0 ? -1.0 : p->cfg.redetect_fraction + p->cfg.detect_margin;
  if (p->cfg.debug_never_detect) p->detect_limit = 0.0;     // test hook: only forced detections (state hand-over, host path)
  if (const char* e = getenv("VO_DETECT_LOSSES")) p->detect_losses = atof(e);
  memcpy(p->cam.K, cfg->K, sizeof(p->cam.K));
  {
    bool given = false;
    for (double v : cfg->Kinv) given |= v != 0.0;
    if (given) {
      memcpy(p->cam.Kinv, cfg->Kinv, sizeof(p->cam.Kinv));
    } else {
      const double fx = cfg->K[0], fy = cfg->K[4], cx = cfg->K[2], cy = cfg->K[5];
      const double ki[9] = {1.0 / fx, 0.0, -cx / fx, 0.0, 1.0 / fy, -cy / fy, 0.0, 0.0, 1.0};
      memcpy(p->cam.Kinv, ki, sizeof(ki));
    }
  }
  int rc = VO_OK;
  // Three streams -- main (the caller's), tracker, detection -- plus the null stream (the caller's synchronous
  // copies, torch): the runtime spreads streams over four hardware queues and kernels of one queue run in order.
  // A fifth stream shares a queue: with two detection streams the second sat on the tracker's queue and delayed
  // it every other frame (rocprofv3 trace, same queue id; 8.2k vs 5.1k frames/s run to run).  Hence the next
  // frame's pyramid on the tracker's stream, one detection stream, no hipMemcpy in here.
  // (VO_ONE_STREAM=1, measurements only: every kernel on the caller's stream, so that a kernel trace shows each
  // kernel's duration without the others running beside it)
  void* side = getenv("VO_ONE_STREAM") ? (void*)ctx->stream : nullptr;
  {
    // VO_SIDE_PRIORITY=low: tracker and detection streams at the device's least priority (measurement knob)
    const char* sp = getenv("VO_SIDE_PRIORITY");
    const char* saved = getenv("VO_STREAM_PRIORITY");
    std::string keep = saved ? saved : "";
    if (sp) setenv("VO_STREAM_PRIORITY", sp, 1);
    // Tracker and detection streams keep off the first 32 compute units when the pipeline runs one or two sequences with
    // the detector gated: the next frame's pyramid (858 workgroups) reaches the GPU beside the hypothesis kernel's 144,
    // and that kernel then takes 44 us instead of 23 -- its workgroups wait for a place behind the pyramid's -- unless
    // some compute units stay out of the side streams' reach (measured: 32 of the 256 are enough, 16 are not; step period
    // 122 -> 102 us).  With many sequences, or the detector on every frame, the side streams' kernels are the throughput and
    // the mask costs more than it gives (16 sequences: -5 %).  VO_SIDE_CUS="lo-hi" overrides, "all" switches it off.
    const char* sc = getenv("VO_SIDE_CUS");
    std::string auto_mask;
    if (!sc && p->S <= 2 && p->detect_limit >= 0.0 && !side && cfg->tracker_mode == 0) {   // (KLT mode only)
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount >= 128) {
        auto_mask = "32-" + std::to_string(prop.multiProcessorCount - 1);
        sc = auto_mask.c_str();
      }
    }
    if (sc && !strcmp(sc, "all")) sc = nullptr;
    {
      const int w = cfg->klt_win;
      const char* g = getenv("VO_GATES");
      // Off unless VO_GATES=1.  Measured: the two event waits they replace cost 17-19 us each, but every polling
      // workgroup needs an acquire at agent scope when its gate opens and a release before it arrives, and on this part
      // those are an invalidate / a write-back of the XCD's L2: with ~1000 tracker workgroups per frame the hypothesis and
      // pose kernels running beside them lose their cached population over and over (hypotheses -> pose 23 -> 40 us,
      // pose 34 -> 45, step 103 -> 133 us).  The gates would need the gated data to bypass L2 altogether.
      p->ext_events = !(getenv("VO_EXT_EVENTS") && getenv("VO_EXT_EVENTS")[0] == '0');
      p->gates = sc != nullptr && p->S <= 2 && !side && (w == 15 || w == 17 || w == 21) && g && (g[0] == '1' || g[0] == '2');
      p->gate_mode = g && g[0] == '2' ? 2 : 1;
    }
    const char* saved_c = getenv("VO_STREAM_CUS");
    std::string keep_c = saved_c ? saved_c : "";
    const char* dc = getenv("VO_DET_CUS");     // (the detection stream's own range; default: VO_SIDE_CUS)
    if (dc || sc) setenv("VO_STREAM_CUS", dc ? dc : sc, 1);
    else unsetenv("VO_STREAM_CUS");
    p->det_key = side_key();
    const std::string det_cus = getenv("VO_STREAM_CUS") ? getenv("VO_STREAM_CUS") : "";
    if (sc) setenv("VO_STREAM_CUS", sc, 1);
    else unsetenv("VO_STREAM_CUS");
    p->trk_key = side_key();
    if (!side && side_pool_evicts()) side_evict_except(ctx->device, p->det_key, p->trk_key);
    if ((side ? vo_create(ctx->device, side, &p->trk) : side_take(ctx->device, p->trk_key, &p->trk)) != VO_OK)
      rc = vo_set_error(ctx, VO_EHIP, "pipeline: cannot create the side streams");
    if (!det_cus.empty()) setenv("VO_STREAM_CUS", det_cus.c_str(), 1);
    else unsetenv("VO_STREAM_CUS");
    if ((side ? vo_create(ctx->device, side, &p->det) : side_take(ctx->device, p->det_key, &p->det)) != VO_OK)
      rc = vo_set_error(ctx, VO_EHIP, "pipeline: cannot create the side streams");
    if (sp) {
      if (saved) setenv("VO_STREAM_PRIORITY", keep.c_str(), 1);
      else unsetenv("VO_STREAM_PRIORITY");
    }
    if (saved_c) setenv("VO_STREAM_CUS", keep_c.c_str(), 1);
    else unsetenv("VO_STREAM_CUS");
  }
  dbg_stage("create: side streams made");
  const int N = cfg->n_keypoints, Hyp = cfg->hyp;
  const size_t px = (size_t)cfg->H * cfg->W, Sz = (size_t)S;
  p->px = px;
  p->n_levels = vo_klt_num_levels(cfg->H, cfg->W, cfg->klt_win, cfg->klt_max_level);
  p->pyr_bytes = vo_pyramid_bytes(cfg->H, cfg->W, p->n_levels);
#define PA(expr) do { if (rc == VO_OK) rc = (expr); } while (0)
  PA(dev_alloc(ctx, &p->d_img, Sz * cfg->n_frames * px));
  PA(dev_alloc(ctx, &p->d_pyr, Sz * 3 * p->pyr_bytes));
  PA(dev_alloc(ctx, &p->d_kp, Sz * 3 * N * 2));
  PA(dev_alloc(ctx, &p->d_scores[0], Sz * px));
  PA(dev_alloc(ctx, &p->d_scores[1], Sz * px));
  PA(dev_alloc(ctx, &p->d_det_go, 3 * Sz));
  {
    const size_t fb = feat_bytes(cap, S);
    char* mem = nullptr;
    PA(dev_alloc(ctx, &mem, 2 * fb));
    p->feat_mem = mem;
    p->feat_block = fb;
    if (mem) {
      char* q = mem;
      p->F[0] = carve(q, cap, S);
      p->F[1] = carve(q, cap, S);
    }
  }
  PA(dev_alloc(ctx, &p->d_ctl, Sz));
  PA(dev_alloc(ctx, &p->d_next, Sz * cap * 2));
  PA(dev_alloc(ctx, &p->d_err, Sz * cap));
  PA(dev_alloc(ctx, &p->d_status, Sz * cap));
  PA(dev_alloc(ctx, &p->d_R, Sz * Hyp * 9));
  PA(dev_alloc(ctx, &p->d_t, Sz * Hyp * 3));
  PA(dev_alloc(ctx, &p->d_valid, Sz * Hyp));
  PA(dev_alloc(ctx, &p->d_counts, Sz * Hyp));
  PA(dev_alloc(ctx, &p->d_samples, (size_t)Hyp * 4));
  PA(dev_alloc(ctx, &p->d_masks, Sz * Hyp * p->words));
  PA(dev_alloc(ctx, &p->d_best_mask, Sz * p->words));
  PA(dev_alloc(ctx, &p->d_pend, Sz * cap));
  PA(dev_alloc(ctx, &p->d_newkp, (size_t)cap * 2));
  PA(dev_alloc(ctx, &p->d_pairs, (size_t)cap * 2));
  if (cfg->tracker_mode != 0) {
    p->desc_row = cfg->tracker_mode == 2 ? 384 : 128;
    p->sift_cap = cfg->tracker_mode == 2 ? cfg->n_keypoints : (cfg->sift_cap > 0 ? cfg->sift_cap : cfg->n_keypoints);
    if (rc == VO_OK && (p->sift_cap > cap || p->sift_cap > 4000))
      rc = vo_set_error(ctx, VO_EINVAL, "pipeline: sift_cap %d exceeds the feature capacity %d (or 4000)", p->sift_cap, cap);
    PA(dev_alloc(ctx, &p->d_skp, (size_t)3 * p->sift_cap * 6));
    PA(dev_alloc(ctx, &p->d_sdesc, (size_t)3 * p->sift_cap * p->desc_row));
    PA(dev_alloc(ctx, &p->d_sn, 8));
    PA(dev_alloc(ctx, &p->d_fdesc, (size_t)2 * cap * p->desc_row));
    PA(dev_alloc(ctx, &p->d_srcrow, (size_t)cap));
    if (rc == VO_OK && (hipMemset(p->d_sn, 0, 32) != hipSuccess || hipMemset(p->d_fdesc, 0, (size_t)2 * cap * p->desc_row) != hipSuccess))
      rc = vo_set_error(ctx, VO_EHIP, "pipeline: hipMemset failed");
    if (rc == VO_OK && cfg->tracker_mode == 2) {          // (every frame has exactly N detector keypoints)
      const int32_t nn[3] = {cfg->n_keypoints, cfg->n_keypoints, cfg->n_keypoints};
      if (hipMemcpy(p->d_sn, nn, 12, hipMemcpyHostToDevice) != hipSuccess) rc = vo_set_error(ctx, VO_EHIP, "pipeline: hipMemcpy failed");
    }
  }
  // n_iterations as a step function of the outlier ratio (state_device.h, table_lookup): a batch of `hyp`
  // samples cannot finish a rule that needs more than `hyp` iterations, so hyp + 1 thresholds suffice
  // -- unless the loop continues over several launches (VO_FAULT_CONTINUE): then the table holds the whole budget
  //    (every bound up to max_iterations; an unbounded budget: up to 65536, beyond that the host's loop takes over)
  {
    const int64_t mi = cfg->ransac_max_iterations;
    const int64_t want = mi >= 0 ? std::min<int64_t>(mi, 65536) : 65536;
    p->table_len = (int)std::max<int64_t>(Hyp + 1, want + 1);
  }
  p->table.assign((size_t)p->table_len + 1, 0.0);
  vo_ransac_build_table(cfg->ransac_confidence, 4, p->table_len, p->table.data());
  PA(dev_alloc(ctx, &p->d_table, p->table.size()));
  const size_t need = (size_t)7 * Hyp;
  p->ring_len = next_pow2(32 * need);
  p->stage_cap = 16 * need;
  PA(dev_alloc(ctx, &p->d_raws, Sz * p->ring_len));
  PA(pin_alloc(ctx, &p->h_stage, p->stage_cap));
  PA(pin_alloc(ctx, &p->h_res, 4 * Sz));
  {
    unsigned* q = nullptr;
    PA(pin_alloc(ctx, &q, 4 * Sz + 16));
    if (q) memset(q, 0, (4 * Sz + 16) * sizeof(unsigned));
    p->h_seq = q;
  }
  dbg_stage("create: allocations made");
  if (rc == VO_OK && (hipHostGetDevicePointer((void**)&p->m_res, (void*)p->h_res, 0) != hipSuccess ||
                      hipHostGetDevicePointer((void**)&p->m_seq, (void*)p->h_seq, 0) != hipSuccess))
    rc = vo_set_error(ctx, VO_EHIP, "hipHostGetDevicePointer failed");
#undef PA
  if (rc == VO_OK) {
    hipEvent_t* evs[] = {&p->evPyr[0], &p->evPyr[1], &p->evPyr[2], &p->evDet[0], &p->evDet[1], &p->evDet[2],
                         &p->evRaw, &p->evA, &p->evB, &p->evKlt[0], &p->evKlt[1], &p->evRegroup[0], &p->evRegroup[1]};
    for (hipEvent_t* e : evs)
      if (rc == VO_OK && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess)
        rc = vo_set_error(ctx, VO_EHIP, "hipEventCreate failed");
  }
  dbg_stage("create: events made");
  if (rc == VO_OK && (mcpy(ctx->stream, p->d_table, p->table.data(), p->table.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
                      mset(ctx->stream, p->d_ctl, 0, Sz * sizeof(vo_seq_ctl)) != hipSuccess))
    rc = vo_set_error(ctx, VO_EHIP, "pipeline: initial uploads failed");
  if (rc != VO_OK) {
    vo_pipeline_destroy(p);
    return rc;
  }
  dbg_stage("create: uploads made");
  p->h_img.assign(Sz * cfg->n_frames, nullptr);
  p->evImg.assign((size_t)cfg->n_frames, nullptr);
  p->side.assign((size_t)cfg->n_frames, 0);
  for (hipEvent_t& e : p->evImg)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = vo_set_error(ctx, VO_EHIP, "hipEventCreate failed");
  if (rc != VO_OK) {
    vo_pipeline_destroy(p);
    return rc;
  }
  p->gen_upto.assign(Sz, 0);
  p->pos_known.assign(Sz, 0);
  p->pos_dev.assign(Sz, 0);
  p->raw_gen.resize(Sz);
  p->rng.resize(Sz);
  p->slot_seq.assign(4 * Sz, 0u);
  p->seq_state.assign(Sz, 0);
  for (auto& g : p->rng) memset(&g, 0, sizeof(g));
  // First use in a fixed order -- main, tracker, detection: the runtime attaches a stream to a hardware queue when it
  // first runs, and the three streams of the frame loop should end up on three different queues.
  {
    hipStream_t order[3] = {ctx->stream, p->trk->stream, p->det->stream};
    for (hipStream_t q : order) {
      (void)hipMemsetAsync(p->d_status, 0, 4, q);
      (void)hipStreamSynchronize(q);
      dbg_stage("create: a stream ran");
    }
  }
  if (const char* e = getenv("VO_HOST_THREADS_BUDGET")) p->threads_budget = atoi(e) <= 1 ? 1 : 2;
  if (p->threads_budget == 1) p->spin_s = 0.0;
  if (const char* e = getenv("VO_HOST_SPIN_US")) p->spin_s = 1e-6 * (double)std::max(0, atoi(e));
  if (p->threads_budget >= 2) p->worker = std::thread(worker_main, p);
  *out = p;
  return VO_OK;
}

// The side contexts kept from closed pipelines (side_pool above) are destroyed now.  For a process that goes on WITHOUT a
// pipeline and wants its other queues at full speed (live CU-masked queues slow every queue of the process).
void vo_pipeline_release_cached(void) { side_evict_except(-1, "", ""); }
int vo_pipeline_release_cached_at_exit(void) { return side_pool_destroy_at_exit() ? 1 : 0; }

int vo_pipeline_feature_cap(vo_pipeline* p) { return p ? p->cap : 0; }
int vo_pipeline_sequences(vo_pipeline* p) { return p ? p->S : 0; }

int64_t vo_pipeline_ransac_bound(vo_pipeline* p, double outlier_ratio) {
  if (!p) return -1;
  return vo_ransac_table_lookup(p->table.data(), p->table_len, p->cfg.ransac_max_iterations, outlier_ratio);
}

int vo_pipeline_set_frame_seq(vo_pipeline* p, int seq, int idx, const uint8_t* img) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, seq >= 0 && seq < p->S && idx >= 0 && idx < p->cfg.n_frames && img, "pipeline_set_frame: bad arguments");
  for (int k = 0; k < p->n_flight; ++k)
    VO_REQUIRE(ctx, p->flight[k].prev_idx != idx && p->flight[k].next_idx != idx,
               "pipeline_set_frame: slot %d belongs to a step in flight", idx);
  // the frame submitted last is what the next step tracks FROM (and what a skipped detection is made up from)
  VO_REQUIRE(ctx, !(p->have_state && p->primed && idx == p->prev_frame),
             "pipeline_set_frame: slot %d holds the frame the next step starts from", idx);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // Through a pinned staging buffer of this (sequence, slot), as one DMA queued on the tracker's stream -- in front of
  // the pyramid that reads the slot; the detector's stream waits for evImg.  The call does not wait for the GPU (the
  // runtime's pageable-memory path did, and drained the tracker's stream on top: 0.9 ms per frame through the Python
  // API); the caller's buffer is free on return.
  uint8_t*& stage = p->h_img[(size_t)seq * p->cfg.n_frames + idx];
  if (!stage) {
    hipError_t e = hipHostMalloc((void**)&stage, p->px, hipHostMallocDefault);
    if (e != hipSuccess) {
      stage = nullptr;
      return vo_set_error(ctx, VO_ENOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
  }
  VO_HIP_TRY(ctx, hipEventSynchronize(p->evImg[idx]));       // (the slot's previous upload has left the staging buffer)
  memcpy(stage, img, p->px);
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->img(seq, idx), stage, p->px, hipMemcpyHostToDevice, p->trk->stream));
  VO_HIP_TRY(ctx, hipEventRecord(p->evImg[idx], p->trk->stream));
  p->side[(size_t)idx] = 0;
  if (p->prepared_idx == idx) p->prepared_idx = p->prepared_slot = -1;
  return VO_OK;
}

// The same from a buffer the caller holds in pinned memory (vo_host_alloc): no staging copy, and the DMA runs on a
// stream of its own.  The buffer must stay as it is until the upload is over: vo_pipeline_frame_uploaded(idx), or the
// collect of a step that read the slot.
int vo_pipeline_set_frame_pinned(vo_pipeline* p, int seq, int idx, const uint8_t* pinned_img) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, seq >= 0 && seq < p->S && idx >= 0 && idx < p->cfg.n_frames && pinned_img, "pipeline_set_frame_pinned: bad arguments");
  for (int k = 0; k < p->n_flight; ++k)
    VO_REQUIRE(ctx, p->flight[k].prev_idx != idx && p->flight[k].next_idx != idx,
               "pipeline_set_frame_pinned: slot %d belongs to a step in flight", idx);
  VO_REQUIRE(ctx, !(p->have_state && p->primed && idx == p->prev_frame),
             "pipeline_set_frame_pinned: slot %d holds the frame the next step starts from", idx);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!p->up_stream) VO_HIP_TRY(ctx, hipStreamCreateWithFlags(&p->up_stream, hipStreamNonBlocking));
  // (a slot the tracker's stream filled last: that copy is in front of everything that read the slot; a step that read it
  //  has been collected -- the check above --, so nothing on the GPU still reads what this copy overwrites)
  if (!p->side[(size_t)idx]) VO_HIP_TRY(ctx, hipStreamWaitEvent(p->up_stream, p->evImg[idx], 0));   // (a copy vo_pipeline_set_frame queued)
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->img(seq, idx), pinned_img, p->px, hipMemcpyHostToDevice, p->up_stream));
  VO_HIP_TRY(ctx, hipEventRecord(p->evImg[idx], p->up_stream));
  p->side[(size_t)idx] = 1;
  if (p->prepared_idx == idx) p->prepared_idx = p->prepared_slot = -1;
  return VO_OK;
}

int vo_pipeline_frame_uploaded(vo_pipeline* p, int idx, int wait) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames, "pipeline_frame_uploaded: bad slot");
  if (wait) {
    VO_HIP_TRY(ctx, hipEventSynchronize(p->evImg[idx]));
    return 1;
  }
  return hipEventQuery(p->evImg[idx]) == hipSuccess ? 1 : 0;
}

// The pyramid of a frame that a coming step will track INTO, built now, behind the tracker of the step submitted last (on
// the tracker's stream): the next vo_pipeline_submit whose `next_idx` is this slot finds it ready.  Without the hint a
// step's pyramid is enqueued by its own submit -- which the host makes when it has collected the step before the previous
// one -- and the tracker, which needs nothing else that late, starts behind it: 31 us after the previous regroup instead
// of ~15.  (KLT tracker mode; a no-op in the others.  A hint that turns out wrong costs one wasted pyramid.)
int vo_pipeline_prepare(vo_pipeline* p, int idx) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames, "pipeline_prepare: bad frame slot");
  if (p->cfg.tracker_mode != 0 || !p->primed || !p->have_state) return VO_OK;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int s = (p->slot + 1) % 3;                    // the pyramid slot the next submit gives its `next` frame
  // (that slot held the `prev` pyramid of the step before the one submitted last: its tracker is earlier on this stream)
  VO_TRY(enqueue_pyramid(p, idx, s));
  p->prepared_idx = idx;
  p->prepared_slot = s;
  return VO_OK;
}

int vo_host_alloc(vo_ctx* ctx, size_t bytes, void** out) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, out && bytes > 0, "host_alloc: bad arguments");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return vo_set_error(ctx, VO_ENOMEM, "hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
  }
  return VO_OK;
}

int vo_host_free(vo_ctx* ctx, void* q) {      // (ctx may be null: a buffer can outlive the context it was made with)
  if (q && hipHostFree(q) != hipSuccess) return ctx ? vo_set_error(ctx, VO_EHIP, "hipHostFree failed") : VO_EHIP;
  return VO_OK;
}

int vo_pipeline_set_frame(vo_pipeline* p, int idx, const uint8_t* img) { return vo_pipeline_set_frame_seq(p, 0, idx, img); }

int vo_pipeline_seed(vo_pipeline* p, const vo_pcg64* rng) {
  if (!p || !rng) return VO_EINVAL;
  VO_REQUIRE(p->ctx, p->n_flight == 0, "pipeline_seed: %d submitted step(s) not collected", p->n_flight);
  for (int q = 0; q < p->S; ++q) {     // every sequence has its own estimator object: each starts from this state
    p->rng[q] = *rng;
    p->raw_gen[q] = *rng;
    // the device continues at the end of what has been generated so far; that look-ahead is dropped
    p->pos_known[q] = p->gen_upto[q];
    VO_HIP_TRY(p->ctx, mcpy(p->ctx->stream, &p->d_ctl[q].raw_pos, &p->gen_upto[q], 8, hipMemcpyHostToDevice));
  }
  p->seeded = true;
  return VO_OK;
}

int vo_pipeline_get_rng_seq(vo_pipeline* p, int seq, vo_pcg64* rng) {
  if (!p || !rng || seq < 0 || seq >= p->S) return VO_EINVAL;
  *rng = p->rng[seq];
  return VO_OK;
}

int vo_pipeline_get_rng(vo_pipeline* p, vo_pcg64* rng) { return vo_pipeline_get_rng_seq(p, 0, rng); }

}  // extern "C"

// ---- launches; (q0, Sn): sequences q0 .. q0 + Sn - 1 (all of them, or one when a step is redone) ----

// Harris + NMS of frame slot `frame` into keypoint slot `s` on the detection stream; evDet[s] when done
// (err_buf: the worker thread's private error text -- the pipeline context's buffer belongs to the caller's thread)
static int enqueue_detection(vo_pipeline* p, int frame, int s, bool force, char* err_buf = nullptr) {
  const vo_pipeline_config& c = p->cfg;
  p->det_flip ^= 1;
  vo_ctx* det = p->det;
  double* scores = p->d_scores[p->det_flip];
  det->nms_kp_f32 = nullptr;
  int* go = p->d_det_go + (size_t)s * p->S;
  if (hipStreamWaitEvent(det->stream, p->evImg[frame], 0) != hipSuccess) {   // the frame's upload (tracker's stream)
    if (err_buf) {
      snprintf(err_buf, 256, "detection: hipStreamWaitEvent failed");
      return VO_EHIP;
    }
    return vo_set_error(p->ctx, VO_EHIP, "detection: hipStreamWaitEvent failed");
  }
  hipLaunchKernelGGL(detect_decide_kernel, dim3(vo_cdiv(p->S, 64)), dim3(64), 0, det->stream, p->d_ctl, p->S, p->detect_limit,
                     c.n_keypoints, force ? 1 : 0, go, p->detect_losses);
  int rc = vo_check_launch(det, "detect_decide_kernel");
  if (rc == VO_OK)
    rc = vo_harris_response_batch_dev(det, p->img(0, frame), p->img_stride(), p->S, c.H, c.W, c.harris_patch, c.harris_kappa,
                                      scores, go);
  if (rc == VO_OK)
    rc = vo_nms_keypoints_batch_dev(det, scores, p->S, c.H, c.W, c.n_keypoints, c.nms_radius, p->kp(0, s), p->det_stride(),
                                    go);
  if (rc == VO_OK && hipEventRecord(p->evDet[s], det->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) {
    if (err_buf) {
      snprintf(err_buf, 256, "detection: %s", vo_last_error(det));
      return rc;
    }
    return vo_set_error(p->ctx, rc, "detection: %s", vo_last_error(det));
  }
  return VO_OK;
}

static int enqueue_pyramid(vo_pipeline* p, int frame, int s) {
  if (s == p->prepared_slot) p->prepared_idx = p->prepared_slot = -1;      // (whatever vo_pipeline_prepare left there goes)
  if (p->side[(size_t)frame] && hipStreamWaitEvent(p->trk->stream, p->evImg[frame], 0) != hipSuccess)   // (vo_pipeline_set_frame_pinned)
    return vo_set_error(p->ctx, VO_EHIP, "pyramid: hipStreamWaitEvent failed");
  const vo_pipeline_config& c = p->cfg;
  const int rc = vo_pyramid_build_batch_dev(p->trk, p->img(0, frame), p->img_stride(), p->S, c.H, c.W, p->n_levels,
                                            p->pyr(0, s), p->pyr_stride());
  if (rc != VO_OK) return vo_set_error(p->ctx, rc, "pyramid: %s", vo_last_error(p->trk));
  VO_HIP_TRY(p->ctx, hipEventRecord(p->evPyr[s], p->trk->stream));
  return VO_OK;
}

// keeps sequence q's ring of generator outputs filled ahead of every step that may be in flight
static int ensure_raws(vo_pipeline* p, int q) {
  vo_ctx* ctx = p->ctx;
  const uint64_t need = (uint64_t)7 * p->cfg.hyp;
  const uint64_t pos = std::max(p->pos_known[q], p->pos_dev[q]);
  if (p->gen_upto[q] >= pos + 4 * need) return VO_OK;
  const uint64_t target = pos + 16 * need;
  const size_t m = (size_t)(target - p->gen_upto[q]);        // <= stage_cap
  if (p->raw_pending) {
    VO_HIP_TRY(ctx, hipEventSynchronize(p->evRaw));          // the staging buffer's last copy (long done)
    p->raw_pending = false;
  }
  vo_rng_raw32(&p->raw_gen[q], (int)m, p->h_stage);
  uint32_t* ring = p->d_raws + (size_t)q * p->ring_len;
  const uint32_t off = (uint32_t)(p->gen_upto[q] & (p->ring_len - 1));
  const size_t first = std::min(m, (size_t)(p->ring_len - off));
  VO_HIP_TRY(ctx, hipMemcpyAsync(ring + off, p->h_stage, first * 4, hipMemcpyHostToDevice, ctx->stream));
  if (first < m)
    VO_HIP_TRY(ctx, hipMemcpyAsync(ring, p->h_stage + first, (m - first) * 4, hipMemcpyHostToDevice, ctx->stream));
  VO_HIP_TRY(ctx, hipEventRecord(p->evRaw, ctx->stream));
  p->raw_pending = true;
  p->gen_upto[q] = target;
  return VO_OK;
}

static vo_pose_job make_pose_job(vo_pipeline* p, const vo_feat& B, int do_replay, int q0) {
  const vo_pipeline_config& c = p->cfg;
  const size_t q = (size_t)q0;
  vo_pose_job j;
  j.ctl = p->d_ctl + q;
  j.rp.valid = p->d_valid + q * c.hyp;
  j.rp.counts = p->d_counts + q * c.hyp;
  j.rp.R = p->d_R + q * c.hyp * 9;
  j.rp.t = p->d_t + q * c.hyp * 3;
  j.rp.masks = (const unsigned long long*)p->d_masks + q * c.hyp * p->words;
  j.rp.words = p->words;
  j.rp.hyp = c.hyp;
  j.rp.table = p->d_table;
  j.rp.table_len = p->table_len;
  j.rp.max_it = c.ransac_max_iterations;
  j.rp.best_mask = (unsigned long long*)p->d_best_mask + q * p->words;
  j.do_replay = do_replay;
  j.B = vo_feat_seq(B, q);
  j.cam = p->cam;
  j.bearing_thr = c.bearing_threshold;
  j.max_iter = c.refine_iters;
  j.tail = 0;
  j.res = nullptr;
  j.seq_word = nullptr;
  j.seq = 0u;
  static const int stamps = getenv("VO_POSE_STAMPS") ? 1 : 0;
  j.stamps = stamps;
  j.debug_fault_every = 0;
  return j;
}

// tracker of one step, on its own stream: it needs the previous step's regroup (the features' positions) and this
// frame's pyramid, nothing of the previous step's pose estimation, which runs beside it on the main stream
static int enqueue_tracker(vo_pipeline* p, const vo_pipeline::flight_t& f, bool with_pyramid, int q0, int Sn,
                           bool gated = false) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  const vo_feat A = vo_feat_seq(p->F[f.fcur], (size_t)q0);
  hipStream_t ts = p->trk->stream;
  if (with_pyramid) VO_TRY(enqueue_pyramid(p, f.next_idx, f.b));
  if ((!gated || p->gate_resync) && f.k > 0 && hipEventQuery(p->evRegroup[(f.k - 1) & 1]) != hipSuccess)
    VO_HIP_TRY(ctx, hipStreamWaitEvent(ts, p->evRegroup[(f.k - 1) & 1], 0));
  if (hipEventQuery(p->evDet[f.a]) != hipSuccess) VO_HIP_TRY(ctx, hipStreamWaitEvent(ts, p->evDet[f.a], 0));
  vo_seq_ctl* ctl = p->d_ctl + q0;
  vo_klt_source src;
  src.n = &ctl->n2;                  // (= n once the previous step has closed; known as soon as its regroup has run)
  src.num_features = &ctl->num_features;
  src.frac = c.redetect_fraction;
  src.det_kp = p->kp(q0, f.a);
  src.n_det = c.n_keypoints;
  src.ts = &ctl->ts[0];
  src.det_go = p->d_det_go + (size_t)f.a * p->S + q0;
  if (gated) {
    src.gate_wait = &ctl->gate_regroup;
    src.gate_want = (uint32_t)f.k;                    // published by the regroup of flight k - 1 (or a state hand-over)
    src.gate_set = &ctl->gate_klt;
    src.gate_cnt = &ctl->gate_klt_cnt;
    src.gate_set_to = (uint32_t)f.k + 1u;
    src.gate_fault = &ctl->fault;
    src.gate_mode = p->gate_mode;
  }
  vo_klt_batch kb;
  kb.S = Sn;
  kb.pyr = p->pyr_stride();
  kb.xy = (size_t)p->cap * 2;
  kb.out = (size_t)p->cap;
  kb.ctl = sizeof(vo_seq_ctl);
  kb.det = p->det_stride();
  // The tracker's and the regroup's events are the kernels' own completion signals (vo_ctx::next_stop), not markers behind
  // them: a marker between the regroup and the hypothesis kernel cost the main chain 3.7 us, the tracker started 3.4 us
  // later behind it (step 85.8 -> 84.2 us).  VO_EXT_EVENTS=0: hipEventRecord.
  const bool ext_events = p->ext_events;
  if (ext_events) p->trk->next_stop = p->evKlt[f.k & 1];
  {
    const size_t q = (size_t)q0;
    const int rc = vo_klt_track_ndev(p->trk, p->img(q0, f.prev_idx), p->pyr(q0, f.a), p->img(q0, f.next_idx), p->pyr(q0, f.b),
                                     c.H, c.W, p->n_levels, A.kp, p->cap, nullptr, c.klt_win, c.klt_max_iter, c.klt_eps,
                                     c.klt_min_eig, p->d_next + q * p->cap * 2, p->d_status + q * p->cap,
                                     p->d_err + q * p->cap, &src, &kb);
    if (rc != VO_OK) return vo_set_error(ctx, rc, "tracker: %s", vo_last_error(p->trk));
  }
  if (p->trk->next_stop) {             // (the launch did not take the event)
    p->trk->next_stop = nullptr;
    VO_HIP_TRY(ctx, hipEventRecord(p->evKlt[f.k & 1], ts));
  } else if (!ext_events) {
    VO_HIP_TRY(ctx, hipEventRecord(p->evKlt[f.k & 1], ts));
  }
  return VO_OK;
}

static int enqueue_pose_half(vo_pipeline* p, const vo_pipeline::flight_t& f, int q0, int Sn, unsigned seq);

// the main-stream chain of one step (the tracker's event must have been recorded);
// first_half_only: stop behind the regroup (recover_step continues on the host)
static int enqueue_chain(vo_pipeline* p, const vo_pipeline::flight_t& f, bool first_half_only, int debug_fault_every,
                         int q0, int Sn, unsigned seq, bool gated = false) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  const size_t q = (size_t)q0;
  const vo_feat A = vo_feat_seq(p->F[f.fcur], q), B = vo_feat_seq(p->F[1 - f.fcur], q);
  vo_seq_ctl* ctl = p->d_ctl + q0;
  if (!gated) VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evKlt[f.k & 1], 0));
  vo_append ap;
  ap.gate_klt_want = gated ? (uint32_t)f.k + 1u : 0u;
  ap.gate_regroup_set = gated ? (uint32_t)f.k + 1u : 0u;
  ap.gate_mode = p->gate_mode;
  ap.det_kp = p->kp(q0, f.a);
  ap.det_stride = p->det_stride();
  ap.n_det = c.n_keypoints;
  ap.frac = c.redetect_fraction;
  ap.pose_mode = c.redetect_start_pose;
  ap.debug_fault_every = debug_fault_every > 0 ? debug_fault_every : 0;
  ap.det_go = p->d_det_go + (size_t)f.a * p->S + q0;
  const bool ext_events = p->ext_events;
  if (ext_events) ctx->next_stop = p->evRegroup[f.k & 1];
  VO_TRY(vo_state_regroup_klt(ctx, ctl, A, B, p->d_next + q * p->cap * 2, p->d_status + q * p->cap, p->d_err + q * p->cap,
                              (float)c.klt_err_threshold, ap, p->cap, Sn));
  if (!ext_events) VO_HIP_TRY(ctx, hipEventRecord(p->evRegroup[f.k & 1], ctx->stream));
  if (first_half_only) return VO_OK;
  return enqueue_pose_half(p, f, q0, Sn, seq);
}

// hypotheses + pose kernel of one step (the second half of its main-stream chain; also the next batch of a step whose
// RANSAC loop continues)
static int enqueue_pose_half(vo_pipeline* p, const vo_pipeline::flight_t& f, int q0, int Sn, unsigned seq) {
  vo_ctx* ctx = p->ctx;
  const int debug_pose_fault = p->pose_fault_hook && p->cfg.debug_fault_every < 0 ? -p->cfg.debug_fault_every : 0;
  const vo_pipeline_config& c = p->cfg;
  const size_t q = (size_t)q0;
  const vo_feat B = vo_feat_seq(p->F[1 - f.fcur], q);
  vo_seq_ctl* ctl = p->d_ctl + q0;
  vo_hyp_batch hb;
  hb.S = Sn;
  hb.X = (size_t)p->cap * 3;
  hb.x = (size_t)p->cap * 2;
  hb.raws = p->ring_len;
  hb.ctl = sizeof(vo_seq_ctl);
  VO_TRY(vo_p3p_hypotheses_ring_dev(ctx, B.land, B.kp64, &ctl->n_p3p, p->cap, c.K, p->d_raws + q * p->ring_len, &ctl->raw_pos,
                                    p->ring_len - 1, c.hyp, c.p3p_thr_sq, p->d_R + q * c.hyp * 9, p->d_t + q * c.hyp * 3,
                                    p->d_valid + q * c.hyp, p->d_counts + q * c.hyp, p->d_masks + q * c.hyp * p->words,
                                    (uint32_t*)&ctl->solve_flag, (uint64_t*)&ctl->ts[2], &hb));
  // pose, candidates, candidate triangulation, landmark update and the record in one launch (frame_pose_kernel's tail)
  vo_pose_job job = make_pose_job(p, p->F[1 - f.fcur], 1, q0);
  job.tail = 1;
  job.res = p->m_res + (size_t)f.rslot * p->S + q;
  job.seq_word = p->m_seq + (size_t)f.rslot * p->S + q;
  job.seq = seq;
  job.debug_fault_every = debug_pose_fault;
  // The landmark stage as its own launch of cap / 256 workgroups per sequence.  (Round 2 had the pose kernel's one
  // workgroup go on with it -- VO_FUSED_TAIL=1 -- which was neutral at ~450 candidates per frame on a stream that never
  // lost a track; the forward stream triangulates ~1100 per frame, three rounds of DLTs for one workgroup: 50 us
  // against 18 for the launch, its boundary included; step period 152 -> 121 us.)
  static const bool split_tail = getenv("VO_FUSED_TAIL") == nullptr;
  if (split_tail) {
    // The feature walk (reset_outliers, bearing-angle candidates: fp64 arithmetic of every feature) was the last third of the
    // pose kernel, on its ONE compute unit: 10 us.  VO_SPLIT_WALK=1: a launch of its own, cap / 256 workgroups, between the pose
    // kernel and the landmark stage (step period 93.4 -> 87.8 us); default (2): walk and landmark stage in ONE launch
    // (state_walk_landmarks_kernel), one boundary less; 0: the walk inside the pose kernel.
    static const int walk_mode = getenv("VO_SPLIT_WALK") ? atoi(getenv("VO_SPLIT_WALK")) : 2;
    job.tail = 0;
    job.walk = walk_mode ? 0 : 1;
    VO_TRY(vo_frame_pose(ctx, job, Sn));
    const uint64_t* bm = p->d_best_mask + (size_t)q0 * p->words;
    const int rec_refined = c.refine_iters > 0 ? 1 : 0;
    if (walk_mode == 2) {
      VO_TRY(vo_state_walk_landmarks(ctx, ctl, B, bm, p->words, p->cam, c.bearing_threshold, rec_refined, p->cap,
                                     p->d_pend + q * p->cap, job.res, job.seq_word, seq, Sn));
      return VO_OK;
    }
    if (walk_mode == 1)
      VO_TRY(vo_state_candidates(ctx, ctl, B, bm, p->cam, c.bearing_threshold, 1 /* ctl->refined: what the pose kernel's walk uses */,
                                 p->cap, Sn, p->words));
    VO_TRY(vo_state_landmarks(ctx, ctl, B, p->cam, rec_refined, p->cap, job.res, job.seq_word, seq, Sn));
    return VO_OK;
  }
  VO_TRY(vo_frame_pose(ctx, job, Sn));
  return VO_OK;
}

// ---- SIFT tracker mode ----
extern "C" int vo_sift_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int cap, float* d_kp, float* d_desc,
                           uint8_t* d_desc_u8, int32_t* d_n);

// detect + describe of the step's new frame on the tracker's stream (it depends on the image only): slot f.b
// (The scale space of a frame is a chain of ~60 dependent launches, most of them on images too small to fill anything:
//  consecutive frames alternate between two contexts -- the tracker's and, idle in this mode, the detection's -- each with
//  its own streams and arena, so that two frames' chains are in flight side by side.  The launches are made by the worker
//  thread; err_buf: its private error text.)
static int enqueue_sift(vo_pipeline* p, const vo_pipeline::flight_t& f, char* err_buf = nullptr) {
  static const bool one_ctx = getenv("VO_SIFT_ONE_CONTEXT") != nullptr;
  vo_ctx* sc = ((f.k & 1) && !one_ctx) ? p->det : p->trk;
  const vo_pipeline_config& c = p->cfg;
  int rc = VO_OK;
  if ((sc != p->trk || p->side[(size_t)f.next_idx]) && hipStreamWaitEvent(sc->stream, p->evImg[f.next_idx], 0) != hipSuccess) rc = VO_EHIP;   // (the upload)
  if (rc == VO_OK)
    rc = vo_sift_dev(sc, p->img(0, f.next_idx), c.H, c.W, p->sift_cap, p->d_skp + (size_t)f.b * p->sift_cap * 6, nullptr,
                     p->d_sdesc + (size_t)f.b * p->sift_cap * 128, p->d_sn + f.b);
  if (rc == VO_OK && hipEventRecord(p->evPyr[f.b], sc->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) {
    if (err_buf) {
      snprintf(err_buf, 256, "sift: %s", vo_last_error(sc));
      return rc;
    }
    return vo_set_error(p->ctx, rc, "sift: %s", vo_last_error(sc));
  }
  return VO_OK;
}

// Harris tracker mode: the new frame's N keypoints (Harris response + greedy NMS, every frame) and their raw patches as
// bytes, on the detection stream; slot f.b
static int enqueue_harris_front(vo_pipeline* p, const vo_pipeline::flight_t& f, char* err_buf = nullptr) {
  const vo_pipeline_config& c = p->cfg;
  int rc = enqueue_detection(p, f.next_idx, f.b, true, err_buf);
  if (rc != VO_OK) return rc;
  rc = vo_patch_descriptors_u8_dev(p->det, p->img(0, f.next_idx), c.H, c.W, p->kp(0, f.b), c.n_keypoints, 9,
                                   p->d_sdesc + (size_t)f.b * p->sift_cap * p->desc_row, p->desc_row);
  if (rc == VO_OK && hipEventRecord(p->evPyr[f.b], p->det->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) {
    if (err_buf) {
      snprintf(err_buf, 256, "harris front: %s", vo_last_error(p->det));
      return rc;
    }
    return vo_set_error(p->ctx, rc, "harris front: %s", vo_last_error(p->det));
  }
  return VO_OK;
}

// main-stream chain of a step in SIFT mode: 2-NN + ratio + uniqueness against the current Features' descriptors
// (sift.py:38-54), Matches regroup from the pair list (matches.py:26-212) with the descriptors following their
// keypoints, then hypotheses and pose as in the KLT mode
static int enqueue_chain_sift(vo_pipeline* p, const vo_pipeline::flight_t& f, bool first_half_only, unsigned seq) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  hipStream_t st = ctx->stream;
  const bool harris = c.tracker_mode == 2;
  const size_t row = (size_t)p->desc_row;
  const vo_feat A = p->F[f.fcur], B = p->F[1 - f.fcur];
  const uint8_t* descA = p->d_fdesc + (size_t)f.fcur * p->cap * row;
  uint8_t* descB = p->d_fdesc + (size_t)(1 - f.fcur) * p->cap * row;
  const uint8_t* sdesc = p->d_sdesc + (size_t)f.b * p->sift_cap * row;
  const float* skp = p->d_skp + (size_t)f.b * p->sift_cap * 6;
  int32_t* n_new = p->d_sn + f.b;
  int32_t* n_pairs = p->d_sn + 3;
  VO_HIP_TRY(ctx, hipStreamWaitEvent(st, p->evPyr[f.b], 0));
  const double ratio = c.match_ratio > 0.0 ? c.match_ratio : (harris ? 0.85 : 0.8);        // harris.py:255 / sift.py:49
  VO_TRY(vo_match_u8_dev(ctx, descA, &p->d_ctl->n, p->cap, sdesc, n_new, p->sift_cap, ratio, p->d_pairs, n_pairs, p->desc_row));
  const double* new_kp = p->d_newkp;
  if (harris) {
    new_kp = p->kp(0, f.b);            // the detector's keypoints are float64 pairs already
  } else {
    hipLaunchKernelGGL(sift_kp_f64_kernel, dim3(vo_cdiv(p->sift_cap, 256)), dim3(256), 0, st, skp, (const int*)n_new, p->sift_cap,
                       p->d_newkp);
    VO_TRY(vo_check_launch(ctx, "sift_kp_f64_kernel"));
  }
  VO_TRY(vo_state_regroup_pairs(ctx, p->d_ctl, A, B, p->d_pairs, p->cap, new_kp, p->sift_cap, p->cap, n_pairs, n_new,
                                p->d_srcrow));
  hipLaunchKernelGGL(desc_gather_kernel, dim3(vo_cdiv(p->cap * (p->desc_row / 4), 256)), dim3(256), 0, st, sdesc,
                     (const int*)p->d_srcrow, (const vo_seq_ctl*)p->d_ctl, p->cap, descB, p->desc_row / 4);
  VO_TRY(vo_check_launch(ctx, "desc_gather_kernel"));
  VO_HIP_TRY(ctx, hipEventRecord(p->evRegroup[f.k & 1], st));
  if (first_half_only) return VO_OK;
  return enqueue_pose_half(p, f, 0, 1, seq);
}

// ---- detection worker ----
static void worker_main(vo_pipeline* p) {
  (void)hipSetDevice(p->ctx->device);
  unsigned seen = 0;
  long idle = 0;
  double idle_since = 0.0;
  for (;;) {
    if (p->job_posted.load(std::memory_order_acquire) == seen) {
      if (p->quit.load(std::memory_order_acquire)) return;
      // a step is ~120 us: stay hot between the steps of a running stream, then sleep until a job is posted
      if (idle == 0) idle_since = now_s();
      if ((++idle & 63) != 0 || now_s() - idle_since < p->spin_s) {
        __builtin_ia32_pause();
      } else {
        p->worker_asleep.store(1, std::memory_order_seq_cst);
        if (p->job_posted.load(std::memory_order_seq_cst) == seen && !p->quit.load(std::memory_order_seq_cst))
          futex_wait(&p->job_posted, seen);
        p->worker_asleep.store(0, std::memory_order_seq_cst);
        idle = 0;
      }
      continue;
    }
    idle = 0;
    const vo_pipeline::flight_t j = p->jobs[seen & 3];
    const int rc = p->cfg.tracker_mode == 1   ? enqueue_sift(p, j, p->worker_err)
                   : p->cfg.tracker_mode == 2 ? enqueue_harris_front(p, j, p->worker_err)
                                              : enqueue_detection(p, j.next_idx, j.b, false, p->worker_err);
    if (rc != VO_OK) p->worker_rc = rc;
    ++seen;
    p->job_done.store(seen, std::memory_order_release);
  }
}

static int worker_check(vo_pipeline* p) {
  if (p->worker_rc != VO_OK) {
    const int rc = p->worker_rc;
    p->worker_rc = VO_OK;
    return vo_set_error(p->ctx, rc, "%s", p->worker_err);
  }
  return VO_OK;
}

// waits (host) until the worker has enqueued everything it was given
static int worker_idle(vo_pipeline* p) {
  if (p->threads_budget < 2) return VO_OK;
  const unsigned posted = p->job_posted.load(std::memory_order_relaxed);
  wait_until(50e-6, [&] { return p->job_done.load(std::memory_order_acquire) == posted; });
  return worker_check(p);
}

extern "C" {

int vo_pipeline_set_state_seq(vo_pipeline* p, int seq, int idx, int n, const double* kp, const uint8_t* state,
                              const double* landmarks, const double* tracks, const double* poses, const double* T_wc,
                              const double* T_cw, const double* T_wc_prev, const double* T_cw_prev, int num_features) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, seq >= 0 && seq < p->S, "pipeline_set_state: bad sequence index");
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames, "pipeline_set_state: bad frame index");
  VO_REQUIRE(ctx, n >= 0 && n <= p->cap, "pipeline_set_state: %d features exceed the capacity %d", n, p->cap);
  VO_REQUIRE(ctx, (n == 0 || (kp && state && landmarks && tracks && poses)) && T_wc && T_cw && T_wc_prev && T_cw_prev,
             "pipeline_set_state: null pointer");
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_set_state: %d submitted step(s) not collected", p->n_flight);
  // the sequences step together through one frame slot: the states of one hand-over all belong to the same frame
  VO_REQUIRE(ctx, !(p->S > 1 && p->have_state && !p->primed && idx != p->prev_frame),
             "pipeline_set_state: sequence %d is handed over for frame %d, the others of this hand-over for frame %d", seq, idx,
             p->prev_frame);
  VO_TRY(worker_idle(p));
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  const vo_feat F = vo_feat_seq(p->F[p->cur], (size_t)seq);
  std::vector<double> pose12((size_t)n * 12);
  std::vector<float> kp32((size_t)n * 2);
  std::vector<uint8_t> zeros((size_t)n, 0);
  for (int i = 0; i < 2 * n; ++i) kp32[i] = (float)kp[i];
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 12; ++k) pose12[(size_t)k * n + i] = poses[(size_t)16 * i + k];   // component-major on the device
  if (n > 0) {
    VO_HIP_TRY(ctx, mcpy(st, F.kp, kp32.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    VO_HIP_TRY(ctx, mcpy(st, F.kp64, kp, (size_t)n * 16, hipMemcpyHostToDevice));
    VO_HIP_TRY(ctx, mcpy(st, F.state, state, (size_t)n, hipMemcpyHostToDevice));
    VO_HIP_TRY(ctx, mcpy(st, F.cand, zeros.data(), (size_t)n, hipMemcpyHostToDevice));
    VO_HIP_TRY(ctx, mcpy(st, F.land, landmarks, (size_t)n * 24, hipMemcpyHostToDevice));
    VO_HIP_TRY(ctx, mcpy(st, F.track, tracks, (size_t)n * 16, hipMemcpyHostToDevice));
    for (int k = 0; k < 12; ++k)
      VO_HIP_TRY(ctx, mcpy(st, F.pose + (size_t)k * F.pitch, &pose12[(size_t)k * n], (size_t)n * 8, hipMemcpyHostToDevice));
  }
  vo_seq_ctl h;
  VO_HIP_TRY(ctx, mcpy(st, &h, p->d_ctl + seq, sizeof(h), hipMemcpyDeviceToHost));
  const uint64_t raw_pos = h.raw_pos;
  const int64_t n_it = h.n_iterations;
  const double orat = h.outlier_ratio;
  const uint32_t gate_klt = h.gate_klt;
  const bool keep_ransac = p->seq_state[seq] != 0;
  memset(&h, 0, sizeof(h));
  h.n = n;
  h.n2 = n;
  h.num_features = num_features;
  h.raw_pos = raw_pos;
  h.gate_regroup = (uint32_t)p->steps_submitted;      // the features the next flight's tracker waits for are these
  h.gate_klt = gate_klt;
  if (keep_ransac) {
    h.n_iterations = n_it;
    h.outlier_ratio = orat;
  } else {
    // RANSAC.__init__ (ransac.py:47-56)
    h.outlier_ratio = p->cfg.ransac_outlier_ratio;
    const int64_t k0 = vo_ransac_num_iterations(p->cfg.ransac_confidence, p->cfg.ransac_outlier_ratio, 4);
    h.n_iterations = (p->cfg.ransac_max_iterations >= 0 && p->cfg.ransac_max_iterations < k0) ? p->cfg.ransac_max_iterations : k0;
  }
  memcpy(h.T_wc, T_wc, 96);
  memcpy(h.T_cw, T_cw, 96);
  memcpy(h.T_wc_prev, T_wc_prev, 96);
  memcpy(h.T_cw_prev, T_cw_prev, 96);
  VO_HIP_TRY(ctx, mcpy(st, p->d_ctl + seq, &h, sizeof(h), hipMemcpyHostToDevice));
  // the pyramid and the detector's output of the frame the states belong to are made by the first submit
  // (for all sequences at once: they share the frame slot, the last call's idx counts)
  p->seq_state[seq] = 1;
  p->slot = 0;
  p->prev_frame = idx;
  p->have_state = true;
  p->primed = false;
  return VO_OK;
}

int vo_pipeline_set_state(vo_pipeline* p, int idx, int n, const double* kp, const uint8_t* state,
                          const double* landmarks, const double* tracks, const double* poses, const double* T_wc,
                          const double* T_cw, const double* T_wc_prev, const double* T_cw_prev, int num_features) {
  return vo_pipeline_set_state_seq(p, 0, idx, n, kp, state, landmarks, tracks, poses, T_wc, T_cw, T_wc_prev, T_cw_prev,
                                   num_features);
}

int vo_pipeline_set_descriptors(vo_pipeline* p, const float* desc, int n) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->cfg.tracker_mode != 0, "pipeline_set_descriptors: the pipeline is not in a descriptor tracker mode");
  VO_REQUIRE(ctx, p->have_state && p->n_flight == 0, "pipeline_set_descriptors: hand the state over first (nothing in flight)");
  VO_REQUIRE(ctx, n >= 0 && n <= p->cap && (n == 0 || desc), "pipeline_set_descriptors: bad arguments");
  const int D = p->cfg.tracker_mode == 2 ? 361 : 128;     // values per row handed in; rows are padded to desc_row bytes
  std::vector<uint8_t> b((size_t)n * p->desc_row, 0);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < D; ++k) {
      const float v = desc[(size_t)i * D + k];
      VO_REQUIRE(ctx, v >= 0.f && v <= 255.f && v == (float)(int)v, "pipeline_set_descriptors: descriptor values must be whole numbers 0..255");
      b[(size_t)i * p->desc_row + k] = (uint8_t)v;
    }
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n > 0)
    VO_HIP_TRY(ctx, mcpy(ctx->stream, p->d_fdesc + (size_t)p->cur * p->cap * p->desc_row, b.data(), b.size(), hipMemcpyHostToDevice));
  return VO_OK;
}

// pyramid and detection of the frame the handed-over states belong to, all sequences, synchronously
static int prime(vo_pipeline* p, bool wait = true) {
  vo_ctx* ctx = p->ctx;
  if (p->cfg.tracker_mode != 0) {      // descriptor modes: the frame's own descriptors travel with its Features
    p->primed = true;
    return VO_OK;
  }
  VO_TRY(worker_idle(p));
  sync_prof(p);
  p->prepared_idx = p->prepared_slot = -1;           // (a hand-over or a rewind: the slots start over)
  VO_TRY(enqueue_pyramid(p, p->prev_frame, p->slot));
  VO_TRY(enqueue_detection(p, p->prev_frame, p->slot, true));
  if (wait) {        // (not needed for order: the tracker sits behind the pyramid on its stream and waits for evDet)
    VO_HIP_TRY(ctx, hipStreamSynchronize(p->trk->stream));
    VO_HIP_TRY(ctx, hipStreamSynchronize(p->det->stream));
  }
  p->primed = true;
  return VO_OK;
}

extern "C" int vo_pipeline_checkpoint(vo_pipeline* p) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->have_state, "pipeline_checkpoint: no state was handed over");
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_checkpoint: %d submitted step(s) not collected", p->n_flight);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!p->d_ckpt_feat) {
    VO_TRY(dev_alloc(ctx, &p->d_ckpt_feat, p->feat_block));
    VO_TRY(dev_alloc(ctx, &p->d_ckpt_ctl, (size_t)p->S));
  }
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_ckpt_feat, (char*)p->feat_mem + (size_t)p->cur * p->feat_block, p->feat_block,
                                 hipMemcpyDeviceToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_ckpt_ctl, p->d_ctl, (size_t)p->S * sizeof(vo_seq_ctl), hipMemcpyDeviceToDevice, st));
  if (p->cfg.tracker_mode != 0) {
    if (!p->d_ckpt_fdesc) VO_TRY(dev_alloc(ctx, &p->d_ckpt_fdesc, (size_t)p->cap * p->desc_row));
    VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_ckpt_fdesc, p->d_fdesc + (size_t)p->cur * p->cap * p->desc_row, (size_t)p->cap * p->desc_row,
                                   hipMemcpyDeviceToDevice, st));
  }
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  p->ckpt_frame = p->prev_frame;
  return VO_OK;
}

extern "C" int vo_pipeline_rewind(vo_pipeline* p) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->ckpt_frame >= 0, "pipeline_rewind: no checkpoint");
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_rewind: %d submitted step(s) not collected", p->n_flight);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  VO_TRY(worker_idle(p));
  hipStream_t st = ctx->stream;
  // every step has been collected: its chain -- tracker included -- is done, nothing reads the Features any more
  VO_HIP_TRY(ctx, hipMemcpyAsync((char*)p->feat_mem + (size_t)p->cur * p->feat_block, p->d_ckpt_feat, p->feat_block,
                                 hipMemcpyDeviceToDevice, st));
  if (p->cfg.tracker_mode != 0)
    VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_fdesc + (size_t)p->cur * p->cap * p->desc_row, p->d_ckpt_fdesc,
                                   (size_t)p->cap * p->desc_row, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(ctl_rewind_kernel, dim3(vo_cdiv(p->S, 64)), dim3(64), 0, st, p->d_ctl, p->d_ckpt_ctl, p->S);
  VO_TRY(vo_check_launch(ctx, "ctl_rewind_kernel"));
  // the next step's tracker waits for "the previous step's regroup": that event now stands for the restored state
  if (p->steps_submitted > 0) VO_HIP_TRY(ctx, hipEventRecord(p->evRegroup[(p->steps_submitted - 1) & 1], st));
  else VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  p->slot = 0;
  p->prev_frame = p->ckpt_frame;
  p->gate_resync = true;             // (the next tracker waits for that event, not only for its gate)
  return prime(p, false);            // pyramid + detector of that frame, queued on their streams
}

int vo_pipeline_get_state_seq(vo_pipeline* p, int seq, int32_t* n_out, double* kp, uint8_t* state,
                              uint8_t* candidate_mask, double* landmarks, double* tracks, double* poses, double* T_wc,
                              double* T_wc_prev, vo_ransac_state* rs, int32_t* num_features) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, seq >= 0 && seq < p->S, "pipeline_get_state: bad sequence index");
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_get_state: %d submitted step(s) not collected", p->n_flight);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  vo_seq_ctl h;
  VO_HIP_TRY(ctx, mcpy(st, &h, p->d_ctl + seq, sizeof(h), hipMemcpyDeviceToHost));
  const int n = h.n;
  const vo_feat F = vo_feat_seq(p->F[p->cur], (size_t)seq);
  if (n_out) *n_out = n;
  if (num_features) *num_features = h.num_features;
  if (n > 0) {
    if (kp) VO_HIP_TRY(ctx, mcpy(st, kp, F.kp64, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (state) VO_HIP_TRY(ctx, mcpy(st, state, F.state, (size_t)n, hipMemcpyDeviceToHost));
    if (candidate_mask) VO_HIP_TRY(ctx, mcpy(st, candidate_mask, F.cand, (size_t)n, hipMemcpyDeviceToHost));
    if (landmarks) VO_HIP_TRY(ctx, mcpy(st, landmarks, F.land, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (tracks) VO_HIP_TRY(ctx, mcpy(st, tracks, F.track, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (poses) {
      std::vector<double> p12((size_t)n * 12);
      for (int k = 0; k < 12; ++k)
        VO_HIP_TRY(ctx, mcpy(st, &p12[(size_t)k * n], F.pose + (size_t)k * F.pitch, (size_t)n * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i) {
        double row[12];
        for (int k = 0; k < 12; ++k) row[k] = p12[(size_t)k * n + i];
        expand_pose(row, poses + (size_t)16 * i);
      }
    }
  }
  if (T_wc) expand_pose(h.T_wc, T_wc);
  if (T_wc_prev) expand_pose(h.T_wc_prev, T_wc_prev);
  if (rs) {
    rs->outlier_ratio = h.outlier_ratio;
    rs->confidence = p->cfg.ransac_confidence;
    rs->max_iterations = p->cfg.ransac_max_iterations;
    rs->n_iterations = h.n_iterations;
    rs->s = 4;
    rs->adaptive = 1;
  }
  return VO_OK;
}

int vo_pipeline_get_state(vo_pipeline* p, int32_t* n_out, double* kp, uint8_t* state, uint8_t* candidate_mask,
                          double* landmarks, double* tracks, double* poses, double* T_wc, double* T_wc_prev,
                          vo_ransac_state* rs, int32_t* num_features) {
  return vo_pipeline_get_state_seq(p, 0, n_out, kp, state, candidate_mask, landmarks, tracks, poses, T_wc, T_wc_prev, rs,
                                   num_features);
}

int vo_pipeline_get_detection(vo_pipeline* p, double* kp_xy) {
  if (!p || !kp_xy) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_get_detection: %d submitted step(s) not collected", p->n_flight);
  VO_TRY(worker_idle(p));
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (p->have_state && !p->primed) VO_TRY(prime(p));
  VO_HIP_TRY(ctx, hipEventSynchronize(p->evDet[p->slot]));
  {
    int ran = 0;
    VO_HIP_TRY(ctx, mcpy(ctx->stream, &ran, p->d_det_go + (size_t)p->slot * p->S, 4, hipMemcpyDeviceToHost));
    if (!ran) {                          // the frame's detection was skipped: made now (all sequences)
      VO_TRY(enqueue_detection(p, p->prev_frame, p->slot, true));
      VO_HIP_TRY(ctx, hipStreamSynchronize(p->det->stream));
    }
  }
  VO_HIP_TRY(ctx, mcpy(ctx->stream, kp_xy, p->kp(0, p->slot), (size_t)p->cfg.n_keypoints * 16, hipMemcpyDeviceToHost));
  return VO_OK;
}

}  // extern "C"

// SIFT mode: enqueue the main-stream chains of the flights that do not have theirs yet (oldest first)
// (in flight order, as far as their SIFT launches have been made -- by this thread, or by the worker: sift_job = the
//  worker's job count that says so; must_reach: flights up to this index are waited for)
static int sift_flush_chains(vo_pipeline* p, int must_reach = -1) {
  while (p->sift_chain_pending > 0) {
    const int k = p->n_flight - p->sift_chain_pending;
    const vo_pipeline::flight_t& f = p->flight[k];
    if (f.sift_job != 0 && (int)(p->job_done.load(std::memory_order_acquire) - f.sift_job) < 0) {
      if (k > must_reach) break;
      const unsigned want = f.sift_job;
      wait_until(50e-6, [&] { return (int)(p->job_done.load(std::memory_order_acquire) - want) >= 0; });
    }
    VO_TRY(worker_check(p));
    VO_TRY(ensure_raws(p, 0));
    VO_TRY(enqueue_chain_sift(p, f, false, f.seq));
    --p->sift_chain_pending;
  }
  return VO_OK;
}

extern "C" {

int vo_pipeline_submit(vo_pipeline* p, int prev_idx, int next_idx) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  VO_REQUIRE(ctx, next_idx >= 0 && next_idx < c.n_frames, "pipeline_submit: bad frame index");
  VO_REQUIRE(ctx, p->have_state && p->seeded, "pipeline_submit: call vo_pipeline_seed and vo_pipeline_set_state first");
  VO_REQUIRE(ctx, prev_idx == p->prev_frame, "pipeline_submit: prev frame %d is not the frame last submitted (%d)",
             prev_idx, p->prev_frame);
  VO_REQUIRE(ctx, p->n_flight < 2, "pipeline_submit: two steps are already in flight, collect one first");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!p->primed) VO_TRY(prime(p));
  const double t_in = now_s();
  vo_pipeline::flight_t f;
  f.prev_idx = prev_idx;
  f.next_idx = next_idx;
  f.a = p->slot;
  f.b = (p->slot + 1) % 3;
  f.fcur = p->cur;
  f.seq = ++p->seq;
  f.rslot = (int)(p->steps_submitted & 3);
  f.k = p->steps_submitted;
  // The detection of `next` (half of the step's launches, needed only by the NEXT step) goes to the worker thread;
  // this thread enqueues the pyramid, the tracker and the main-stream chain.  The tracker waits for the event behind
  // the detection of `prev`: the worker must have recorded it (it was posted a whole step ago).
  if (c.tracker_mode != 0) {
    // The frame's ~75 SIFT launches go to the worker thread; this thread first gives the flight submitted before its
    // main-stream chain (its SIFT launches are made by now), so the two threads' launches overlap across frames.
    // Two threads make the launches, a frame each: even flights' go to the worker, odd flights' are made here (each
    // thread on its own SIFT context, so that two frames' chains also run side by side on the GPU).
    f.sift_job = 0;
    if (p->threads_budget >= 2 && ((f.k & 1) == 0 || c.tracker_mode == 2)) {      // (harris: one detection context, the worker's)
      const unsigned my = p->job_posted.load(std::memory_order_relaxed);
      if ((int)(p->job_done.load(std::memory_order_acquire) - my) >= 0) {       // (idle worker: its context's flags are ours)
        p->trk->prof_on = ctx->prof_on;
        p->trk->prof_kernel = ctx->prof_kernel;
        p->trk->prof_every = ctx->prof_every;
        p->det->prof_on = ctx->prof_on;
        p->det->prof_kernel = ctx->prof_kernel;
        p->det->prof_every = ctx->prof_every;
      }
      f.sift_job = my + 1;
      p->jobs[my & 3] = f;
      p->job_posted.store(my + 1, std::memory_order_seq_cst);
      if (p->worker_asleep.load(std::memory_order_seq_cst)) futex_wake(&p->job_posted);
    } else {
      vo_ctx* mine = (f.k & 1) ? p->det : p->trk;          // (budget 1: both contexts are this thread's)
      mine->prof_on = ctx->prof_on;
      mine->prof_kernel = ctx->prof_kernel;
      mine->prof_every = ctx->prof_every;
      if (c.tracker_mode == 2) {
        p->det->prof_on = ctx->prof_on;
        p->det->prof_kernel = ctx->prof_kernel;
        p->det->prof_every = ctx->prof_every;
        VO_TRY(enqueue_harris_front(p, f));
      } else {
        VO_TRY(enqueue_sift(p, f));
      }
    }
    p->slot_seq[(size_t)f.rslot] = f.seq;
    p->flight[p->n_flight++] = f;
    ++p->sift_chain_pending;
    VO_TRY(sift_flush_chains(p));
    ++p->steps_submitted;
    p->slot = f.b;
    p->cur = 1 - f.fcur;
    p->prev_frame = next_idx;
    p->dbg_submit += now_s() - t_in;
    ++p->dbg_steps;
    return VO_OK;
  }
  double tq = now_s();
  // WHEN the detection's seven launches reach the GPU matters more than who makes them.  Arriving beside the hypothesis
  // kernel -- the worker used to get them at the start of submit -- they cost that kernel 20 us (hypotheses -> pose 43 us
  // against 23.5: a chain of launches that mostly return at once still keeps the command processor busy while the
  // 144 workgroups of the hypotheses are being dispatched), 121 against 108 us per step.  So they are handed to the worker
  // behind the step's chain -- unless the detector executes on every frame (detect_margin < 0): its kernels are then real
  // work the next step's tracker waits for, and an early start pays (16 sequences: 19.2k against 17.4k frames/s).
  const bool detect_early = p->detect_limit < 0.0;
  auto post_detection = [&]() -> int {
    const unsigned my = p->job_posted.load(std::memory_order_relaxed);
    p->jobs[my & 3] = f;
    p->job_posted.store(my + 1, std::memory_order_seq_cst);
    if (p->worker_asleep.load(std::memory_order_seq_cst)) futex_wake(&p->job_posted);
    return VO_OK;
  };
  if (p->threads_budget >= 2) {
    const unsigned my = p->job_posted.load(std::memory_order_relaxed);
    wait_until(50e-6, [&] { return (int)(p->job_done.load(std::memory_order_acquire) - my) >= 0; });
    VO_TRY(worker_check(p));
    sync_prof(p);                      // (the worker is idle: the detection context's profiling flags are ours to write)
    if (detect_early) VO_TRY(post_detection());
  } else {
    sync_prof(p);
  }
  double tn = now_s();
  p->dbg_part[0] += tn - tq;
  tq = tn;
  const bool have_pyr = p->prepared_idx == next_idx && p->prepared_slot == f.b;
  p->prepared_idx = p->prepared_slot = -1;
  VO_TRY(enqueue_tracker(p, f, !have_pyr, 0, p->S, p->gates));
  tn = now_s();
  p->dbg_part[1] += tn - tq;
  tq = tn;
  for (int q = 0; q < p->S; ++q) VO_TRY(ensure_raws(p, q));
  tn = now_s();
  p->dbg_part[2] += tn - tq;
  tq = tn;
  VO_TRY(enqueue_chain(p, f, false, c.debug_fault_every, 0, p->S, f.seq, p->gates));
  p->gate_resync = false;
  for (int q = 0; q < p->S; ++q) p->slot_seq[(size_t)f.rslot * p->S + q] = f.seq;
  if (p->threads_budget < 2) VO_TRY(enqueue_detection(p, f.next_idx, f.b, false));   // (needed by the NEXT step only)
  else if (!detect_early) VO_TRY(post_detection());
  p->dbg_part[3] += now_s() - tq;
  p->flight[p->n_flight++] = f;
  ++p->steps_submitted;
  p->slot = f.b;
  p->cur = 1 - f.fcur;
  p->prev_frame = next_idx;
  p->dbg_submit += now_s() - t_in;
  ++p->dbg_steps;
  return VO_OK;
}

}  // extern "C"

// Waits for sequence q's record of step `seq` in slot rslot and copies it out.  The kernel writes the record,
// fences at system scope, then the sequence word; the record also carries the number at both ends and the generator
// position can only grow, so a copy taken while some of the record's lines were still on their way (seen twice in
// ~40k steps: the sequence word visible, a field behind it not yet) is recognised and taken again.
// state_device.h: seq_tail = the step's number, seq_head = the number XOR every other dword of the record
static bool record_fits(vo_step_result* out, unsigned seq) {
  const unsigned* dw = reinterpret_cast<const unsigned*>(out);
  unsigned x = 0u, y = 0u;
  for (size_t k = 0; k + 2 < sizeof(*out) / 4; ++k) {
    x ^= dw[k];
    y += vo_state_dev::record_mix(dw[k], (int)k);
  }
  if (out->seq_tail != seq + y || out->seq_head != (seq ^ x)) return false;
  out->seq_head = seq;           // (what the caller sees: both equal the step's number)
  out->seq_tail = seq;
  return true;
}

// The record's check as the C ABI exposes it (tests; a host that reads the mapped records itself): _seal writes the two
// closing words the way the device does, _check says whether a copy is one whole record of step `seq`.
extern "C" void vo_record_seal(vo_step_result* rec, unsigned seq) {
  if (!rec) return;
  const unsigned* dw = reinterpret_cast<const unsigned*>(rec);
  unsigned x = 0u, y = 0u;
  for (size_t k = 0; k + 2 < sizeof(*rec) / 4; ++k) {
    x ^= dw[k];
    y += vo_state_dev::record_mix(dw[k], (int)k);
  }
  rec->seq_head = seq ^ x;
  rec->seq_tail = seq + y;
}

extern "C" int vo_record_check(const vo_step_result* rec, unsigned seq) {
  if (!rec) return 0;
  vo_step_result copy = *rec;
  return record_fits(&copy, seq) ? 1 : 0;
}

static int wait_record(vo_pipeline* p, int rslot, int q, unsigned seq, uint64_t floor, vo_step_result* out) {
  volatile unsigned* w = p->seq_h(rslot, q);
  const double t0 = now_s();
  long it = 0;
  bool spinning = p->spin_s > 0.0;
  for (;;) {
    if (*w == seq) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      memcpy(out, (const void*)p->res_h(rslot, q), sizeof(*out));
      if (record_fits(out, seq) && out->raw_pos >= floor) return VO_OK;
    }
    // poll for spin_s, then look every 20 us (the GPU cannot wake a host thread; a blocking stream wait would also wait
    // for the look-ahead step queued behind this one)
    if (spinning) {
      __builtin_ia32_pause();
      if ((++it & 31) == 0 && now_s() - t0 > p->spin_s) spinning = false;
      continue;
    }
    nap(20000);
    if ((++it & 0xff) == 0 && now_s() - t0 > 5.0) {
      VO_HIP_TRY(p->ctx, hipStreamSynchronize(p->ctx->stream));
      memcpy(out, (const void*)p->res_h(rslot, q), sizeof(*out));
      if (*w == seq && record_fits(out, seq)) return VO_OK;
      return vo_set_error(p->ctx, VO_EHIP, "pipeline: the GPU never published the record of step %u (sequence %d)", seq, q);
    }
  }
}

// Sequence q's step of flight f raised a fault: nothing persistent of that sequence was touched, so the step is run
// again from its first main-stream kernel (for that sequence alone) with the sequential sampler and the reference's
// loop on the host (ransac.py:90-121), then handed back to the device for the refinement and the bookkeeping.
static int recover_step(vo_pipeline* p, const vo_pipeline::flight_t& f, int q, vo_step_result* out) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  hipStream_t st = ctx->stream;
  VO_TRY(worker_idle(p));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(p->trk->stream));
  vo_seq_ctl* ctl = p->d_ctl + q;
  vo_seq_ctl h;
  VO_HIP_TRY(ctx, mcpy(st, &h, ctl, sizeof(h), hipMemcpyDeviceToHost));
  const int fault_reason = h.fault;
  if (h.fault & VO_FAULT_CAPACITY)
    return vo_set_error(ctx, VO_ECAPACITY, "pipeline: %d features + %d new keypoints exceed the capacity %d", h.n,
                        c.n_keypoints, p->cap);
  const int zero = 0;
  VO_HIP_TRY(ctx, mcpy(st, &ctl->fault, &zero, 4, hipMemcpyHostToDevice));
  // The tracker reads its feature count from n2, which the step's own regroup has replaced by the NEW frame's count
  // when the fault came from the pose kernel (a possibly rejected draw, an unfinished loop): the tracker below would
  // redo only the first n2 features, and the rest of d_next would be whatever the next step's tracker left there --
  // the step's own values unless that one appended a detection (found by tests/pipeline_fuzz.py, now and then).
  if (c.tracker_mode == 0) VO_HIP_TRY(ctx, mcpy(st, &ctl->n2, &h.n, 4, hipMemcpyHostToDevice));
  // tracker and regroup of this sequence alone, without the forced fault.  A regroup that needs the detector's keypoints
  // of `prev` and finds that the detection was skipped (the tracks fell through the margin within one frame -- the
  // fault this step came with, or one that another fault had hidden) says so: the keypoints are made now, once more.
  for (int attempt = 0;; ++attempt) {
    if (h.fault & VO_FAULT_NO_DETECTION) {
      vo_ctx* det = p->det;
      VO_HIP_TRY(ctx, hipStreamSynchronize(det->stream));
      double* scores = p->d_scores[p->det_flip] + (size_t)q * p->px;
      det->nms_kp_f32 = nullptr;
      int rc = vo_harris_response_batch_dev(det, p->img(q, f.prev_idx), 0, 1, c.H, c.W, c.harris_patch, c.harris_kappa, scores);
      if (rc == VO_OK) rc = vo_nms_keypoints_batch_dev(det, scores, 1, c.H, c.W, c.n_keypoints, c.nms_radius, p->kp(q, f.a), 0);
      if (rc != VO_OK) return vo_set_error(ctx, rc, "detection: %s", vo_last_error(det));
      VO_HIP_TRY(ctx, hipStreamSynchronize(det->stream));
      const int one = 1;
      VO_HIP_TRY(ctx, mcpy(st, p->d_det_go + (size_t)f.a * p->S + q, &one, 4, hipMemcpyHostToDevice));
    }
    if (c.tracker_mode != 0) {
      VO_TRY(enqueue_chain_sift(p, f, true, 0u));      // (the frame's keypoints and descriptors are still in their slot)
    } else {
      VO_TRY(enqueue_tracker(p, f, false, q, 1));
      VO_TRY(enqueue_chain(p, f, true, 0, q, 1, 0u));
    }
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
    VO_HIP_TRY(ctx, mcpy(st, &h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (!(h.fault & VO_FAULT_NO_DETECTION) || attempt > 0) break;
    VO_HIP_TRY(ctx, mcpy(st, &ctl->fault, &zero, 4, hipMemcpyHostToDevice));
  }
  if (h.fault & VO_FAULT_CAPACITY)
    return vo_set_error(ctx, VO_ECAPACITY, "pipeline: %d features + %d new keypoints exceed the capacity %d", h.n,
                        c.n_keypoints, p->cap);
  if (h.fault & VO_FAULT_NO_DETECTION) return vo_set_error(ctx, VO_EHIP, "pipeline: the detector's keypoints are missing");
  const int n = h.n_tri;
  if (n < 4) return vo_set_error(ctx, VO_ETRACKING, "pipeline: only %d triangulated tracks survive, no pose", n);
  // (fewer than 8 landmarks is no fault here: the regroup leaves it in ctl->few, and the sequential sampler below draws from
  //  any population of 4 or more)
  const vo_feat B = vo_feat_seq(p->F[1 - f.fcur], (size_t)q);
  double* dR = p->d_R + (size_t)q * c.hyp * 9;
  double* dt = p->d_t + (size_t)q * c.hyp * 3;
  uint8_t* dvalid = p->d_valid + (size_t)q * c.hyp;
  int32_t* dcounts = p->d_counts + (size_t)q * c.hyp;
  uint64_t* dmasks = p->d_masks + (size_t)q * c.hyp * p->words;
  uint64_t* dbest = p->d_best_mask + (size_t)q * p->words;
  vo_ransac_state rs;
  // (a step that had walked some batches on the device before it met this fault is redone from its start: the fields
  //  the estimator object held then, and the host's generator, which follows closed steps only)
  rs.outlier_ratio = h.cont > 0 ? h.outlier_ratio0 : h.outlier_ratio;
  rs.confidence = c.ransac_confidence;
  rs.max_iterations = c.ransac_max_iterations;
  rs.n_iterations = h.cont > 0 ? h.n_iterations0 : h.n_iterations;
  rs.s = 4;
  rs.adaptive = 1;
  vo_pcg64 g = p->rng[q];
  std::vector<int32_t> samples((size_t)4 * c.hyp), counts(c.hyp);
  std::vector<uint8_t> valid(c.hyp);
  int64_t n_done = 0;
  int32_t best_count = -1, best_idx = -1;
  int total_consumed = 0, finished = 0, batches = 0, hyp_valid = 0;
  double best_pose[12];
  while (!finished) {
    VO_TRY(vo_rng_choice(&g, n, 4, c.hyp, samples.data()));
    VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_samples, samples.data(), samples.size() * 4, hipMemcpyHostToDevice, st));
    VO_TRY(vo_p3p_hypotheses_dev(ctx, B.land, B.kp64, n, c.K, p->d_samples, c.hyp, c.p3p_thr_sq, dR, dt, dvalid, dcounts,
                                 dmasks));
    VO_HIP_TRY(ctx, hipMemcpyAsync(valid.data(), dvalid, (size_t)c.hyp, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipMemcpyAsync(counts.data(), dcounts, (size_t)c.hyp * 4, hipMemcpyDeviceToHost, st));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
    int consumed = 0;
    const int32_t before = best_idx;
    VO_TRY(vo_ransac_replay(&rs, valid.data(), counts.data(), c.hyp, n, &n_done, &best_count, &best_idx, batches * c.hyp,
                            &consumed, &finished));
    for (int i = 0; i < consumed; ++i) hyp_valid += valid[i] ? 1 : 0;
    total_consumed += consumed;
    if (best_idx != before) {
      // the winner so far lives in this batch: take its pose and mask row before the buffers are reused
      // (vo_p3p_hypotheses_dev packs mask rows with ceil(n / 64) words)
      const int local = best_idx - batches * c.hyp;
      VO_HIP_TRY(ctx, mcpy(st, best_pose, dR + (size_t)local * 9, 72, hipMemcpyDeviceToHost));
      VO_HIP_TRY(ctx, mcpy(st, best_pose + 9, dt + (size_t)local * 3, 24, hipMemcpyDeviceToHost));
      VO_HIP_TRY(ctx, mcpy(st, dbest, dmasks + (size_t)local * vo_cdiv(n, 64), (size_t)vo_cdiv(n, 64) * 8,
                           hipMemcpyDeviceToDevice));
    }
    if (++batches > 64 && !finished)
      return vo_set_error(ctx, VO_ETRACKING, "pipeline: the RANSAC rule is not done after %d samples", batches * c.hyp);
  }
  if (best_idx < 0) return vo_set_error(ctx, VO_ETRACKING, "pipeline: no hypothesis had a solution");
  // the generator moves by exactly the samples the reference loop drew; the look-ahead restarts behind it
  {
    std::vector<int32_t> tmp((size_t)4 * (total_consumed > 0 ? total_consumed : 1));
    VO_TRY(vo_rng_choice(&p->rng[q], n, 4, total_consumed, tmp.data()));
  }
  p->raw_gen[q] = p->rng[q];
  p->pos_known[q] = p->gen_upto[q];
  p->pos_dev[q] = p->gen_upto[q];
  h.fault = 0;
  h.few = 0;
  h.cont = 0;
  h.n_p3p = n;
  h.n_iterations = rs.n_iterations;
  h.outlier_ratio = rs.outlier_ratio;
  h.raw_pos = p->gen_upto[q];
  h.best_idx = best_idx;
  h.best_count = best_count;
  h.consumed = total_consumed;
  h.hyp_valid = hyp_valid;
  h.n_done = n_done;
  h.n_cand = h.n_dropped = h.n_land = h.done = 0;
  memcpy(h.best_pose, best_pose, 96);
  VO_HIP_TRY(ctx, mcpy(st, ctl, &h, sizeof(h), hipMemcpyHostToDevice));
  const unsigned seq = ++p->seq;           // the fault record carried the step's number: the new record gets its own
  p->slot_seq[(size_t)f.rslot * p->S + q] = seq;
  {
    vo_pose_job job = make_pose_job(p, p->F[1 - f.fcur], 0, q);
    job.tail = 1;
    job.res = p->m_res + (size_t)f.rslot * p->S + q;
    job.seq_word = p->m_seq + (size_t)f.rslot * p->S + q;
    job.seq = seq;
    VO_TRY(vo_frame_pose(ctx, job, 1));
  }
  VO_TRY(wait_record(p, f.rslot, q, seq, 0, out));
  out->recovered = 1;
  out->reserved = fault_reason;          // (why the step left the device-only path: VO_FAULT_* bits)
  ++p->n_recovered;
  return VO_OK;
}

// Sequence q's step of flight f is open: its RANSAC loop has walked the launch's `hyp` samples and wants more
// (VO_FAULT_CONTINUE; the loop's state is in the control block, the generator position moved on).  The next batch --
// hypotheses + pose kernel for that sequence alone -- is launched until the record is a closed step's or a real fault's.
// Nothing is recomputed and nothing comes back but the records: the loop stays on the device (ransac.py:90-121 with
// max_iterations beyond one launch, as src/main.py:194-201 configures it).
static int continue_step(vo_pipeline* p, const vo_pipeline::flight_t& f, int q, vo_step_result* out) {
  vo_ctx* ctx = p->ctx;
  for (long round = 0; out->fault == VO_FAULT_CONTINUE; ++round) {
    if (round >= (1 << 16))
      return vo_set_error(ctx, VO_ETRACKING, "pipeline: the RANSAC rule is not done after %ld batches of %d samples", round, p->cfg.hyp);
    p->pos_dev[q] = out->raw_pos;
    VO_TRY(ensure_raws(p, q));
    const unsigned seq = ++p->seq;
    p->slot_seq[(size_t)f.rslot * p->S + q] = seq;
    hipLaunchKernelGGL(ctl_resume_kernel, dim3(1), dim3(1), 0, ctx->stream, p->d_ctl + q);
    VO_TRY(vo_check_launch(ctx, "ctl_resume_kernel"));
    VO_TRY(enqueue_pose_half(p, f, q, 1, seq));
    VO_TRY(wait_record(p, f.rslot, q, seq, out->raw_pos, out));
    ++p->n_continued;
  }
  return VO_OK;
}

extern "C" {

int vo_pipeline_collect_all(vo_pipeline* p, vo_step_result* outs) {
  if (!p || !outs) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->n_flight > 0, "pipeline_collect: nothing submitted");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const vo_pipeline::flight_t f = p->flight[0];
  if (p->cfg.tracker_mode != 0) VO_TRY(sift_flush_chains(p, 0));
  {
    const double t_in = now_s();
    for (int q = 0; q < p->S; ++q)
      VO_TRY(wait_record(p, f.rslot, q, p->slot_seq[(size_t)f.rslot * p->S + q], p->pos_known[q], &outs[q]));
    p->dbg_wait += now_s() - t_in;
  }
  for (int q = 0; q < p->S; ++q) {
    vo_step_result* out = &outs[q];
    const bool was_open = out->fault == VO_FAULT_CONTINUE;
    int rc = was_open ? continue_step(p, f, q, out) : VO_OK;
    if (rc != VO_OK) {
      p->n_flight = 0;
      return rc;
    }
    if (out->fault || was_open) {
      p->gate_resync = true;             // (what is enqueued again below is ordered by events, and so is the next submit)
      if (p->cfg.tracker_mode != 0) {    // (every flight has its chain before any is enqueued again)
        rc = sift_flush_chains(p, p->n_flight - 1);
        if (rc != VO_OK) {
          p->n_flight = 0;
          return rc;
        }
      }
      rc = out->fault ? recover_step(p, f, q, out) : VO_OK;
      // steps submitted behind it saw the fault and did nothing for this sequence: their main-stream chains are
      // enqueued again for it alone (pyramids and detections are done and still in place)
      for (int k = 1; rc == VO_OK && k < p->n_flight; ++k) {
        const unsigned seq = ++p->seq;
        p->slot_seq[(size_t)p->flight[k].rslot * p->S + q] = seq;
        rc = ensure_raws(p, q);
        if (p->cfg.tracker_mode != 0) {
          if (rc == VO_OK) rc = enqueue_chain_sift(p, p->flight[k], false, seq);
          continue;
        }
        if (rc == VO_OK) rc = enqueue_tracker(p, p->flight[k], false, q, 1);
        if (rc == VO_OK) rc = enqueue_chain(p, p->flight[k], false, 0, q, 1, seq);
      }
      if (rc != VO_OK) {
        // the pipeline cannot go on from here: drop what was in flight so the caller can reset the state
        p->n_flight = 0;
        return rc;
      }
    }
    if (!out->recovered) {
      // the estimator's generator follows the device: 7 outputs per consumed sample
      const uint64_t delta = out->raw_pos - p->pos_known[q];
      if (delta > 0) {
        std::vector<uint32_t> tmp((size_t)delta);
        vo_rng_raw32(&p->rng[q], (int)delta, tmp.data());
      }
      p->pos_known[q] = out->raw_pos;
      p->pos_dev[q] = out->raw_pos;
    }
  }
  p->flight[0] = p->flight[1];
  --p->n_flight;
  p->last_fbuf = 1 - f.fcur;
  return VO_OK;
}

int vo_pipeline_collect(vo_pipeline* p, vo_step_result* out) {
  if (!p || !out) return VO_EINVAL;
  if (p->S == 1) return vo_pipeline_collect_all(p, out);
  std::vector<vo_step_result> all((size_t)p->S);
  VO_TRY(vo_pipeline_collect_all(p, all.data()));
  *out = all[0];
  return VO_OK;
}

int vo_pipeline_step(vo_pipeline* p, int prev_idx, int next_idx, vo_step_result* out) {
  if (!p || !out) return VO_EINVAL;
  VO_REQUIRE(p->ctx, p->n_flight == 0, "pipeline_step: %d submitted step(s) not collected", p->n_flight);
  VO_TRY(vo_pipeline_submit(p, prev_idx, next_idx));
  return vo_pipeline_collect(p, out);
}

int vo_pipeline_bookkeeping(vo_pipeline* p, int phases, const double* new_kp, int n2, const int32_t* pairs, int M,
                            const double* T_wc, const double* T_cw, const uint8_t* p3p_inliers) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_bookkeeping: %d submitted step(s) not collected", p->n_flight);
  VO_REQUIRE(ctx, phases >= 1 && phases <= 3, "pipeline_bookkeeping: phases must be 1, 2 or 3");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (phases & 1) {
    VO_REQUIRE(ctx, new_kp && pairs && T_wc && T_cw && n2 >= 0 && n2 <= p->cap && M >= 0 && M <= n2,
               "pipeline_bookkeeping: bad arguments");
    for (int k = 0; k < M; ++k)
      VO_REQUIRE(ctx, pairs[2 * k] >= 0 && pairs[2 * k] < p->cap && pairs[2 * k + 1] >= 0 && pairs[2 * k + 1] < n2,
                 "pipeline_bookkeeping: pair %d = (%d, %d) is out of range", k, (int)pairs[2 * k], (int)pairs[2 * k + 1]);
    VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_newkp, new_kp, (size_t)n2 * 16, hipMemcpyHostToDevice, st));
    VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_pairs, pairs, (size_t)M * 8, hipMemcpyHostToDevice, st));
    VO_TRY(vo_state_regroup_pairs(ctx, p->d_ctl, p->F[p->cur], p->F[1 - p->cur], p->d_pairs, M, p->d_newkp, n2, p->cap));
    VO_HIP_TRY(ctx, hipStreamSynchronize(st));
    p->cur = 1 - p->cur;
    vo_seq_ctl h;
    VO_HIP_TRY(ctx, mcpy(st, &h, p->d_ctl, sizeof(h), hipMemcpyDeviceToHost));
    memcpy(h.T_in_wc, T_wc, 96);
    memcpy(h.T_in_cw, T_cw, 96);
    h.n_cand = h.n_dropped = h.n_land = h.done = 0;
    VO_HIP_TRY(ctx, mcpy(st, p->d_ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    std::vector<uint64_t> bits((size_t)p->words, ~0ull);
    if (p3p_inliers)
      for (int i = 0; i < h.n_tri; ++i)
        if (!p3p_inliers[i]) bits[i >> 6] &= ~(1ull << (i & 63));
    VO_HIP_TRY(ctx, mcpy(st, p->d_best_mask, bits.data(), bits.size() * 8, hipMemcpyHostToDevice));
  }
  if (phases & 1)
    VO_TRY(vo_state_candidates(ctx, p->d_ctl, p->F[p->cur], p->d_best_mask, p->cam, p->cfg.bearing_threshold, -1, p->cap));
  if (phases & 2)
    VO_TRY(vo_state_landmarks(ctx, p->d_ctl, p->F[p->cur], p->cam, -1, p->cap, nullptr, nullptr, 0u));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_pipeline_export_state_post_seq(vo_pipeline* p, int seq, const vo_step_result* r, int cap, double* d_record) {
  if (!p || !r || !d_record) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, cap >= 0 && seq >= 0 && seq < p->S, "pipeline_export_state: bad capacity or sequence index");
  pose17 h;
  for (int row = 0; row < 3; ++row) {
    for (int c = 0; c < 3; ++c) h.v[4 * row + c] = r->R_refined[3 * row + c];
    h.v[4 * row + 3] = r->t_refined[row];
  }
  h.v[12] = h.v[13] = h.v[14] = 0.0;
  h.v[15] = 1.0;
  const int n = r->best_index >= 0 ? (r->n_triangulated < cap ? r->n_triangulated : cap) : 0;
  h.v[16] = (double)n;
  // The features of the step collected last stay in their buffer until the step after next is submitted
  // (a step in flight only reads them), so the record can be queued behind whatever the main stream holds.
  const int threads = n * 3 > 17 ? n * 3 : 17;
  {
    vo_prof_scope ps(ctx, VO_K_EXPORT);
    hipLaunchKernelGGL(export_state_kernel, dim3(vo_cdiv(threads, 256)), dim3(256), 0, ctx->stream, h,
                       vo_feat_seq(p->F[p->last_fbuf], (size_t)seq).land, n, cap, d_record);
  }
  return vo_check_launch(ctx, "export_state_kernel");
}

int vo_pipeline_export_state_post(vo_pipeline* p, const vo_step_result* r, int cap, double* d_record) {
  return vo_pipeline_export_state_post_seq(p, 0, r, cap, d_record);
}

int vo_pipeline_export_state_join(vo_pipeline* p, void* consumer) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t use = consumer ? (hipStream_t)consumer : ctx->stream;
  hipStream_t st = ctx->stream;
  if (use == st) return VO_OK;
  VO_HIP_TRY(ctx, hipEventRecord(p->evB, st));
  VO_HIP_TRY(ctx, hipStreamWaitEvent(use, p->evB, 0));   // consumer: behind the records
  VO_HIP_TRY(ctx, hipEventRecord(p->evA, use));
  VO_HIP_TRY(ctx, hipStreamWaitEvent(st, p->evA, 0));    // later records: behind what the consumer holds so far
  return VO_OK;
}

// per-kernel event times accumulated over all of the pipeline's streams
int vo_pipeline_prof_read(vo_pipeline* p, int kernel_id, double* total_ms, int64_t* launches) {
  if (!p) return VO_EINVAL;
  VO_TRY(worker_idle(p));
  double sum = 0;
  int64_t n = 0;
  for (vo_ctx* q : {p->ctx, p->det, p->trk}) {
    double ms = 0;
    int64_t k = 0;
    const int rc = vo_prof_read(q, kernel_id, &ms, &k);
    if (rc != VO_OK) return q == p->ctx ? rc : vo_set_error(p->ctx, rc, "%s", vo_last_error(q));
    sum += ms;
    n += k;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = n;
  return VO_OK;
}

int vo_pipeline_prof_reset(vo_pipeline* p) {
  if (!p) return VO_EINVAL;
  VO_TRY(worker_idle(p));
  for (vo_ctx* q : {p->ctx, p->det, p->trk}) VO_TRY(vo_prof_reset(q));
  return VO_OK;
}

}  // extern "C"
