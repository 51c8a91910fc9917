// Device-resident per-frame front-end (vo_pipeline_*): the call sequence of the
// reference driver's steady-state loop (src/main.py:248-286) with every array kept in
// HBM between stages.  See include/vo_hip.h for the stage list.
//
// Plan of one step (two branches, one host wait):
//   tracking:  pyramid(next) -> KLT -> gather/compact -> P3P solve -> P3P score -> mirror
//          enqueued back to back: the solve kernel derives its samples on the device from raw
//          PCG64 outputs the host wrote to mapped memory beforehand, and reads the tracked
//          count the gather left in HBM, so nothing waits for the host.
//   detection: DLT of the previous step -> Harris response + NMS on next (feeds the next step)
//   host:  spins on a sequence word the mirror kernel publishes in mapped memory, replays the
//          sequential RANSAC rule over (valid, count), takes the winning pose.
// The two branches are enqueued by two host threads (the pipeline owns a worker for the
// detection branch): with ~16 launches per step the host's launch cost, not the GPU, bounded
// the step when one thread issued them all.  (A captured hipGraph per step was tried and is no
// faster to launch on this runtime than the individual kernels.)
#include <time.h>

#include <atomic>
#include <thread>

#include "vo_internal.h"

#pragma clang fp contract(off)

struct vo_pipeline {
  vo_ctx* ctx = nullptr;
  vo_ctx* det = nullptr;            // second context (own stream + NMS workspace): detection runs beside tracking
  vo_ctx* det2 = nullptr;           // a second detection stream: with a frame of look-ahead consecutive detections overlap
  vo_ctx* pyr = nullptr;            // stream the next frame's pyramid is built on while a step is in flight
  vo_ctx* tri = nullptr;            // stream of the DLT (independent of the detection it used to queue behind)
  vo_ctx* redo = nullptr;           // another stream: hypothesis batches of the sequential sampler (rare) must not queue
                                    // behind a step submitted later, whose solve kernel waits for this step's outcome
  hipEvent_t evDet[2] = {nullptr, nullptr};   // keypoints ready; steps alternate, so the wait for the last step's
                                              // event cannot catch this step's record
  int ev_last = 0;                            // index of the event the latest detection records
  hipEvent_t evDlt[3] = {nullptr, nullptr, nullptr};   // the DLT of track set s has run (its inputs may be overwritten, its output read)
  // detection worker: a mailbox the main thread posts (frame, buffers) to; it enqueues the branch
  // on det->stream and records evDet[ev]
  std::thread worker;
  std::atomic<unsigned> job_posted{0}, job_done{0};
  std::atomic<bool> quit{false};
  struct job_t { int kind, frame, slot, ev, which; };   // kind 0: detection of `frame` into keypoint buffer `slot`; 1: DLT of track set `slot`;
                                                        // 2: shared-map record of track set `slot` (behind its DLT)
  job_t jobs[8];
  struct export_t { double head[17]; int n, cap; double* rec; };   // payload of a kind-2 job (same ring index)
  export_t exports[8];
  unsigned last_det_job = 0;         // job count after the latest detection post (the worker has enqueued it once job_done reaches it)
  unsigned dlt_job[3] = {0, 0, 0};   // same for the latest DLT of each track set
  int job_rc = 0;
  int job_which = 0;                 // whose error text goes with job_rc: 0 det, 1 det2, 2 tri
  bool det_warm = false;
  vo_pipeline_config cfg;
  int n_levels = 1;
  size_t pyr_bytes = 0;
  // resident stream
  std::vector<uint8_t*> d_img;
  std::vector<float*> d_depth;
  std::vector<double> T_wc;          // n_frames * 16 (camera -> world)
  double* d_T_wc = nullptr;          // same, on the device
  // per-step state (double-buffered where the next step reads the previous one's output)
  // Pyramids and keypoints of a frame live in slot (frame count mod 3): with a step in flight the next
  // frame's pyramid and detection run beside it on their own streams, so they must not land in
  // a buffer the step in flight still reads.
  uint8_t* d_pyr[3] = {nullptr, nullptr, nullptr};
  double* d_kp[3] = {nullptr, nullptr, nullptr};
  int cur = 0;                       // slot holding `prev`'s pyramid / keypoints
  int det_flip = 0;                  // detections alternate between two streams (det, det2)
  hipEvent_t evPyr[3] = {nullptr, nullptr, nullptr};   // pyramid of a slot built (when built off the main stream)
  int prev_frame = -1;
  double* d_scores = nullptr;
  double* d_scores2 = nullptr;       // score map of the second detection stream
  double* d_land_all[3] = {nullptr, nullptr, nullptr};   // landmark of every keypoint of the slot's frame (N x 3)
  float *d_kp_f32[3] = {nullptr, nullptr, nullptr}, *d_next_f32 = nullptr, *d_err = nullptr;   // d_kp as float pairs
  uint8_t* d_status = nullptr;
  // compacted tracks, two sets: the deferred DLT of step k reads set k&1 while step k+1 fills the other
  // Track sets (compacted pairs + landmarks, triangulated points, DLT cameras) rotate over THREE slots: the
  // DLT of step k reads set k mod 3 on its own stream after step k has been collected, and the first step
  // that writes that set again is k + 3, submitted a whole step later -- by then the DLT has long run
  // (checked: its job must have been enqueued and its event is asked).  With two sets the writer was the
  // step submitted right after the DLT was posted.
  double *d_prev_c[3] = {nullptr, nullptr, nullptr}, *d_next_c[3] = {nullptr, nullptr, nullptr},
         *d_land_c[3] = {nullptr, nullptr, nullptr};
  int tset = 0;                      // track set of the last submitted step
  int tset_collected = 0;            // ... of the last collected step
  int dlt_n = 0;                     // tracked pairs of the last collected step (the DLT's point count)
  // Everything a step's hypotheses produce exists twice ("slot" = its track set, alternating):
  // a step may be submitted while the previous one's results are still being read.
  double* d_tri = nullptr;           // 3 x N x 3
  int cset = 0;                      // set of the last submitted step
  int32_t* d_ntracked = nullptr;     // 2 x 8: [0] tracked count, [1] mirror arrival counter, [2] sampler flag
  double *d_R = nullptr, *d_t = nullptr;
  uint8_t* d_valid = nullptr;
  int32_t* d_counts = nullptr;
  uint64_t* d_masks = nullptr;
  // steps submitted and not yet collected (at most two), oldest first
  struct flight_t { int prev_idx, next_idx, slot, tslot; unsigned seq; bool raw_published; };
  flight_t flight[2];
  int n_flight = 0;
  bool dlt_unflushed = false;        // the last collected step's DLT job has not been posted
  int cset_collected = 0;            // slot of the last collected step
  // pinned host
  int32_t* h_ntracked = nullptr;     // 2 x 4
  int32_t* h_samples = nullptr;
  // look-ahead of the estimator's generator for the device-side sampler: h_raw[raw_pos ..
  // raw_fill) are its next 32-bit outputs (raw_gen = its state behind raw_fill).  A step
  // consumes 7 per sample of the sequential rule, so the tail serves the following steps and
  // the top-up happens while the GPU works, not on the way to the launches.
  uint32_t* h_raw = nullptr;
  size_t raw_cap = 0, raw_pos = 0, raw_fill = 0;
  vo_pcg64 raw_gen;
  bool raw_valid = false;
  uint8_t* h_valid = nullptr;
  int32_t* h_counts = nullptr;
  double* h_pose = nullptr;          // 12
  double *h_R = nullptr, *h_t = nullptr;     // all hypotheses' poses, written by the GPU into mapped host memory
  volatile unsigned* h_seq = nullptr;         // per slot s: [4s+1] published by the mirror kernel, [4s+2] value it shall publish;
                                              // [8+2s], [9+2s]: {tag, offset} of the slot's outputs in h_raw
  unsigned seq = 0;
  double* h_C = nullptr;             // 3 x 24 (C1, C2), one pair per track set
  double* h_ref = nullptr;           // 2 x 32: [0..11] pose handed to the refinement, [16..30] its 14 outputs + tag
  double* m_ref = nullptr;
  unsigned ref_seq = 0;
  // device aliases of the mapped host buffers
  int32_t *m_ntracked = nullptr, *m_samples = nullptr, *m_counts = nullptr;
  uint32_t* m_raw = nullptr;
  unsigned* m_seq = nullptr;
  uint8_t* m_valid = nullptr;
  double *m_R = nullptr, *m_t = nullptr, *m_C = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr;
  // RANSAC object state (persists across frames like the reference's estimator)
  vo_pcg64 rng;
  vo_ransac_state rs;
  bool seeded = false;
  // last step
  int last_ntracked = 0, last_best = -1, last_words = 0;
  // VO_DEBUG_TIMING=1: host-side view of a step, printed by vo_pipeline_destroy
  long dbg_steps = 0;
  double dbg_t[4] = {0, 0, 0, 0};   // entry->enqueued, enqueued->results, results->return, return->next entry
  double dbg_last_return = 0;
};

namespace {

// Landmark of every detected keypoint of a frame, X_w = T_wc * (depth * K^-1 (x, y, 1)), computed behind
// the detection (off the tracking chain): the gather below then needs a single round trip to memory.
__global__ __launch_bounds__(256) void keypoint_landmarks_kernel(const double* __restrict__ kp, int N,
                                                                 const float* __restrict__ depth, int H, int W,
                                                                 double fx, double fy, double cx, double cy,
                                                                 const double* __restrict__ T_wc,
                                                                 double* __restrict__ land) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double x = kp[2 * i], y = kp[2 * i + 1];
  int xi = (int)x, yi = (int)y;
  xi = min(max(xi, 0), W - 1);
  yi = min(max(yi, 0), H - 1);
  const double z = (double)depth[(size_t)yi * W + xi];
  const double xc = (x - cx) / fx * z, yc = (y - cy) / fy * z;
  land[3 * i] = T_wc[0] * xc + T_wc[1] * yc + T_wc[2] * z + T_wc[3];
  land[3 * i + 1] = T_wc[4] * xc + T_wc[5] * yc + T_wc[6] * z + T_wc[7];
  land[3 * i + 2] = T_wc[8] * xc + T_wc[9] * yc + T_wc[10] * z + T_wc[11];
}

// Copies the hypotheses' (valid, count, R, t) into mapped host memory and then publishes a
// sequence number: the host polls that word, which costs far less than an event wait.
__global__ __launch_bounds__(256) void mirror_hypotheses_kernel(const uint8_t* __restrict__ valid,
                                                                const int32_t* __restrict__ counts,
                                                                const double* __restrict__ R,
                                                                const double* __restrict__ t, int hyp,
                                                                uint8_t* __restrict__ h_valid, int32_t* __restrict__ h_counts,
                                                                double* __restrict__ h_R, double* __restrict__ h_t,
                                                                unsigned* __restrict__ seq_host,
                                                                const unsigned* __restrict__ seq_expect,
                                                                unsigned* __restrict__ done,
                                                                int32_t* __restrict__ n_flag,
                                                                int32_t* __restrict__ h_n_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && n_flag) {
    h_n_flag[0] = n_flag[0];   // tracked count
    h_n_flag[2] = n_flag[2];   // sampler flag
    n_flag[2] = 0;             // (the slot's next solve kernel raises it again if it has to)
  }
  const int stride = gridDim.x * blockDim.x;
  for (int k = i; k < hyp; k += stride) {
    h_valid[k] = valid[k];
    h_counts[k] = counts[k];
  }
  if (h_R) {   // (the device-side-sampler path has the solve kernel write the poses to the host itself)
    for (int k = i; k < hyp * 9; k += stride) h_R[k] = R[k];
    for (int k = i; k < hyp * 3; k += stride) h_t[k] = t[k];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(done, 1u);
    if (prev == gridDim.x - 1) {       // last workgroup: everything above is visible to the host
      *done = 0;
      const unsigned seq = *seq_expect;   // written by the host (mapped memory) before this launch
      __threadfence_system();
      *seq_host = seq;
    }
  }
}

struct pose17 {
  double v[17];
};

// record = [T_cw 4x4 row-major | n | landmarks cap x 3]: what one rank contributes to the shared map
__global__ __launch_bounds__(256) void export_state_kernel(pose17 head, const double* __restrict__ tri, int n, int cap,
                                                           double* __restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 17) rec[i] = head.v[i];
  const int m = min(n, cap) * 3;
  if (i < m) rec[17 + i] = tri[i];
}

template <typename T>
int dev_alloc(vo_ctx* ctx, T** p, size_t count) {
  hipError_t e = hipMalloc((void**)p, count * sizeof(T) ? count * sizeof(T) : 256);
  if (e != hipSuccess) return vo_set_error(ctx, VO_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  return VO_OK;
}

template <typename T>
int pin_alloc(vo_ctx* ctx, T** p, size_t count) {
  hipError_t e = hipHostMalloc((void**)p, count * sizeof(T), hipHostMallocMapped);
  if (e != hipSuccess) return vo_set_error(ctx, VO_ENOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
  return VO_OK;
}

// Polls a word the GPU writes into mapped host memory; falls back to a stream wait if the
// value has not appeared after ~2 s (a fault would otherwise spin forever).
int spin_until(vo_ctx* ctx, volatile unsigned* word, unsigned value) {
  for (long it = 0; it < 400000000L; ++it) {
    if (*word == value) return VO_OK;
    __builtin_ia32_pause();
  }
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return *word == value ? VO_OK : vo_set_error(ctx, VO_EHIP, "pipeline: the GPU never published sequence %u", value);
}

double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

void rigid_inverse(const double* T, double* Ti) {
  // T = [R t; 0 1] -> [R^T  -R^T t]
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Ti[4 * r + c] = T[4 * c + r];
  for (int r = 0; r < 3; ++r) Ti[4 * r + 3] = -(Ti[4 * r] * T[3] + Ti[4 * r + 1] * T[7] + Ti[4 * r + 2] * T[11]);
  Ti[12] = Ti[13] = Ti[14] = 0.0;
  Ti[15] = 1.0;
}

void k_times_rt(const double* K, const double* Rt34, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      C[4 * r + c] = K[3 * r] * Rt34[c] + K[3 * r + 1] * Rt34[4 + c] + K[3 * r + 2] * Rt34[8 + c];
}

}  // namespace

static void worker_main(vo_pipeline* p);
static int worker_idle(vo_pipeline* p);
static int wait_job(vo_pipeline* p, unsigned target);

extern "C" {

int vo_klt_num_levels(int H, int W, int win, int max_level);
size_t vo_pyramid_bytes(int H, int W, int n_levels);

int vo_pipeline_create(vo_ctx* ctx, const vo_pipeline_config* cfg, vo_pipeline** out) {
  if (!ctx || !cfg || !out) return VO_EINVAL;
  *out = nullptr;
  VO_REQUIRE(ctx, cfg->H > 0 && cfg->W > 0 && cfg->n_frames >= 2, "pipeline: bad stream shape");
  VO_REQUIRE(ctx, cfg->n_keypoints >= 4 && cfg->n_keypoints <= 16384, "pipeline: n_keypoints must be in 4..16384");
  VO_REQUIRE(ctx, cfg->hyp >= 1, "pipeline: hyp must be >= 1");
  VO_REQUIRE(ctx, cfg->K[0] != 0.0 && cfg->K[4] != 0.0, "pipeline: singular intrinsics");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_pipeline* p = new (std::nothrow) vo_pipeline();
  if (!p) return VO_ENOMEM;
  p->ctx = ctx;
  p->cfg = *cfg;
  if (vo_create(ctx->device, nullptr, &p->det) != VO_OK || vo_create(ctx->device, nullptr, &p->redo) != VO_OK ||
      vo_create(ctx->device, nullptr, &p->tri) != VO_OK || vo_create(ctx->device, nullptr, &p->det2) != VO_OK ||
      vo_create(ctx->device, nullptr, &p->pyr) != VO_OK) {
    if (p->det) vo_destroy(p->det);
    if (p->redo) vo_destroy(p->redo);
    if (p->tri) vo_destroy(p->tri);
    if (p->det2) vo_destroy(p->det2);
    delete p;
    return vo_set_error(ctx, VO_EHIP, "pipeline: cannot create the detection stream");
  }
  const int N = cfg->n_keypoints, Hyp = cfg->hyp;
  const size_t px = (size_t)cfg->H * cfg->W;
  p->n_levels = vo_klt_num_levels(cfg->H, cfg->W, cfg->klt_win, cfg->klt_max_level);
  p->pyr_bytes = vo_pyramid_bytes(cfg->H, cfg->W, p->n_levels);
  p->d_img.assign(cfg->n_frames, nullptr);
  p->d_depth.assign(cfg->n_frames, nullptr);
  p->T_wc.assign((size_t)cfg->n_frames * 16, 0.0);
  int rc = VO_OK;
#define PA(expr) do { if (rc == VO_OK) rc = (expr); } while (0)
  for (int f = 0; f < cfg->n_frames; ++f) {
    PA(dev_alloc(ctx, &p->d_img[f], px));
    PA(dev_alloc(ctx, &p->d_depth[f], px));
  }
  PA(dev_alloc(ctx, &p->d_T_wc, (size_t)cfg->n_frames * 16));
  for (int k = 0; k < 3; ++k) {
    PA(dev_alloc(ctx, &p->d_pyr[k], p->pyr_bytes));
    PA(dev_alloc(ctx, &p->d_kp[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_kp_f32[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_land_all[k], (size_t)N * 3));
  }
  PA(dev_alloc(ctx, &p->d_scores, px));
  PA(dev_alloc(ctx, &p->d_scores2, px));
  PA(dev_alloc(ctx, &p->d_next_f32, (size_t)N * 2));
  PA(dev_alloc(ctx, &p->d_err, (size_t)N));
  PA(dev_alloc(ctx, &p->d_status, (size_t)N));
  for (int k = 0; k < 3; ++k) {
    PA(dev_alloc(ctx, &p->d_prev_c[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_next_c[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_land_c[k], (size_t)N * 3));
  }
  PA(dev_alloc(ctx, &p->d_tri, (size_t)3 * N * 3));
  PA(dev_alloc(ctx, &p->d_ntracked, 16));
  PA(dev_alloc(ctx, &p->d_R, (size_t)2 * Hyp * 9));
  PA(dev_alloc(ctx, &p->d_t, (size_t)2 * Hyp * 3));
  PA(dev_alloc(ctx, &p->d_valid, (size_t)2 * Hyp));
  PA(dev_alloc(ctx, &p->d_counts, (size_t)2 * Hyp));
  PA(dev_alloc(ctx, &p->d_masks, (size_t)2 * Hyp * vo_cdiv(N, 64)));
  PA(pin_alloc(ctx, &p->h_ntracked, 8));
  PA(pin_alloc(ctx, &p->h_samples, (size_t)Hyp * 4));
  p->raw_cap = (size_t)Hyp * 7 * 16;
  PA(pin_alloc(ctx, &p->h_raw, p->raw_cap));
  PA(pin_alloc(ctx, &p->h_valid, (size_t)2 * Hyp));
  PA(pin_alloc(ctx, &p->h_counts, (size_t)2 * Hyp));
  PA(pin_alloc(ctx, &p->h_pose, 12));
  PA(pin_alloc(ctx, &p->h_R, (size_t)2 * Hyp * 9));
  PA(pin_alloc(ctx, &p->h_t, (size_t)2 * Hyp * 3));
  {
    unsigned* q = nullptr;
    PA(pin_alloc(ctx, &q, 16));
    if (q) memset(q, 0, 64);
    p->h_seq = q;
  }
  PA(pin_alloc(ctx, &p->h_C, 72));
  PA(pin_alloc(ctx, &p->h_ref, 64));
#define MAP(dst, src) do { if (rc == VO_OK && hipHostGetDevicePointer((void**)&(dst), (void*)(src), 0) != hipSuccess) \
    rc = vo_set_error(ctx, VO_EHIP, "hipHostGetDevicePointer failed"); } while (0)
  MAP(p->m_ntracked, p->h_ntracked);
  MAP(p->m_samples, p->h_samples);
  MAP(p->m_raw, p->h_raw);
  MAP(p->m_seq, p->h_seq);
  MAP(p->m_valid, p->h_valid);
  MAP(p->m_counts, p->h_counts);
  MAP(p->m_R, p->h_R);
  MAP(p->m_t, p->h_t);
  MAP(p->m_C, p->h_C);
  MAP(p->m_ref, p->h_ref);
#undef MAP
#undef PA
  if (rc == VO_OK && (hipEventCreateWithFlags(&p->evA, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evB, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDet[0], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDet[1], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evPyr[0], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evPyr[1], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evPyr[2], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDlt[0], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDlt[1], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDlt[2], hipEventDisableTiming) != hipSuccess))
    rc = vo_set_error(ctx, VO_EHIP, "hipEventCreate failed");
  if (rc != VO_OK) {
    vo_pipeline_destroy(p);
    return rc;
  }
  if (hipMemset(p->d_ntracked, 0, 64) != hipSuccess) {
    vo_pipeline_destroy(p);
    return vo_set_error(ctx, VO_EHIP, "pipeline: hipMemset failed");
  }
  p->rs.outlier_ratio = cfg->ransac_outlier_ratio;
  p->rs.confidence = cfg->ransac_confidence;
  p->rs.max_iterations = cfg->ransac_max_iterations;
  p->rs.s = 4;
  p->rs.adaptive = 1;
  const int64_t k0 = vo_ransac_num_iterations(p->rs.confidence, p->rs.outlier_ratio, 4);
  p->rs.n_iterations = (p->rs.max_iterations >= 0 && p->rs.max_iterations < k0) ? p->rs.max_iterations : k0;
  memset(&p->rng, 0, sizeof(p->rng));
  p->worker = std::thread(worker_main, p);
  *out = p;
  return VO_OK;
}

void vo_pipeline_destroy(vo_pipeline* p) {
  if (!p) return;
  if (p->dbg_steps > 0)
    fprintf(stderr, "[vo_pipeline] %ld steps: enqueue %.1f us, wait %.1f us, replay %.1f us, between steps %.1f us\n",
            p->dbg_steps, p->dbg_t[0] / p->dbg_steps, p->dbg_t[1] / p->dbg_steps, p->dbg_t[2] / p->dbg_steps,
            p->dbg_t[3] / p->dbg_steps);
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto q : p->d_img) (void)hipFree(q);
  for (auto q : p->d_depth) (void)hipFree(q);
  if (p->worker.joinable()) {
    p->quit.store(true, std::memory_order_release);
    p->worker.join();
  }
  void* dev[] = {p->d_T_wc, p->d_pyr[0], p->d_pyr[1], p->d_pyr[2], p->d_kp[0], p->d_kp[1], p->d_kp[2], p->d_scores,
                 p->d_scores2, p->d_kp_f32[0], p->d_kp_f32[1], p->d_kp_f32[2], p->d_land_all[0], p->d_land_all[1],
                 p->d_land_all[2],
                 p->d_next_f32, p->d_err, p->d_status, p->d_prev_c[0], p->d_next_c[0], p->d_land_c[0], p->d_prev_c[1],
                 p->d_next_c[1], p->d_land_c[1], p->d_prev_c[2], p->d_next_c[2], p->d_land_c[2], p->d_tri,
                 p->d_ntracked, p->d_R, p->d_t, p->d_valid, p->d_counts, p->d_masks};
  for (void* q : dev)
    if (q) (void)hipFree(q);
  void* pin[] = {p->h_ntracked, p->h_samples, p->h_raw, p->h_valid, p->h_counts, p->h_pose, p->h_C, p->h_ref, p->h_R, p->h_t, (void*)p->h_seq};
  for (void* q : pin)
    if (q) (void)hipHostFree(q);
  if (p->evA) (void)hipEventDestroy(p->evA);
  if (p->evB) (void)hipEventDestroy(p->evB);
  for (int k = 0; k < 2; ++k)
    if (p->evDet[k]) (void)hipEventDestroy(p->evDet[k]);
  for (int k = 0; k < 3; ++k)
    if (p->evDlt[k]) (void)hipEventDestroy(p->evDlt[k]);
  for (int k = 0; k < 3; ++k)
    if (p->evPyr[k]) (void)hipEventDestroy(p->evPyr[k]);
  if (p->det) vo_destroy(p->det);
  if (p->det2) vo_destroy(p->det2);
  if (p->pyr) vo_destroy(p->pyr);
  if (p->redo) vo_destroy(p->redo);
  if (p->tri) vo_destroy(p->tri);
  delete p;
}

int vo_pipeline_set_frame(vo_pipeline* p, int idx, const uint8_t* img, const float* depth, const double* T_wc) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames && img && depth && T_wc, "pipeline_set_frame: bad arguments");
  const size_t px = (size_t)p->cfg.H * p->cfg.W;
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_img[idx], img, px, hipMemcpyHostToDevice, ctx->stream));
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_depth[idx], depth, px * 4, hipMemcpyHostToDevice, ctx->stream));
  memcpy(&p->T_wc[(size_t)idx * 16], T_wc, 16 * sizeof(double));
  VO_HIP_TRY(ctx, hipMemcpyAsync(p->d_T_wc + (size_t)idx * 16, T_wc, 128, hipMemcpyHostToDevice, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

int vo_pipeline_seed(vo_pipeline* p, const vo_pcg64* rng) {
  if (!p || !rng) return VO_EINVAL;
  p->rng = *rng;
  p->raw_valid = false;
  p->seeded = true;
  return VO_OK;
}

// ---- launches of one step -----------------------------------------------------------------
// slot-indexed views (slot = track set of the step)
static inline double* sl_R(vo_pipeline* p, int s) { return p->d_R + (size_t)s * p->cfg.hyp * 9; }
static inline double* sl_t(vo_pipeline* p, int s) { return p->d_t + (size_t)s * p->cfg.hyp * 3; }
static inline uint8_t* sl_valid(vo_pipeline* p, int s) { return p->d_valid + (size_t)s * p->cfg.hyp; }
static inline int32_t* sl_counts(vo_pipeline* p, int s) { return p->d_counts + (size_t)s * p->cfg.hyp; }
static inline uint64_t* sl_masks(vo_pipeline* p, int s) {
  return p->d_masks + (size_t)s * p->cfg.hyp * vo_cdiv(p->cfg.n_keypoints, 64);
}
static inline int32_t* sl_nt(vo_pipeline* p, int s) { return p->d_ntracked + 8 * s; }
static inline double* sl_tri(vo_pipeline* p, int s) { return p->d_tri + (size_t)s * p->cfg.n_keypoints * 3; }

// detection of `frame` into keypoint buffer `slot`; evDet[ev]: the next step's tracker may start
static int enqueue_detection(vo_pipeline* p, int frame, int slot, int ev, int which) {
  const vo_pipeline_config& c = p->cfg;
  vo_ctx* det = which ? p->det2 : p->det;
  double* scores = which ? p->d_scores2 : p->d_scores;
  det->nms_kp_f32 = p->d_kp_f32[slot];   // the tracker's float copy of the keypoints
  int rc = vo_harris_response_dev(det, p->d_img[frame], c.H, c.W, c.harris_patch, c.harris_kappa, scores);
  if (rc == VO_OK) rc = vo_nms_keypoints_dev(det, scores, c.H, c.W, c.n_keypoints, c.nms_radius, p->d_kp[slot]);
  if (rc == VO_OK) {
    hipLaunchKernelGGL(keypoint_landmarks_kernel, dim3(vo_cdiv(c.n_keypoints, 256)), dim3(256), 0, det->stream,
                       p->d_kp[slot], c.n_keypoints, p->d_depth[frame], c.H, c.W, c.K[0], c.K[4], c.K[2], c.K[5],
                       p->d_T_wc + (size_t)frame * 16, p->d_land_all[slot]);
    rc = vo_check_launch(det, "keypoint_landmarks_kernel");
  }
  if (rc == VO_OK && hipEventRecord(p->evDet[ev], det->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) return vo_set_error(p->ctx, rc, "%s", vo_last_error(det));
  return VO_OK;
}

// DLT of track set `s` (n pairs); evDlt[s]: the set may be overwritten.  Cameras are read from mapped
// host memory (one pair per track set); the point count comes with the job: the word the solve kernel
// left in HBM belongs to the step's hypothesis slot, which the step after next writes again.
static int enqueue_dlt(vo_pipeline* p, int s, int n) {
  vo_ctx* det = p->tri;   // (own stream: the tracks are complete -- the host has collected the step -- and nothing
                          //  on the detection stream depends on it)
  int rc = vo_triangulate_dlt_dev(det, p->d_prev_c[s], p->d_next_c[s], n, p->m_C + 24 * s, 0, p->m_C + 24 * s + 12,
                                  sl_tri(p, s));
  if (rc == VO_OK && hipEventRecord(p->evDlt[s], det->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) return vo_set_error(p->ctx, rc, "%s", vo_last_error(det));
  return VO_OK;
}

// shared-map record of track set `s`, on the DLT's stream right behind it
static int enqueue_export(vo_pipeline* p, int s, const vo_pipeline::export_t& e) {
  pose17 h;
  memcpy(h.v, e.head, sizeof(h.v));
  const int threads = e.n * 3 > 17 ? e.n * 3 : 17;
  hipLaunchKernelGGL(export_state_kernel, dim3(vo_cdiv(threads, 256)), dim3(256), 0, p->tri->stream, h, sl_tri(p, s), e.n,
                     e.cap, e.rec);
  if (hipGetLastError() != hipSuccess) return vo_set_error(p->tri, VO_EHIP, "launch of export_state_kernel failed");
  return VO_OK;
}

// tracking branch (main stream): KLT -> gather -> hypotheses
static int enqueue_tracking(vo_pipeline* p, int prev_idx, int next_idx, int a, int b, int cs, int ts, unsigned raw_tag,
                            bool raw_known) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  const int N = c.n_keypoints;
  VO_TRY(vo_klt_track_dev(ctx, p->d_img[prev_idx], p->d_pyr[a], p->d_img[next_idx], p->d_pyr[b], c.H, c.W,
                          p->n_levels, p->d_kp_f32[a], N, c.klt_win, c.klt_max_iter, c.klt_eps, c.klt_min_eig,
                          p->d_next_f32, p->d_status, p->d_err));
  // the track set this step fills was the input of the DLT of three steps ago: that DLT must have been
  // enqueued (its event recorded) before the event can speak for it
  VO_TRY(wait_job(p, p->dlt_job[ts]));
  if (hipEventQuery(p->evDlt[ts]) != hipSuccess) VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evDlt[ts], 0));
  // The solve kernel selects the tracked keypoints itself (no separate gather launch on the chain) and
  // leaves the compacted arrays and the count for the kernels behind it.
  vo_track_source src;
  src.status = p->d_status;
  src.err = p->d_err;
  src.err_thr = (float)c.klt_err_threshold;
  src.N = N;
  src.kp_prev = p->d_kp[a];
  src.next_xy = p->d_next_f32;
  src.land_all = p->d_land_all[a];
  src.prev_c = p->d_prev_c[ts];
  // where the generator outputs start: known now (passed by value), or published later by the
  // collect of the step before (the kernel polls the mapped word)
  VO_TRY(vo_p3p_hypotheses_raw_dev(ctx, p->d_land_c[ts], p->d_next_c[ts], sl_nt(p, cs), N, c.K,
                                   raw_known ? p->m_raw + p->raw_pos : p->m_raw,
                                   raw_known ? (const uint32_t*)nullptr : (const uint32_t*)(p->m_seq + 8 + 2 * cs),
                                   raw_tag, c.hyp, c.p3p_thr_sq, sl_R(p, cs),
                                   sl_t(p, cs), sl_valid(p, cs), sl_counts(p, cs), sl_masks(p, cs),
                                   (uint32_t*)sl_nt(p, cs) + 2, p->m_R + (size_t)cs * c.hyp * 9,
                                   p->m_t + (size_t)cs * c.hyp * 3, &src));
  return VO_OK;
}

static int launch_mirror(vo_pipeline* p, int s, bool with_count, hipStream_t st) {
  const vo_pipeline_config& c = p->cfg;
  const size_t h = (size_t)s * c.hyp;
  // with_count: the step's own launch (poses already on the host) -> (valid, count) only, 2 workgroups
  hipLaunchKernelGGL(mirror_hypotheses_kernel, dim3(with_count ? 2 : 16), dim3(256), 0, st, sl_valid(p, s), sl_counts(p, s),
                     sl_R(p, s), sl_t(p, s), c.hyp, p->m_valid + h, p->m_counts + h,
                     with_count ? (double*)nullptr : p->m_R + h * 9, with_count ? (double*)nullptr : p->m_t + h * 3,
                     p->m_seq + 4 * s + 1, p->m_seq + 4 * s + 2, (unsigned*)sl_nt(p, s) + 1,
                     with_count ? sl_nt(p, s) : (int32_t*)nullptr, p->m_ntracked + 4 * s);
  return vo_check_launch(p->ctx, "mirror_hypotheses_kernel");
}

// ---- detection worker ---------------------------------------------------------------------
static void worker_main(vo_pipeline* p) {
  (void)hipSetDevice(p->ctx->device);
  unsigned seen = 0;
  long idle = 0;
  for (;;) {
    const unsigned posted = p->job_posted.load(std::memory_order_acquire);
    if (posted == seen) {
      if (p->quit.load(std::memory_order_acquire)) return;
      if (++idle < 200000) __builtin_ia32_pause();            // a step is ~150 us: stay hot between steps
      else std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    idle = 0;
    const vo_pipeline::job_t j = p->jobs[seen & 7];
    const int rc = j.kind == 0   ? enqueue_detection(p, j.frame, j.slot, j.ev, j.which)
                   : j.kind == 1 ? enqueue_dlt(p, j.slot, j.frame)
                                 : enqueue_export(p, j.slot, p->exports[seen & 7]);
    if (rc != VO_OK) {
      p->job_rc = rc;
      p->job_which = j.kind == 0 ? j.which : 2;   // (DLT and export both run on the tri context)
    }
    ++seen;
    p->job_done.store(seen, std::memory_order_release);
  }
}

static void post_job(vo_pipeline* p, int kind, int frame, int slot, int ev, int which) {
  for (vo_ctx* q : {p->det, p->det2, p->tri, p->pyr}) {
    q->prof_on = p->ctx->prof_on;
    q->prof_kernel = p->ctx->prof_kernel;
    q->prof_every = p->ctx->prof_every;
  }
  const unsigned n = p->job_posted.load(std::memory_order_relaxed);
  while (n - p->job_done.load(std::memory_order_acquire) >= 8) __builtin_ia32_pause();   // ring full (never in practice)
  p->jobs[n & 7] = {kind, frame, slot, ev, which};
  p->job_posted.store(n + 1, std::memory_order_release);
}

// hands the detection of `frame` to the worker; returns the index of the event it will record
static int post_detection(vo_pipeline* p, int frame, int slot) {
  p->ev_last ^= 1;
  p->det_flip ^= 1;
  post_job(p, 0, frame, slot, p->ev_last, p->det_flip);
  p->last_det_job = p->job_posted.load(std::memory_order_relaxed);
  return p->ev_last;
}

// waits (host) until the worker has enqueued the first `target` jobs, their event records included
static int wait_job(vo_pipeline* p, unsigned target) {
  while ((int)(p->job_done.load(std::memory_order_acquire) - target) < 0) __builtin_ia32_pause();
  if (p->job_rc != VO_OK) {
    const int rc = p->job_rc;
    p->job_rc = VO_OK;
    return vo_set_error(p->ctx, rc, "worker: %s",
                        vo_last_error(p->job_which == 0 ? p->det : p->job_which == 1 ? p->det2 : p->tri));
  }
  return VO_OK;
}

// waits (host) until the worker has enqueued everything it was given, its event records included
static int worker_idle(vo_pipeline* p) {
  const unsigned posted = p->job_posted.load(std::memory_order_relaxed);
  while (p->job_done.load(std::memory_order_acquire) != posted) __builtin_ia32_pause();
  if (p->job_rc != VO_OK) {
    const int rc = p->job_rc;
    p->job_rc = VO_OK;
    return vo_set_error(p->ctx, rc, "detection branch: %s",
                        vo_last_error(p->job_which == 0 ? p->det : p->job_which == 1 ? p->det2 : p->tri));
  }
  return VO_OK;
}

static int detect_join(vo_pipeline* p) {
  VO_TRY(worker_idle(p));
  VO_HIP_TRY(p->ctx, hipStreamWaitEvent(p->ctx->stream, p->evDet[p->ev_last], 0));
  return VO_OK;
}

// the DLT of the last collected step (posted lazily: fetch / export / the next collect need it)
static int flush_dlt(vo_pipeline* p) {
  if (!p->dlt_unflushed) return VO_OK;
  p->dlt_unflushed = false;
  post_job(p, 1, p->dlt_n, p->tset_collected, 0, 0);
  p->dlt_job[p->tset_collected] = p->job_posted.load(std::memory_order_relaxed);
  return VO_OK;
}

int vo_pipeline_prime(vo_pipeline* p, int idx) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames, "pipeline_prime: bad frame index");
  VO_REQUIRE(ctx, p->seeded, "pipeline_prime: call vo_pipeline_seed first");
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_prime: %d submitted step(s) not collected", p->n_flight);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  p->cur = 0;
  p->dlt_unflushed = false;
  VO_TRY(vo_pyramid_build_dev(ctx, p->d_img[idx], p->cfg.H, p->cfg.W, p->n_levels, p->d_pyr[0]));
  VO_TRY(worker_idle(p));
  post_detection(p, idx, 0);
  VO_TRY(detect_join(p));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  p->prev_frame = idx;
  return VO_OK;
}

// makes h_raw[raw_pos .. raw_pos + need) valid and tells slot s's solve kernel where it starts
static void publish_raws(vo_pipeline* p, int s, unsigned tag) {
  const size_t need = (size_t)7 * p->cfg.hyp;
  if (!p->raw_valid || p->raw_pos + 2 * need > p->raw_cap) {
    // (re)start the look-ahead at the generator's present position.  Kernels of earlier steps are
    // past their reads (their results have been collected), so the buffer may be rewritten.
    p->raw_gen = p->rng;
    p->raw_pos = p->raw_fill = 0;
    p->raw_valid = true;
  }
  if (p->raw_fill < p->raw_pos + need) {
    vo_rng_raw32(&p->raw_gen, (int)(p->raw_pos + need - p->raw_fill), p->h_raw + p->raw_fill);
    p->raw_fill = p->raw_pos + need;
  }
  // {tag, offset} as one 8-byte word: the kernel reads it with a single load
  __atomic_store_n(reinterpret_cast<volatile unsigned long long*>(p->h_seq + 8 + 2 * s),
                   (unsigned long long)tag | ((unsigned long long)p->raw_pos << 32), __ATOMIC_RELEASE);
}

int vo_pipeline_submit(vo_pipeline* p, int prev_idx, int next_idx) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  VO_REQUIRE(ctx, next_idx >= 0 && next_idx < c.n_frames, "pipeline_submit: bad frame index");
  VO_REQUIRE(ctx, prev_idx == p->prev_frame, "pipeline_submit: prev frame %d is not the frame last submitted (%d)",
             prev_idx, p->prev_frame);
  VO_REQUIRE(ctx, p->n_flight < 2, "pipeline_submit: two steps are already in flight, collect one first");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int a = p->cur, b = (p->cur + 1) % 3;
  const int cs = 1 - p->cset;                          // hypothesis slot this step fills
  const int ts = (p->tset + 1) % 3;                    // track set this step fills
  static const bool dbg = getenv("VO_DEBUG_TIMING") != nullptr;
  double t_entry = 0;
  if (dbg) {
    t_entry = now_us();
    if (p->dbg_last_return > 0) p->dbg_t[3] += t_entry - p->dbg_last_return;
  }
  const unsigned seq = ++p->seq;
  p->h_seq[4 * cs + 2] = seq;                          // the mirror kernel publishes this value when it is done
  // where the step's generator outputs start is known once every earlier step has been
  // collected; otherwise the collect of the step before publishes it (the solve kernel waits)
  const bool publish_now = p->n_flight == 0;
  if (publish_now) publish_raws(p, cs, seq);

  // ---- all launches of the step: detection from the worker thread, tracking from this one ----
  VO_TRY(wait_job(p, p->last_det_job));                // (the last detection has recorded its event; it was posted a step ago)
  const int ev_prev = p->ev_last;                      // recorded behind the last step's detection
  post_detection(p, next_idx, b);
  VO_TRY(flush_dlt(p));                                // the last collected step's DLT, behind this detection
  if (p->n_flight == 0) {
    VO_TRY(vo_pyramid_build_dev(ctx, p->d_img[next_idx], c.H, c.W, p->n_levels, p->d_pyr[b]));
  } else {
    // a step is in flight on the main stream: the pyramid need not queue behind it (slot b is not
    // one of the two that step reads), the tracker of this step waits for its event instead
    const int rc = vo_pyramid_build_dev(p->pyr, p->d_img[next_idx], c.H, c.W, p->n_levels, p->d_pyr[b]);
    if (rc != VO_OK) return vo_set_error(ctx, rc, "%s", vo_last_error(p->pyr));
    VO_HIP_TRY(ctx, hipEventRecord(p->evPyr[b], p->pyr->stream));
    VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evPyr[b], 0));
  }
  // keypoints of `prev`: usually long finished, and then no barrier goes into the queue
  if (hipEventQuery(p->evDet[ev_prev]) != hipSuccess)
    VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evDet[ev_prev], 0));
  VO_TRY(enqueue_tracking(p, prev_idx, next_idx, a, b, cs, ts, seq, publish_now));
  VO_TRY(launch_mirror(p, cs, true, ctx->stream));

  vo_pipeline::flight_t& f = p->flight[p->n_flight++];
  f.prev_idx = prev_idx;
  f.next_idx = next_idx;
  f.slot = cs;
  f.tslot = ts;
  f.seq = seq;
  f.raw_published = publish_now;
  p->cset = cs;
  p->tset = ts;
  p->cur = b;
  p->prev_frame = next_idx;
  if (dbg) p->dbg_t[0] += now_us() - t_entry;
  return VO_OK;
}

// the oldest step in flight has been replayed: drop it, and tell the step submitted after it
// where its generator outputs start
static int retire_step(vo_pipeline* p, bool raw_ok) {
  if (!raw_ok) p->raw_valid = false;
  p->flight[0] = p->flight[1];
  --p->n_flight;
  if (p->n_flight > 0 && !p->flight[0].raw_published) {
    publish_raws(p, p->flight[0].slot, p->flight[0].seq);
    p->flight[0].raw_published = true;
  }
  return VO_OK;
}

int vo_pipeline_collect(vo_pipeline* p, vo_step_result* out) {
  if (!p || !out) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  VO_REQUIRE(ctx, p->n_flight > 0, "pipeline_collect: nothing submitted");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const vo_pipeline::flight_t f = p->flight[0];
  const int N = c.n_keypoints, cs = f.slot, ts = f.tslot;
  const size_t need = (size_t)7 * c.hyp;
  const bool seq_sampler = getenv("VO_SEQ_SAMPLER") != nullptr;   // test hook: always take the sequential path
  static const bool dbg = getenv("VO_DEBUG_TIMING") != nullptr;
  memset(out, 0, sizeof(*out));
  out->best_index = -1;
  out->refine_iterations = -1;
  VO_TRY(flush_dlt(p));                                // (of the step collected before this one)
  const double t_wait = dbg ? now_us() : 0;

  // while the GPU works: extend the look-ahead so the next steps find their outputs ready
  if (p->raw_valid && f.raw_published && p->raw_fill < p->raw_pos + 2 * need && p->raw_pos + 2 * need <= p->raw_cap) {
    vo_rng_raw32(&p->raw_gen, (int)(p->raw_pos + 2 * need - p->raw_fill), p->h_raw + p->raw_fill);
    p->raw_fill = p->raw_pos + 2 * need;
  }
  unsigned seq_b = f.seq;
  VO_TRY(spin_until(ctx, p->h_seq + 4 * cs + 1, seq_b));
  const double t_res = dbg ? now_us() : 0;
  const int32_t* h_nt = p->h_ntracked + 4 * cs;
  const int n = ((volatile const int32_t*)h_nt)[0];
  const bool redo = ((volatile const int32_t*)h_nt)[2] != 0 || seq_sampler;
  const uint8_t* h_valid = p->h_valid + (size_t)cs * c.hyp;
  const int32_t* h_counts = p->h_counts + (size_t)cs * c.hyp;
  const double* h_R = p->h_R + (size_t)cs * c.hyp * 9;
  const double* h_t = p->h_t + (size_t)cs * c.hyp * 3;
  out->n_tracked = n;
  p->last_ntracked = n;
  p->last_best = -1;
  p->last_words = vo_cdiv(N, 64);
  p->cset_collected = cs;
  p->tset_collected = ts;
  p->dlt_n = n;
  bool raw_ok = true;                                  // the look-ahead is still aligned with the generator

  if (n >= 4) {
    // ---- sequential RANSAC rule replayed on the host over the bulk (valid, count) ----
    int64_t n_done = 0;
    int32_t best_count = -1, best_idx = -1;
    int total_consumed = 0, finished = 0, batches = 0, hyp_valid = 0;
    bool have_batch = !redo;
    int words = vo_cdiv(N, 64);
    // generator copy for batches drawn by the sequential sampler: the first batch again if a draw
    // may have been rejected (or n < 8), later batches if the rule is not done after c.hyp samples
    vo_pcg64 g = p->rng;
    while (!finished) {
      if (!have_batch) {
        if (batches == 1 && !redo) {   // skip what the device-side batch consumed
          std::vector<int32_t> skip((size_t)4 * c.hyp);
          VO_TRY(vo_rng_choice(&g, n, 4, c.hyp, skip.data()));
        }
        VO_TRY(vo_rng_choice(&g, n, 4, c.hyp, p->h_samples));
        seq_b = ++p->seq;
        p->h_seq[4 * cs + 2] = seq_b;
        // On its own stream: the inputs are complete (this step's results were seen), every buffer
        // belongs to this step's slot, and the main stream may already hold the next step, whose
        // solve kernel waits for what this collect publishes.
        {
          const int rc = vo_p3p_hypotheses_dev(p->redo, p->d_land_c[ts], p->d_next_c[ts], n, c.K, p->m_samples, c.hyp,
                                               c.p3p_thr_sq, sl_R(p, cs), sl_t(p, cs), sl_valid(p, cs), sl_counts(p, cs),
                                               sl_masks(p, cs));
          if (rc != VO_OK) return vo_set_error(ctx, rc, "%s", vo_last_error(p->redo));
        }
        VO_TRY(launch_mirror(p, cs, false, p->redo->stream));
        VO_TRY(spin_until(ctx, p->h_seq + 4 * cs + 1, seq_b));
        words = vo_cdiv(n, 64);
      }
      have_batch = false;
      int consumed = 0;
      const int32_t before = best_idx;
      VO_TRY(vo_ransac_replay(&p->rs, h_valid, h_counts, c.hyp, n, &n_done, &best_count, &best_idx, batches * c.hyp,
                              &consumed, &finished));
      for (int i = 0; i < c.hyp; ++i) hyp_valid += h_valid[i] ? 1 : 0;
      total_consumed += consumed;
      if (best_idx != before) {
        // the winner so far lives in this batch: take its pose before the buffers are reused
        const int local = best_idx - batches * c.hyp;
        memcpy(out->R, h_R + (size_t)local * 9, 72);
        memcpy(out->t, h_t + (size_t)local * 3, 24);
        p->last_best = local;
        p->last_words = words;
      } else if (batches > 0) {
        p->last_best = -1;   // winner's mask row was overwritten by a later batch
      }
      ++batches;
      if (batches > 64) break;   // safety: the reference would still be looping
    }
    // advance the real generator by exactly the draws the reference loop consumed
    {
      std::vector<int32_t> tmp((size_t)4 * (total_consumed > 0 ? total_consumed : 1));
      VO_TRY(vo_rng_choice(&p->rng, n, 4, total_consumed, tmp.data()));
    }
    // the look-ahead moves with the generator: 7 outputs per consumed sample when every draw
    // was accepted at once (no flag) and only the device-side batch was used.  The step submitted
    // after this one is told where its outputs start now, before the refinement is waited for.
    if (!redo && batches == 1 && f.raw_published) p->raw_pos += (size_t)7 * total_consumed;
    else raw_ok = false;
    VO_TRY(retire_step(p, raw_ok));
    // ---- refinement of the accepted pose over its inliers (p3p.py:188-213), on the spare stream ----
    memcpy(out->R_refined, out->R, 72);
    memcpy(out->t_refined, out->t, 24);
    out->refine_iterations = -1;
    if (c.refine_iters > 0 && best_idx >= 0 && p->last_best >= 0) {
      double* h = p->h_ref + 32 * cs;
      memcpy(h, out->R, 72);
      memcpy(h + 9, out->t, 24);
      const unsigned tag = ++p->ref_seq;
      ((volatile double*)h)[30] = 0.0;
      p->redo->prof_on = ctx->prof_on;
      p->redo->prof_kernel = ctx->prof_kernel;
      p->redo->prof_every = ctx->prof_every;
      const int rc = vo_refine_pose_ndev(p->redo, p->d_land_c[ts], p->d_next_c[ts], N, sl_nt(p, cs), c.K, nullptr,
                                         sl_masks(p, cs) + (size_t)p->last_best * p->last_words, p->m_ref + 32 * cs,
                                         c.refine_iters, p->m_ref + 32 * cs + 16, tag);
      if (rc != VO_OK) return vo_set_error(ctx, rc, "%s", vo_last_error(p->redo));
      long spins = 0;
      while (((volatile double*)h)[30] != (double)tag) {
        __builtin_ia32_pause();
        if (++spins > 400000000L) {
          VO_HIP_TRY(ctx, hipStreamSynchronize(p->redo->stream));
          break;
        }
      }
      memcpy(out->R_refined, h + 16, 72);
      memcpy(out->t_refined, h + 25, 24);
      out->refine_iterations = (int32_t)h[28];
      out->refine_cost = h[29];
    }
    out->n_inliers = best_count > 0 ? best_count : 0;
    out->best_index = best_idx;
    out->ransac_iterations = n_done;
    out->draws_consumed = total_consumed;
    out->hyp_valid = hyp_valid;

    // ---- cameras for the DLT of the tracked pairs: C1 = K T_cw(prev) (stream pose), C2 = K [R | t] ----
    if (best_idx >= 0) {
      double Tcw[16], Rt[12];
      double* hC = p->h_C + 24 * ts;
      rigid_inverse(&p->T_wc[(size_t)f.prev_idx * 16], Tcw);
      k_times_rt(c.K, Tcw, hC);
      for (int r = 0; r < 3; ++r) {
        Rt[4 * r] = out->R_refined[3 * r];         // (= R, t when the refinement is off)
        Rt[4 * r + 1] = out->R_refined[3 * r + 1];
        Rt[4 * r + 2] = out->R_refined[3 * r + 2];
        Rt[4 * r + 3] = out->t_refined[r];
      }
      k_times_rt(c.K, Rt, hC + 12);
      p->dlt_unflushed = true;
      // posted now, not with the next submit: the step that reuses this track set checks that the DLT
      // has been enqueued before it asks its event (enqueue_tracking), and the worker has it out of the
      // way before the next detection arrives
      VO_TRY(flush_dlt(p));
    }
  }
  if (n < 4) VO_TRY(retire_step(p, true));
  if (dbg) {
    const double t_ret = now_us();
    p->dbg_t[1] += t_res - t_wait;
    p->dbg_t[2] += t_ret - t_res;
    p->dbg_last_return = t_ret;
    ++p->dbg_steps;
  }
  return VO_OK;
}

int vo_pipeline_step(vo_pipeline* p, int prev_idx, int next_idx, vo_step_result* out) {
  if (!p || !out) return VO_EINVAL;
  VO_REQUIRE(p->ctx, p->n_flight == 0, "pipeline_step: %d submitted step(s) not collected", p->n_flight);
  VO_TRY(vo_pipeline_submit(p, prev_idx, next_idx));
  return vo_pipeline_collect(p, out);
}

int vo_pipeline_export_state_post(vo_pipeline* p, const vo_step_result* r, int cap, double* d_record) {
  if (!p || !r || !d_record) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, cap >= 0, "pipeline_export_state: bad capacity");
  VO_TRY(flush_dlt(p));                                  // the step's DLT job first: the record job follows it in the FIFO
  const unsigned slot = p->job_posted.load(std::memory_order_relaxed);
  while (slot - p->job_done.load(std::memory_order_acquire) >= 8) __builtin_ia32_pause();
  vo_pipeline::export_t& e = p->exports[slot & 7];
  for (int row = 0; row < 3; ++row) {
    for (int c = 0; c < 3; ++c) e.head[4 * row + c] = r->R_refined[3 * row + c];
    e.head[4 * row + 3] = r->t_refined[row];
  }
  e.head[12] = e.head[13] = e.head[14] = 0.0;
  e.head[15] = 1.0;
  e.n = r->best_index >= 0 ? (r->n_tracked < cap ? r->n_tracked : cap) : 0;
  e.head[16] = (double)e.n;
  e.cap = cap;
  e.rec = d_record;
  post_job(p, 2, 0, p->tset_collected, 0, 0);
  return VO_OK;
}

int vo_pipeline_export_state_join(vo_pipeline* p, void* consumer) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t use = consumer ? (hipStream_t)consumer : ctx->stream;
  hipStream_t st = p->tri->stream;
  VO_TRY(worker_idle(p));                                // every posted record has been enqueued; the worker is quiet
  VO_HIP_TRY(ctx, hipEventRecord(p->evB, st));
  VO_HIP_TRY(ctx, hipStreamWaitEvent(use, p->evB, 0));   // consumer: behind the records
  VO_HIP_TRY(ctx, hipEventRecord(p->evA, use));
  VO_HIP_TRY(ctx, hipStreamWaitEvent(st, p->evA, 0));    // later records: behind what the consumer holds so far
  return VO_OK;
}

int vo_pipeline_export_state_dev(vo_pipeline* p, const vo_step_result* r, int cap, double* d_record, void* consumer) {
  if (!p) return VO_EINVAL;
  // (records posted earlier and everything the consumer already holds are ordered before this one)
  VO_TRY(vo_pipeline_export_state_join(p, consumer));
  VO_TRY(vo_pipeline_export_state_post(p, r, cap, d_record));
  return vo_pipeline_export_state_join(p, consumer);
}

// per-kernel event times accumulated over all of the pipeline's streams
int vo_pipeline_prof_read(vo_pipeline* p, int kernel_id, double* total_ms, int64_t* launches) {
  if (!p) return VO_EINVAL;
  VO_TRY(worker_idle(p));
  double sum = 0;
  int64_t n = 0;
  for (vo_ctx* q : {p->ctx, p->det, p->det2, p->pyr, p->tri, p->redo}) {
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
  for (vo_ctx* q : {p->ctx, p->det, p->det2, p->pyr, p->tri, p->redo}) VO_TRY(vo_prof_reset(q));
  return VO_OK;
}

int vo_pipeline_fetch(vo_pipeline* p, double* kp_next, double* prev_xy, double* next_xy, double* landmarks,
                      double* triangulated, uint8_t* inliers) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  hipStream_t st = ctx->stream;
  VO_REQUIRE(ctx, p->n_flight == 0, "pipeline_fetch: %d submitted step(s) not collected", p->n_flight);
  const int n = p->last_ntracked, N = p->cfg.n_keypoints;
  const int cs = p->cset_collected, ts = p->tset_collected;
  VO_TRY(flush_dlt(p));
  VO_TRY(detect_join(p));
  VO_HIP_TRY(ctx, hipStreamWaitEvent(st, p->evDlt[ts], 0));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  if (kp_next) VO_HIP_TRY(ctx, hipMemcpy(kp_next, p->d_kp[p->cur], (size_t)N * 16, hipMemcpyDeviceToHost));
  if (n > 0) {
    if (prev_xy) VO_HIP_TRY(ctx, hipMemcpy(prev_xy, p->d_prev_c[ts], (size_t)n * 16, hipMemcpyDeviceToHost));
    if (next_xy) VO_HIP_TRY(ctx, hipMemcpy(next_xy, p->d_next_c[ts], (size_t)n * 16, hipMemcpyDeviceToHost));
    if (landmarks) VO_HIP_TRY(ctx, hipMemcpy(landmarks, p->d_land_c[ts], (size_t)n * 24, hipMemcpyDeviceToHost));
    if (triangulated) VO_HIP_TRY(ctx, hipMemcpy(triangulated, sl_tri(p, ts), (size_t)n * 24, hipMemcpyDeviceToHost));
    if (inliers) {
      VO_REQUIRE(ctx, p->last_best >= 0, "pipeline_fetch: no inlier mask for the last step");
      std::vector<uint64_t> row(p->last_words);
      VO_HIP_TRY(ctx, hipMemcpy(row.data(), sl_masks(p, cs) + (size_t)p->last_best * p->last_words,
                                (size_t)p->last_words * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i) inliers[i] = (uint8_t)((row[i >> 6] >> (i & 63)) & 1ull);
    }
  }
  return VO_OK;
}

}  // extern "C"
