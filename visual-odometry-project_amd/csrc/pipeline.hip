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
  hipEvent_t evDet[2] = {nullptr, nullptr};   // keypoints ready; steps alternate, so the wait for the last step's
                                              // event cannot catch this step's record
  int ev_last = 0;                            // index of the event the latest detection records
  hipEvent_t evDltDone = nullptr;             // the DLT queued behind a detection has read its inputs
  // detection worker: a mailbox the main thread posts (frame, buffers) to; it enqueues the branch
  // on det->stream and records evDet[ev]
  std::thread worker;
  std::atomic<unsigned> job_posted{0}, job_done{0};
  std::atomic<bool> quit{false};
  struct { int frame, slot, prev_set, ev; bool with_dlt; } job = {0, 0, 0, 0, false};
  int job_rc = 0;
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
  uint8_t* d_pyr[2] = {nullptr, nullptr};
  double* d_kp[2] = {nullptr, nullptr};
  int cur = 0;                       // buffer index holding `prev`'s pyramid / keypoints
  int prev_frame = -1;
  double* d_scores = nullptr;
  float *d_kp_f32[2] = {nullptr, nullptr}, *d_next_f32 = nullptr, *d_err = nullptr;   // d_kp as float pairs
  uint8_t* d_status = nullptr;
  // compacted tracks, two sets: the deferred DLT of step k reads set k&1 while step k+1 fills the other
  double *d_prev_c[2] = {nullptr, nullptr}, *d_next_c[2] = {nullptr, nullptr}, *d_land_c[2] = {nullptr, nullptr};
  double* d_tri = nullptr;
  int cset = 0;                      // set written by the last step
  bool dlt_pending = false;          // the last step's DLT has not been enqueued yet (the next step does it)
  int32_t* d_ntracked = nullptr;
  double *d_R = nullptr, *d_t = nullptr;
  uint8_t* d_valid = nullptr;
  int32_t* d_counts = nullptr;
  uint64_t* d_masks = nullptr;
  // pinned host
  int32_t* h_ntracked = nullptr;
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
  volatile unsigned* h_seq = nullptr;         // [0] tracking done, [1] hypotheses mirrored: sequence numbers the host spins on
  unsigned seq = 0;
  double* h_C = nullptr;             // 2 x 24 (C1, C2), alternating with the track sets
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

// Keeps tracks with status != 0 and err < thr in their original order (the boolean
// mask of klt.py:244-269), converts to float64 and looks the landmark of each
// previous keypoint up in the depth map:  X_w = T_wc * (depth * K^-1 (x, y, 1)).
__global__ __launch_bounds__(1024) void gather_tracks_kernel(const double* __restrict__ kp_prev,
                                                             const float* __restrict__ next_xy,
                                                             const uint8_t* __restrict__ status,
                                                             const float* __restrict__ err, int N, float err_thr,
                                                             const float* __restrict__ depth, int H, int W, double fx,
                                                             double fy, double cx, double cy,
                                                             const double* __restrict__ T_wc,
                                                             double* __restrict__ prev_c, double* __restrict__ next_c,
                                                             double* __restrict__ land_c, int32_t* __restrict__ n_out,
                                                             int cs) {
  // All loads of up to four passes (4096 keypoints) go out before anything is consumed: one
  // round trip for the tracker's outputs, one for the depth look-ups that depend on them.
  constexpr int GE = 4;
  __shared__ int s_w[GE][16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  double Tm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Tm[k] = T_wc[k];
  for (int c0 = 0; c0 < N; c0 += GE * 1024) {
    bool keep[GE];
    double x[GE], y[GE], z[GE];
    float nxv[GE], nyv[GE];
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      const int i = c0 + k * 1024 + tid;
      const bool in = i < N;
      const int ii = in ? i : 0;
      keep[k] = in && status[ii] != 0 && err[ii] < err_thr;
      x[k] = kp_prev[2 * ii];
      y[k] = kp_prev[2 * ii + 1];
      nxv[k] = next_xy[2 * ii];
      nyv[k] = next_xy[2 * ii + 1];
    }
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      int xi = (int)x[k], yi = (int)y[k];
      xi = min(max(xi, 0), W - 1);
      yi = min(max(yi, 0), H - 1);
      z[k] = (double)depth[(size_t)yi * W + xi];
    }
    unsigned long long m[GE];
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      m[k] = __ballot(keep[k]);
      if (lane == 0) s_w[k][wv] = __popcll(m[k]);
    }
    __syncthreads();
    int off = s_base;
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      int mine = off;
      for (int w = 0; w < 16; ++w) {
        if (w < wv) mine += s_w[k][w];
        off += s_w[k][w];
      }
      if (keep[k]) {
        const int o = mine + __popcll(m[k] & ((1ull << lane) - 1ull));
        prev_c[2 * o] = x[k];
        prev_c[2 * o + 1] = y[k];
        next_c[2 * o] = (double)nxv[k];
        next_c[2 * o + 1] = (double)nyv[k];
        const double xc = (x[k] - cx) / fx * z[k], yc = (y[k] - cy) / fy * z[k];
        land_c[3 * o] = Tm[0] * xc + Tm[1] * yc + Tm[2] * z[k] + Tm[3];
        land_c[3 * o + 1] = Tm[4] * xc + Tm[5] * yc + Tm[6] * z[k] + Tm[7];
        land_c[3 * o + 2] = Tm[8] * xc + Tm[9] * yc + Tm[10] * z[k] + Tm[11];
      }
    }
    __syncthreads();
    if (tid == 0) s_base = off;
    __syncthreads();
  }
  if (tid == 0) {
    n_out[0] = s_base;
    n_out[2] = 0;           // "sampler needs the sequential path" flag of the solve kernel that follows
    n_out[4 + cs] = s_base; // count of track set cs, for the DLT that runs during the next step
  }
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
                                                                const int32_t* __restrict__ n_flag,
                                                                int32_t* __restrict__ h_n_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && n_flag) {
    h_n_flag[0] = n_flag[0];   // tracked count
    h_n_flag[2] = n_flag[2];   // sampler flag
  }
  const int stride = gridDim.x * blockDim.x;
  for (int k = i; k < hyp; k += stride) {
    h_valid[k] = valid[k];
    h_counts[k] = counts[k];
  }
  for (int k = i; k < hyp * 9; k += stride) h_R[k] = R[k];
  for (int k = i; k < hyp * 3; k += stride) h_t[k] = t[k];
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
  if (vo_create(ctx->device, nullptr, &p->det) != VO_OK) {
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
  for (int k = 0; k < 2; ++k) {
    PA(dev_alloc(ctx, &p->d_pyr[k], p->pyr_bytes));
    PA(dev_alloc(ctx, &p->d_kp[k], (size_t)N * 2));
  }
  PA(dev_alloc(ctx, &p->d_scores, px));
  PA(dev_alloc(ctx, &p->d_kp_f32[0], (size_t)N * 2));
  PA(dev_alloc(ctx, &p->d_kp_f32[1], (size_t)N * 2));
  PA(dev_alloc(ctx, &p->d_next_f32, (size_t)N * 2));
  PA(dev_alloc(ctx, &p->d_err, (size_t)N));
  PA(dev_alloc(ctx, &p->d_status, (size_t)N));
  for (int k = 0; k < 2; ++k) {
    PA(dev_alloc(ctx, &p->d_prev_c[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_next_c[k], (size_t)N * 2));
    PA(dev_alloc(ctx, &p->d_land_c[k], (size_t)N * 3));
  }
  PA(dev_alloc(ctx, &p->d_tri, (size_t)N * 3));
  PA(dev_alloc(ctx, &p->d_ntracked, 8));
  PA(dev_alloc(ctx, &p->d_R, (size_t)Hyp * 9));
  PA(dev_alloc(ctx, &p->d_t, (size_t)Hyp * 3));
  PA(dev_alloc(ctx, &p->d_valid, (size_t)Hyp));
  PA(dev_alloc(ctx, &p->d_counts, (size_t)Hyp));
  PA(dev_alloc(ctx, &p->d_masks, (size_t)Hyp * vo_cdiv(N, 64)));
  PA(pin_alloc(ctx, &p->h_ntracked, 4));
  PA(pin_alloc(ctx, &p->h_samples, (size_t)Hyp * 4));
  p->raw_cap = (size_t)Hyp * 7 * 4;
  PA(pin_alloc(ctx, &p->h_raw, p->raw_cap));
  PA(pin_alloc(ctx, &p->h_valid, (size_t)Hyp));
  PA(pin_alloc(ctx, &p->h_counts, (size_t)Hyp));
  PA(pin_alloc(ctx, &p->h_pose, 12));
  PA(pin_alloc(ctx, &p->h_R, (size_t)Hyp * 9));
  PA(pin_alloc(ctx, &p->h_t, (size_t)Hyp * 3));
  {
    unsigned* q = nullptr;
    PA(pin_alloc(ctx, &q, 16));
    if (q) memset(q, 0, 64);
    p->h_seq = q;
  }
  PA(pin_alloc(ctx, &p->h_C, 48));
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
#undef MAP
#undef PA
  if (rc == VO_OK && (hipEventCreateWithFlags(&p->evA, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evB, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDet[0], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDet[1], hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&p->evDltDone, hipEventDisableTiming) != hipSuccess))
    rc = vo_set_error(ctx, VO_EHIP, "hipEventCreate failed");
  if (rc != VO_OK) {
    vo_pipeline_destroy(p);
    return rc;
  }
  if (hipMemset(p->d_ntracked, 0, 32) != hipSuccess) {
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
  void* dev[] = {p->d_T_wc, p->d_pyr[0], p->d_pyr[1], p->d_kp[0], p->d_kp[1], p->d_scores, p->d_kp_f32[0], p->d_kp_f32[1],
                 p->d_next_f32, p->d_err, p->d_status, p->d_prev_c[0], p->d_next_c[0], p->d_land_c[0], p->d_prev_c[1],
                 p->d_next_c[1], p->d_land_c[1], p->d_tri,
                 p->d_ntracked, p->d_R, p->d_t, p->d_valid, p->d_counts, p->d_masks};
  for (void* q : dev)
    if (q) (void)hipFree(q);
  void* pin[] = {p->h_ntracked, p->h_samples, p->h_raw, p->h_valid, p->h_counts, p->h_pose, p->h_C, p->h_R, p->h_t, (void*)p->h_seq};
  for (void* q : pin)
    if (q) (void)hipHostFree(q);
  if (p->evA) (void)hipEventDestroy(p->evA);
  if (p->evB) (void)hipEventDestroy(p->evB);
  for (int k = 0; k < 2; ++k)
    if (p->evDet[k]) (void)hipEventDestroy(p->evDet[k]);
  if (p->evDltDone) (void)hipEventDestroy(p->evDltDone);
  if (p->det) vo_destroy(p->det);
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

// detection branch: Harris response + NMS of `frame` (evDet[ev]: the next step's tracker may
// start), then the DLT of the previous step's tracks (evDltDone: its inputs may be overwritten)
static int enqueue_detection(vo_pipeline* p, int frame, int slot, int prev_set, bool with_dlt, int ev) {
  const vo_pipeline_config& c = p->cfg;
  vo_ctx* det = p->det;
  det->nms_kp_f32 = p->d_kp_f32[slot];   // the tracker's float copy of the keypoints
  int rc = vo_harris_response_dev(det, p->d_img[frame], c.H, c.W, c.harris_patch, c.harris_kappa, p->d_scores);
  if (rc == VO_OK) rc = vo_nms_keypoints_dev(det, p->d_scores, c.H, c.W, c.n_keypoints, c.nms_radius, p->d_kp[slot]);
  if (rc == VO_OK && hipEventRecord(p->evDet[ev], det->stream) != hipSuccess) rc = VO_EHIP;
  // cameras are read from mapped host memory (set s is rewritten two steps later at the earliest),
  // the point count from the word the gather kernel of that step left in HBM
  if (rc == VO_OK && with_dlt)
    rc = vo_triangulate_dlt_ndev(det, p->d_prev_c[prev_set], p->d_next_c[prev_set], p->d_ntracked + 4 + prev_set,
                                 c.n_keypoints, p->m_C + 24 * prev_set, p->m_C + 24 * prev_set + 12, p->d_tri);
  if (rc == VO_OK && hipEventRecord(p->evDltDone, det->stream) != hipSuccess) rc = VO_EHIP;
  if (rc != VO_OK) return vo_set_error(p->ctx, rc, "%s", vo_last_error(det));
  return VO_OK;
}

// tracking branch up to the mirror kernel (main stream)
static void post_detection(vo_pipeline* p, int frame, int slot, int prev_set, bool with_dlt);

static int enqueue_tracking(vo_pipeline* p, int prev_idx, int next_idx, int a, int b, int cs, int det_pos) {
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  const int N = c.n_keypoints;
  const double fx = c.K[0], fy = c.K[4], cx = c.K[2], cy = c.K[5];
  VO_TRY(vo_klt_track_dev(ctx, p->d_img[prev_idx], p->d_pyr[a], p->d_img[next_idx], p->d_pyr[b], c.H, c.W,
                          p->n_levels, p->d_kp_f32[a], N, c.klt_win, c.klt_max_iter, c.klt_eps, c.klt_min_eig,
                          p->d_next_f32, p->d_status, p->d_err));
  if (det_pos == 1) post_detection(p, next_idx, b, 1 - cs, p->dlt_pending);
  // the track set this gather fills was the input of the DLT queued behind the last detection
  if (hipEventQuery(p->evDltDone) != hipSuccess) VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evDltDone, 0));
  {
    vo_prof_scope ps(ctx, VO_K_GATHER);
    hipLaunchKernelGGL(gather_tracks_kernel, dim3(1), dim3(1024), 0, ctx->stream, p->d_kp[a], p->d_next_f32,
                       p->d_status, p->d_err, N, (float)c.klt_err_threshold, p->d_depth[prev_idx], c.H, c.W, fx, fy, cx,
                       cy, p->d_T_wc + (size_t)prev_idx * 16, p->d_prev_c[cs], p->d_next_c[cs], p->d_land_c[cs],
                       p->d_ntracked, cs);
  }
  VO_TRY(vo_check_launch(ctx, "gather_tracks_kernel"));
  VO_TRY(vo_p3p_hypotheses_raw_dev(ctx, p->d_land_c[cs], p->d_next_c[cs], p->d_ntracked, N, c.K, p->m_raw + p->raw_pos, c.hyp,
                                   c.p3p_thr_sq, p->d_R, p->d_t, p->d_valid, p->d_counts, p->d_masks,
                                   (uint32_t*)p->d_ntracked + 2));
  return VO_OK;
}

static int launch_mirror(vo_pipeline* p, bool with_count) {
  const vo_pipeline_config& c = p->cfg;
  hipLaunchKernelGGL(mirror_hypotheses_kernel, dim3(16), dim3(256), 0, p->ctx->stream, p->d_valid, p->d_counts, p->d_R,
                     p->d_t, c.hyp, p->m_valid, p->m_counts, p->m_R, p->m_t, p->m_seq + 1, p->m_seq + 2,
                     (unsigned*)p->d_ntracked + 1, with_count ? (const int32_t*)p->d_ntracked : (const int32_t*)nullptr,
                     p->m_ntracked);
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
    seen = posted;
    p->job_rc = enqueue_detection(p, p->job.frame, p->job.slot, p->job.prev_set, p->job.with_dlt, p->job.ev);
    p->job_done.store(seen, std::memory_order_release);
  }
}

// hands the detection branch of a step to the worker
static void post_detection(vo_pipeline* p, int frame, int slot, int prev_set, bool with_dlt) {
  p->det->prof_on = p->ctx->prof_on;
  p->det->prof_kernel = p->ctx->prof_kernel;
  p->job.frame = frame;
  p->job.slot = slot;
  p->job.prev_set = prev_set;
  p->job.with_dlt = with_dlt;
  p->ev_last ^= 1;
  p->job.ev = p->ev_last;
  p->job_posted.store(p->job_posted.load(std::memory_order_relaxed) + 1, std::memory_order_release);
}

// waits (host) until the worker has enqueued everything it was given, evDet[] included
static int worker_idle(vo_pipeline* p) {
  const unsigned posted = p->job_posted.load(std::memory_order_relaxed);
  while (p->job_done.load(std::memory_order_acquire) != posted) __builtin_ia32_pause();
  if (p->job_rc != VO_OK) {
    const int rc = p->job_rc;
    p->job_rc = VO_OK;
    return vo_set_error(p->ctx, rc, "detection branch: %s", vo_last_error(p->det));
  }
  return VO_OK;
}

static int detect_join(vo_pipeline* p) {
  VO_TRY(worker_idle(p));
  VO_HIP_TRY(p->ctx, hipStreamWaitEvent(p->ctx->stream, p->evDet[p->ev_last], 0));
  return VO_OK;
}

int vo_pipeline_prime(vo_pipeline* p, int idx) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, idx >= 0 && idx < p->cfg.n_frames, "pipeline_prime: bad frame index");
  VO_REQUIRE(ctx, p->seeded, "pipeline_prime: call vo_pipeline_seed first");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  p->cur = 0;
  p->dlt_pending = false;
  VO_TRY(vo_pyramid_build_dev(ctx, p->d_img[idx], p->cfg.H, p->cfg.W, p->n_levels, p->d_pyr[0]));
  VO_TRY(worker_idle(p));
  post_detection(p, idx, 0, 0, false);
  VO_TRY(detect_join(p));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  p->prev_frame = idx;
  return VO_OK;
}

// The DLT of the last step normally runs at the start of the next one (it is not on the path to
// the next pose); fetch / export, which read its result, run it now.
static int flush_dlt(vo_pipeline* p) {
  if (!p->dlt_pending) return VO_OK;
  vo_ctx* ctx = p->ctx;
  p->dlt_pending = false;
  VO_TRY(worker_idle(p));
  p->det->prof_on = ctx->prof_on;
  p->det->prof_kernel = ctx->prof_kernel;
  const int s = p->cset;
  const int rc = vo_triangulate_dlt_ndev(p->det, p->d_prev_c[s], p->d_next_c[s], p->d_ntracked + 4 + s,
                                         p->cfg.n_keypoints, p->m_C + 24 * s, p->m_C + 24 * s + 12, p->d_tri);
  if (rc != VO_OK) return vo_set_error(ctx, rc, "%s", vo_last_error(p->det));
  VO_HIP_TRY(ctx, hipEventRecord(p->evDltDone, p->det->stream));
  VO_HIP_TRY(ctx, hipEventRecord(p->evDet[p->ev_last], p->det->stream));
  return VO_OK;
}

int vo_pipeline_step(vo_pipeline* p, int prev_idx, int next_idx, vo_step_result* out) {
  if (!p || !out) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  const vo_pipeline_config& c = p->cfg;
  VO_REQUIRE(ctx, next_idx >= 0 && next_idx < c.n_frames, "pipeline_step: bad frame index");
  VO_REQUIRE(ctx, prev_idx == p->prev_frame, "pipeline_step: prev frame %d is not the frame last processed (%d)",
             prev_idx, p->prev_frame);
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int N = c.n_keypoints, a = p->cur, b = 1 - p->cur;
  const int cs = 1 - p->cset;                       // track set this step fills
  const bool seq_sampler = getenv("VO_SEQ_SAMPLER") != nullptr;   // test hook: always take the sequential path
  memset(out, 0, sizeof(*out));
  out->best_index = -1;
  static const bool dbg = getenv("VO_DEBUG_TIMING") != nullptr;
  auto now = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  };
  double t_entry = 0, t_enq = 0, t_res = 0;
  if (dbg) {
    t_entry = now();
    if (p->dbg_last_return > 0) p->dbg_t[3] += t_entry - p->dbg_last_return;
  }

  // Generator outputs for the device-side sampler (a look-ahead; the real generator advances
  // by what the sequential rule consumes, below).  Normally they are already there.
  const size_t need = (size_t)7 * c.hyp;
  if (!p->raw_valid || p->raw_fill < p->raw_pos + need) {
    if (!p->raw_valid) {
      p->raw_gen = p->rng;
      p->raw_pos = p->raw_fill = 0;
      p->raw_valid = true;
    }
    vo_rng_raw32(&p->raw_gen, (int)(p->raw_pos + need - p->raw_fill), p->h_raw + p->raw_fill);
    p->raw_fill = p->raw_pos + need;
  }
  const unsigned seq_b0 = ++p->seq;
  p->h_seq[2] = seq_b0;                               // the mirror kernel publishes this value when it is done

  // ---- all launches of the step: detection from the worker thread, tracking from this one ----
  VO_TRY(worker_idle(p));
  const int ev_prev = p->ev_last;                      // recorded behind the last step's detection
  static const int det_pos = getenv("VO_DET_POS") ? atoi(getenv("VO_DET_POS")) : 0;
  if (det_pos == 0) post_detection(p, next_idx, b, 1 - cs, p->dlt_pending);
  VO_TRY(vo_pyramid_build_dev(ctx, p->d_img[next_idx], c.H, c.W, p->n_levels, p->d_pyr[b]));
  // keypoints of `prev`: usually long finished, and then no barrier goes into the queue
  if (hipEventQuery(p->evDet[ev_prev]) != hipSuccess)
    VO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p->evDet[ev_prev], 0));
  VO_TRY(enqueue_tracking(p, prev_idx, next_idx, a, b, cs, det_pos));
  VO_TRY(launch_mirror(p, true));
  if (det_pos == 2) post_detection(p, next_idx, b, 1 - cs, p->dlt_pending);
  p->dlt_pending = false;
  unsigned seq_b = seq_b0;

  if (dbg) t_enq = now();
  // while the GPU works: extend the look-ahead so the next step finds its outputs ready
  if (p->raw_fill < p->raw_pos + 2 * need && p->raw_pos + 2 * need <= p->raw_cap) {
    vo_rng_raw32(&p->raw_gen, (int)(p->raw_pos + 2 * need - p->raw_fill), p->h_raw + p->raw_fill);
    p->raw_fill = p->raw_pos + 2 * need;
  }
  VO_TRY(spin_until(ctx, p->h_seq + 1, seq_b));
  if (dbg) t_res = now();
  const int n = ((volatile int32_t*)p->h_ntracked)[0];
  const bool redo = ((volatile int32_t*)p->h_ntracked)[2] != 0 || seq_sampler;
  out->n_tracked = n;
  p->last_ntracked = n;
  p->last_best = -1;
  p->last_words = vo_cdiv(N, 64);
  p->cset = cs;

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
        p->h_seq[2] = seq_b;
        VO_TRY(vo_p3p_hypotheses_dev(ctx, p->d_land_c[cs], p->d_next_c[cs], n, c.K, p->m_samples, c.hyp, c.p3p_thr_sq,
                                     p->d_R, p->d_t, p->d_valid, p->d_counts, p->d_masks));
        VO_TRY(launch_mirror(p, false));
        VO_TRY(spin_until(ctx, p->h_seq + 1, seq_b));
        words = vo_cdiv(n, 64);
      }
      have_batch = false;
      int consumed = 0;
      const int32_t before = best_idx;
      VO_TRY(vo_ransac_replay(&p->rs, p->h_valid, p->h_counts, c.hyp, n, &n_done, &best_count, &best_idx,
                              batches * c.hyp, &consumed, &finished));
      for (int i = 0; i < c.hyp; ++i) hyp_valid += p->h_valid[i] ? 1 : 0;
      total_consumed += consumed;
      if (best_idx != before) {
        // the winner so far lives in this batch: take its pose before the buffers are reused
        const int local = best_idx - batches * c.hyp;
        memcpy(out->R, p->h_R + (size_t)local * 9, 72);
        memcpy(out->t, p->h_t + (size_t)local * 3, 24);
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
    // was accepted at once (no flag) and only the device-side batch was used
    if (!redo && batches == 1) p->raw_pos += (size_t)7 * total_consumed;
    else p->raw_valid = false;
    if (p->raw_valid && p->raw_pos > p->raw_cap / 2) {   // make room (the solve kernel has finished reading)
      memmove(p->h_raw, p->h_raw + p->raw_pos, (p->raw_fill - p->raw_pos) * sizeof(uint32_t));
      p->raw_fill -= p->raw_pos;
      p->raw_pos = 0;
    }
    out->n_inliers = best_count > 0 ? best_count : 0;
    out->best_index = best_idx;
    out->ransac_iterations = n_done;
    out->draws_consumed = total_consumed;
    out->hyp_valid = hyp_valid;

    // ---- cameras for the DLT of the tracked pairs: C1 = K T_cw(prev) (stream pose), C2 = K [R | t] ----
    if (best_idx >= 0) {
      double Tcw[16], Rt[12];
      double* hC = p->h_C + 24 * cs;
      rigid_inverse(&p->T_wc[(size_t)prev_idx * 16], Tcw);
      k_times_rt(c.K, Tcw, hC);
      for (int r = 0; r < 3; ++r) {
        Rt[4 * r] = out->R[3 * r];
        Rt[4 * r + 1] = out->R[3 * r + 1];
        Rt[4 * r + 2] = out->R[3 * r + 2];
        Rt[4 * r + 3] = out->t[r];
      }
      k_times_rt(c.K, Rt, hC + 12);
      p->dlt_pending = true;
    }
  }
  p->cur = b;
  p->prev_frame = next_idx;
  if (dbg) {
    const double t_ret = now();
    p->dbg_t[0] += t_enq - t_entry;
    p->dbg_t[1] += t_res - t_enq;
    p->dbg_t[2] += t_ret - t_res;
    p->dbg_last_return = t_ret;
    ++p->dbg_steps;
  }
  return VO_OK;
}

int vo_pipeline_export_state_dev(vo_pipeline* p, const vo_step_result* r, int cap, double* d_record) {
  if (!p || !r || !d_record) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  VO_REQUIRE(ctx, cap >= 0, "pipeline_export_state: bad capacity");
  VO_TRY(flush_dlt(p));
  VO_TRY(detect_join(p));
  pose17 h;
  for (int row = 0; row < 3; ++row) {
    for (int c = 0; c < 3; ++c) h.v[4 * row + c] = r->R[3 * row + c];
    h.v[4 * row + 3] = r->t[row];
  }
  h.v[12] = h.v[13] = h.v[14] = 0.0;
  h.v[15] = 1.0;
  const int n = r->best_index >= 0 ? (r->n_tracked < cap ? r->n_tracked : cap) : 0;
  h.v[16] = (double)n;
  const int threads = n * 3 > 17 ? n * 3 : 17;
  hipLaunchKernelGGL(export_state_kernel, dim3(vo_cdiv(threads, 256)), dim3(256), 0, ctx->stream, h, p->d_tri, n, cap,
                     d_record);
  return vo_check_launch(ctx, "export_state_kernel");
}

// per-kernel event times accumulated on the detection stream (vo_prof_read covers the main one)
int vo_pipeline_prof_read(vo_pipeline* p, int kernel_id, double* total_ms, int64_t* launches) {
  if (!p) return VO_EINVAL;
  double a = 0, b = 0;
  int64_t na = 0, nb = 0;
  VO_TRY(worker_idle(p));
  VO_TRY(vo_prof_read(p->ctx, kernel_id, &a, &na));
  int rc = vo_prof_read(p->det, kernel_id, &b, &nb);
  if (rc != VO_OK) return vo_set_error(p->ctx, rc, "%s", vo_last_error(p->det));
  if (total_ms) *total_ms = a + b;
  if (launches) *launches = na + nb;
  return VO_OK;
}

int vo_pipeline_prof_reset(vo_pipeline* p) {
  if (!p) return VO_EINVAL;
  VO_TRY(worker_idle(p));
  VO_TRY(vo_prof_reset(p->ctx));
  return vo_prof_reset(p->det);
}

int vo_pipeline_fetch(vo_pipeline* p, double* kp_next, double* prev_xy, double* next_xy, double* landmarks,
                      double* triangulated, uint8_t* inliers) {
  if (!p) return VO_EINVAL;
  vo_ctx* ctx = p->ctx;
  hipStream_t st = ctx->stream;
  const int n = p->last_ntracked, N = p->cfg.n_keypoints;
  const int cs = p->cset;
  VO_TRY(flush_dlt(p));
  VO_TRY(detect_join(p));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  if (kp_next) VO_HIP_TRY(ctx, hipMemcpy(kp_next, p->d_kp[p->cur], (size_t)N * 16, hipMemcpyDeviceToHost));
  if (n > 0) {
    if (prev_xy) VO_HIP_TRY(ctx, hipMemcpy(prev_xy, p->d_prev_c[cs], (size_t)n * 16, hipMemcpyDeviceToHost));
    if (next_xy) VO_HIP_TRY(ctx, hipMemcpy(next_xy, p->d_next_c[cs], (size_t)n * 16, hipMemcpyDeviceToHost));
    if (landmarks) VO_HIP_TRY(ctx, hipMemcpy(landmarks, p->d_land_c[cs], (size_t)n * 24, hipMemcpyDeviceToHost));
    if (triangulated) VO_HIP_TRY(ctx, hipMemcpy(triangulated, p->d_tri, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (inliers) {
      VO_REQUIRE(ctx, p->last_best >= 0, "pipeline_fetch: no inlier mask for the last step");
      std::vector<uint64_t> row(p->last_words);
      VO_HIP_TRY(ctx, hipMemcpy(row.data(), p->d_masks + (size_t)p->last_best * p->last_words,
                                (size_t)p->last_words * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i) inliers[i] = (uint8_t)((row[i >> 6] >> (i & 63)) & 1ull);
    }
  }
  return VO_OK;
}

}  // extern "C"
