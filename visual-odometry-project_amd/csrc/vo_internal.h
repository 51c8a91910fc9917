// Internal (C++) declarations shared by the HIP translation units of libvo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vo_hip.h"

// A lazily grown device allocation owned by the context.
struct vo_buf {
  void* p = nullptr;
  size_t cap = 0;
};

struct vo_prof_slot {
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct vo_prof_pair {
  int k;
  hipEvent_t a, b;
};

struct vo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  char err[512] = {0};
  // When set, the next launch that supports it (state_regroup_klt, klt_track16) is made with hipExtLaunchKernelGGL and this
  // event as its stop event: the event IS the kernel's completion signal, no marker packet behind the kernel (a marker
  // costs the queue ~4 us before the next dispatch, and a waiting queue one more hop).  Cleared by that launch.
  hipEvent_t next_stop = nullptr;

  // profiling
  bool prof_on = false;
  int prof_kernel = -1;
  int prof_every = 1;                 // bracket every n-th launch of a profiled kernel only (events perturb the stream)
  unsigned prof_seen[VO_K_COUNT] = {0};
  vo_prof_slot prof[VO_K_COUNT];
  std::vector<hipEvent_t> ev_free;
  typedef vo_prof_pair pending_ev;
  std::vector<pending_ev> ev_pending;

  // workspace (device)
  vo_buf img, img2, scores, kp, desc;
  vo_buf nms_keys_l1, nms_idx_l1, nms_keys_a1, nms_idx_a1;   // candidate lists
  vo_buf nms_keys_c, nms_idx_c;                              // compacted candidates
  vo_buf nms_hist, nms_ctl, nms_sel, nms_cand, nms_alive, nms_segcnt, nms_rank;
  bool nms_alive_dirty = false;
  int nms_parity = 0;            // NMS calls alternate between two histograms (the idle one is cleared meanwhile)
  int nms_S = 0;                 // sequences per launch of the last NMS call (the histograms' layout depends on it)
  float* nms_kp_f32 = nullptr;   // optional: the NMS also writes its keypoints as float pairs here (device)
  vo_buf scratch[16];
  vo_buf match_arrived;          // knn2_mfma_kernel's per-query-block arrival counters (zero between calls)
  bool lds_opt_in[2] = {false, false};         // hipFuncSetAttribute is per device: remembered per context (response, NMS)
  hipStream_t aux_stream = nullptr;            // vo_sift: the octaves' last two layers and extrema run beside the next octave
  hipStream_t aux_stream2 = nullptr;           //          (octave 0 on the first, the smaller octaves on the second)
  std::vector<hipEvent_t> aux_events;
  vo_buf sift_arena;
  // pinned host staging
  void* h_pin = nullptr;
  size_t h_pin_cap = 0;
};

// a launch that takes vo_ctx::next_stop as its stop event when one is set (and clears it)
template <class K, class... A>
inline void vo_launch_stop(vo_ctx* ctx, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
  if (ctx->next_stop) {
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, nullptr, ctx->next_stop, 0, args...);
    ctx->next_stop = nullptr;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}

int vo_set_error(vo_ctx* ctx, int code, const char* fmt, ...);
int vo_ensure(vo_ctx* ctx, vo_buf& b, size_t bytes);
int vo_ensure_pinned(vo_ctx* ctx, size_t bytes);

#define VO_HIP_TRY(ctx, expr)                                                         \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return vo_set_error((ctx), VO_EHIP, "%s failed: %s (%s:%d)", #expr,             \
                          hipGetErrorString(e_), __FILE__, __LINE__);                 \
  } while (0)

#define VO_TRY(expr)                 \
  do {                               \
    int s_ = (expr);                 \
    if (s_ != VO_OK) return s_;      \
  } while (0)

#define VO_REQUIRE(ctx, cond, ...)                                  \
  do {                                                              \
    if (!(cond)) return vo_set_error((ctx), VO_EINVAL, __VA_ARGS__); \
  } while (0)

// Brackets one kernel launch with an event pair when profiling is enabled for it.
struct vo_prof_scope {
  vo_ctx* c;
  int k;
  hipEvent_t a = nullptr, b = nullptr;
  vo_prof_scope(vo_ctx* ctx, int kernel);
  ~vo_prof_scope();
};

static inline int vo_check_launch(vo_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return vo_set_error(ctx, VO_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return VO_OK;
}

static inline int vo_cdiv(int a, int b) { return (a + b - 1) / b; }

// Tile order of the image kernels.  Workgroups are dealt to the eight XCDs round-robin by their linear id (ids equal
// mod 8 share an L2 -- observed, not promised: speed only), so with tile = id neighbouring tiles never share an L2
// and every halo byte is fetched once per tile that needs it.  This map gives the workgroups of one XCD a contiguous
// range of the n tiles instead (row-major: whole bands of the image); it is a bijection for any n.
__device__ __forceinline__ unsigned vo_xcd_tile(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, x = id & 7u, j = id >> 3;
  return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + j;
}

struct vo_cam2 {      // the two intrinsic matrices of the two-view bootstrap (bootstrap.hip), row-major
  double K1[9], K2[9];
};

// ---- device-side gates between kernels of different streams (vo_seq_ctl's gate words; pipeline.hip) ----
constexpr int VO_FAULT_GATE_BIT = 128;      // = VO_FAULT_GATE (vo_state.h)
// Two forms (gate_mode).  1: the data handed over are ordinary loads and stores, the waiting side makes an acquire fence at
// agent scope when its word is there, the arriving side a release fence before it arrives -- on this part an invalidate /
// a write-back of the XCD's L2, per workgroup (measured: the kernels running beside the tracker's ~1000 workgroups lose
// their cached population over and over, step 103 -> 133 us).  2: no fences at all -- the handed-over arrays themselves are
// written and read with agent-scope accesses (vo_st_agent / vo_ld_agent: sc1, coherent across the XCDs' L2s one access at
// a time), the arriving side only waits for its own stores to be acknowledged (s_waitcnt) before it counts itself in.
template <class T>
__device__ __forceinline__ T vo_ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ void vo_st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// poll `word` until it reaches `want` (agent scope; acquire: form 1); false after ~2 s of device clock
__device__ __forceinline__ bool vo_gate_wait(const uint32_t* word, uint32_t want, bool acquire = true) {
  const unsigned long long t0 = wall_clock64();
  // (relaxed polls, ONE acquire when the word is there: an acquire at agent scope invalidates the L2 of this XCD)
  while ((int32_t)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 200000000ull) return false;
  }
  if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // (the compiler keeps what follows behind the poll)
  return true;
}

// one arrival per workgroup (call from ONE work item after the workgroup's stores and a __threadfence() -- form 2: after
// vo_stores_done() in every wave and a barrier); the last of `total` publishes `epoch` and resets the counter
__device__ __forceinline__ void vo_gate_arrive(uint32_t* cnt, uint32_t total, uint32_t* word, uint32_t epoch, bool fenced = true) {
  if (fenced) {
    const uint32_t prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == total) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(word, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    const uint32_t prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == total) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// form 2: this wave's stores have been acknowledged (vmcnt / lgkmcnt / expcnt all 0), nothing moves across for the compiler
__device__ __forceinline__ void vo_stores_done() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// ---- internal entry points shared between translation units (not part of the C ABI) ----
// P3P hypotheses + inlier counts of the frame loop (p3p.hip): sample indices derived on the device from raw
// generator outputs in a ring, population size and stream position read on the device.
struct vo_hyp_batch {     // several sequences per launch (grid.y = sequence); the output arrays are S blocks of Hyp entries
  int S = 1;
  size_t X = 0, x = 0;    // doubles between the sequences' landmark / keypoint arrays
  size_t raws = 0;        // words between their generator rings
  size_t ctl = 0;         // bytes between their control blocks (d_n, d_rawpos, d_flag, d_ts point into them)
};
int vo_p3p_hypotheses_ring_dev(vo_ctx* ctx, const double* d_X, const double* d_x, const int32_t* d_n, int n_cap,
                               const double* K, const uint32_t* d_raws, const uint64_t* d_rawpos, uint32_t raw_mask,
                               int Hyp, double thr_sq, double* d_R, double* d_t, uint8_t* d_valid, int32_t* d_counts,
                               uint64_t* d_masks, uint32_t* d_flag, uint64_t* d_ts = nullptr,
                               const vo_hyp_batch* batch = nullptr);
// SIFT tracker mode of the frame pipeline: device-resident detect + describe (sift.hip: extern "C" vo_sift_dev) and
// matching (match.hip)
extern "C" int vo_match_u8_dev(vo_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, int cap_q, const uint8_t* d_t,
                               const int32_t* d_nt, int cap_t, double ratio, int32_t* d_pairs, int32_t* d_npairs,
                               int row_bytes = 128);
// raw (2r+1)^2 patches of the zero-padded image as BYTES, rows padded with zeros to row_bytes (harris.hip)
extern "C" int vo_patch_descriptors_u8_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, const double* d_kp_xy, int N,
                                           int r, uint8_t* d_desc, int row_bytes);
// the next `count` 32-bit outputs of NumPy's PCG64 Generator (ransac_host.hip); advances *rng
void vo_rng_raw32(vo_pcg64* rng, int count, uint32_t* out);
// DLT with a device-resident point count (dlt.hip)
int vo_triangulate_dlt_ndev(vo_ctx* ctx, const double* d_x1, const double* d_x2, const int32_t* d_n, int n_cap,
                            const double* d_C1, const double* d_C2, double* d_X);
// pose refinement with a device-resident point count and either kind of inlier mask (refine.hip)
int vo_refine_pose_ndev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const int32_t* d_n, const double* K,
                        const uint8_t* d_mask8, const uint64_t* d_mask_bits, const double* d_Rt0, int max_iter,
                        double* d_out14, unsigned tag);
// KLT with a device-resident keypoint count (klt.hip).  src (optional): the frame loop's view of its points --
// keypoints 0 .. *n-1 come from d_prev_xy; when *n < frac * *num_features (the re-detect rule, klt.py:207-230)
// the detector's n_det keypoints follow as points *n .. *n + n_det - 1.
struct vo_klt_source {
  const int32_t* n = nullptr;
  const int32_t* num_features = nullptr;
  double frac = 0.0;
  const double* det_kp = nullptr;
  int n_det = 0;
  unsigned long long* ts = nullptr;   // (optional) receives wall_clock64() when the kernel's first work item starts
  const int* det_go = nullptr;        // (optional, one int per sequence) 0: det_kp was not produced, nothing is appended
  // device-side gates (vo_state.h, vo_seq_ctl): wait until *gate_wait reaches gate_want before anything is read; when
  // all workgroups are done publish *gate_set = gate_set_to (arrivals counted in *gate_cnt).  All in the control block.
  const uint32_t* gate_wait = nullptr;
  uint32_t gate_want = 0;
  uint32_t* gate_set = nullptr;
  uint32_t* gate_cnt = nullptr;
  uint32_t gate_set_to = 0;
  int32_t* gate_fault = nullptr;      // receives VO_FAULT_GATE when the wait times out
  int gate_mode = 1;                  // 1 fences, 2 agent-scope accesses to the handed-over arrays (see vo_gate_wait)
};
// several sequences per launch (grid.y = sequence): element strides from one sequence's block to the next
struct vo_klt_batch {
  int S = 1;
  size_t pyr = 0;      // bytes between the sequences' pyramid buffers (same for prev and next)
  size_t xy = 0;       // floats between their keypoint arrays (prev_xy and next_xy)
  size_t out = 0;      // elements between their status / err arrays
  size_t ctl = 0;      // bytes between their control blocks (vo_klt_source.n / num_features / ts point into them)
  size_t det = 0;      // doubles between their detector keypoint lists
};
int vo_klt_track_ndev(vo_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_prev_pyr, const uint8_t* d_next,
                      const uint8_t* d_next_pyr, int H, int W, int n_levels, const float* d_prev_xy, int N,
                      const int32_t* d_n, int win, int max_iter, double eps, double min_eig, float* d_next_xy,
                      uint8_t* d_status, float* d_err, const vo_klt_source* src = nullptr,
                      const vo_klt_batch* batch = nullptr);
int vo_pyramid_build_batch_dev(vo_ctx* ctx, const uint8_t* d_img, size_t img_stride, int S, int H, int W, int n_levels,
                               uint8_t* d_pyr, size_t pyr_stride);
// Several sequences per launch (harris.hip): S images at d_img + s * img_stride -> S score maps at d_scores + s * H * W;
// S score maps -> S keypoint lists at d_kp_xy + s * kp_stride (doubles).  S = 1 is what the C ABI's _dev forms call.
// d_go (optional, S ints on the device): sequences whose word is 0 are skipped by every kernel of the chain
int vo_harris_response_batch_dev(vo_ctx* ctx, const uint8_t* d_img, size_t img_stride, int S, int H, int W, int patch,
                                 double kappa, double* d_scores, const int* d_go = nullptr);
int vo_nms_keypoints_batch_dev(vo_ctx* ctx, const double* d_scores, int S, int H, int W, int N, int r, double* d_kp_xy,
                               size_t kp_stride, const int* d_go = nullptr);
