/*
 * vo_hip.h -- C ABI of libvo_hip.so, the MI355X (gfx950) implementation of the
 * per-frame visual-odometry front-end of saegsali/visual-odometry-project.
 *
 * The reference has no FFI layer: its hot path is reached through Python classes
 * (SURVEY.md section 8b).  Each entry point below replaces the arithmetic of one
 * reference call site, cited as  [ref: path:line]  relative to the reference
 * checkout; the Python package  visual-odometry-project_amd/vo  keeps the
 * reference's class/method names and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - plain C: pointers, sizes, scalars.  No C++ or torch types.
 *  - return 0 (VO_OK) or a negative vo_status; text via vo_last_error().  Nothing
 *    aborts or throws across the boundary.
 *  - there is NO CPU fallback: every compute entry point runs HIP kernels on the
 *    context's device and fails with VO_EHIP if that is impossible.
 *  - "host" entry points take caller-owned host arrays (C-contiguous), copy in/out
 *    and synchronise before returning.  "_dev" entry points take DEVICE pointers,
 *    enqueue on the context's stream and return without synchronising.
 *  - a vo_ctx owns one HIP stream and a device workspace; it is not thread-safe
 *    (one context per host thread / per GPU).
 *  - keypoints are (x, y) pairs; images are row-major uint8, H rows by W columns.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

typedef enum vo_status {
  VO_OK = 0,
  VO_EINVAL = -1,        /* bad argument (shape, range, null pointer)           */
  VO_ENOMEM = -2,        /* host or device allocation failed                    */
  VO_EHIP = -3,          /* HIP runtime error (no device, launch failure, ...)  */
  VO_ECAPACITY = -4      /* an internal candidate list overflowed its capacity  */
} vo_status;

/* ---- context ---------------------------------------------------------------- */

/* device: HIP ordinal.  stream: an existing hipStream_t to enqueue on (e.g. the
 * caller's torch stream), or NULL to create a private non-blocking stream.      */
int vo_create(int device, void* stream, vo_ctx** out);
void vo_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
int vo_version(void);
int vo_sync(vo_ctx* ctx);                       /* hipStreamSynchronize           */
void* vo_stream(vo_ctx* ctx);                   /* the hipStream_t in use         */

/* Device memory helpers so a non-HIP host (ctypes) can keep inputs resident.     */
int vo_dev_alloc(vo_ctx* ctx, size_t bytes, void** out);
int vo_dev_free(vo_ctx* ctx, void* p);
int vo_dev_upload(vo_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vo_dev_download(vo_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* Per-kernel timing with hipEvents on the context's stream.  While enabled every
 * launch of kernel `kernel_id` (VO_K_*) is bracketed by an event pair; the
 * accumulated time and launch count are read back with vo_prof_read (which
 * synchronises).  kernel_id < 0 brackets every kernel.                          */
enum {
  VO_K_HARRIS_RESPONSE = 0,
  VO_K_NMS_CANDIDATES = 1,
  VO_K_NMS_THRESHOLD = 2,
  VO_K_NMS_COMPACT = 3,
  VO_K_NMS_SELECT = 4,
  VO_K_PATCH_DESC = 5,
  VO_K_PYR_DOWN = 6,
  VO_K_KLT_TRACK = 7,
  VO_K_DLT = 8,
  VO_K_P3P_SOLVE = 9,
  VO_K_P3P_SCORE = 10,
  VO_K_REPROJ = 11,
  VO_K_MATCH = 12,
  VO_K_COUNT = 32
};
int vo_prof_enable(vo_ctx* ctx, int kernel_id);
int vo_prof_disable(vo_ctx* ctx);
int vo_prof_read(vo_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
int vo_prof_reset(vo_ctx* ctx);
const char* vo_kernel_name(int kernel_id);

/* ---- Harris response + greedy NMS ------------------------------------------
 * [ref: src/vo/features/harris.py:99-137]  response: true-convolution Sobel ->
 * int products -> patch x patch box sums -> det - kappa*trace^2 (three IEEE
 * roundings, no FMA) -> clamp <0 -> zero border of patch/2+1.  scores: H*W
 * float64 in image coordinates, bit-identical to the reference.
 * [ref: src/vo/features/harris.py:139-152]  greedy argmax NMS, radius r,
 * including its slicing semantics.  kp_xy: N*2 float64 (x, y),
 * bit-identical to the reference's keypoints (N,2,1).
 * patch must be odd, 3..31; 0 <= r <= 32; 1 <= N <= 16384.                      */
int vo_harris_response(vo_ctx* ctx, const uint8_t* img, int H, int W, int patch,
                       double kappa, double* scores);
int vo_harris_keypoints(vo_ctx* ctx, const uint8_t* img, int H, int W, int patch,
                        double kappa, int N, int r, double* kp_xy,
                        double* scores /* nullable */);
int vo_nms_keypoints(vo_ctx* ctx, const double* scores, int H, int W, int N, int r,
                     double* kp_xy);
int vo_harris_response_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int patch,
                           double kappa, double* d_scores);
int vo_nms_keypoints_dev(vo_ctx* ctx, const double* d_scores, int H, int W, int N,
                         int r, double* d_kp_xy);

/* [ref: src/vo/features/harris.py:160-194]  raw (2r+1)^2 patches of the
 * zero-padded image, row-major, as float64.  desc: N*(2r+1)^2.                  */
int vo_patch_descriptors(vo_ctx* ctx, const uint8_t* img, int H, int W,
                         const double* kp_xy, int N, int r, double* desc);
int vo_patch_descriptors_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W,
                             const double* d_kp_xy, int N, int r, double* d_desc);

/* ---- pyramidal KLT -----------------------------------------------------------
 * [ref: src/vo/features/klt.py:233-249]  cv2.calcOpticalFlowPyrLK(prev, next,
 * prevPts, None, winSize=(win,win), maxLevel=max_level, criteria=(EPS|COUNT,
 * max_iter, eps)), default minEigThreshold = 1e-4, flags = 0.  Outputs as OpenCV:
 * next_xy N*2 float32, status N uint8, err N float32 (mean |patch diff| / 32).
 * Pyramid level l+1 is ((H+1)/2, (W+1)/2); the number of levels actually used is
 * vo_klt_num_levels (the builder stops when a level is not larger than the
 * window).  The _dev form takes level 0 (the image) and a buffer holding levels
 * 1..n-1 back to back, each rounded up to 256 bytes (vo_pyramid_bytes).        */
int vo_klt_num_levels(int H, int W, int win, int max_level);
size_t vo_pyramid_bytes(int H, int W, int n_levels);
int vo_pyr_down(vo_ctx* ctx, const uint8_t* img, int H, int W, uint8_t* out);
int vo_pyramid_build_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int n_levels,
                         uint8_t* d_pyr);
int vo_klt_track(vo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int H, int W,
                 const float* prev_xy, int N, int win, int max_level, int max_iter,
                 double eps, double min_eig, float* next_xy, uint8_t* status, float* err);
int vo_klt_track_dev(vo_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_prev_pyr,
                     const uint8_t* d_next, const uint8_t* d_next_pyr, int H, int W,
                     int n_levels, const float* d_prev_xy, int N, int win, int max_iter,
                     double eps, double min_eig, float* d_next_xy, uint8_t* d_status,
                     float* d_err);

/* ---- DLT triangulation --------------------------------------------------------
 * [ref: src/vo/landmarks/triangulation.py:352-389, 38-86; src/vo/helpers.py:57-83]
 * per point A = [[x1]_x C1 ; [x2]_x C2] (6x4), smallest right singular vector,
 * de-homogenised.  x1, x2: n*2 pixels; C1: 3x4 row-major, or n of them when
 * c1_per_point != 0 (triangulate_candidates); C2: 3x4; X: n*3.                  */
int vo_triangulate_dlt(vo_ctx* ctx, const double* x1, const double* x2, int n,
                       const double* C1, int c1_per_point, const double* C2, double* X);
int vo_triangulate_dlt_dev(vo_ctx* ctx, const double* d_x1, const double* d_x2, int n,
                           const double* d_C1, int c1_per_point, const double* d_C2,
                           double* d_X);

/* ---- P3P hypotheses + reprojection scoring -------------------------------------
 * [ref: src/vo/pose_estimation/p3p.py:51-79]   model_fn: cv2.solvePnP(P3P) on the 4
 *       sampled correspondences -> (R, t) world->camera, or None
 * [ref: src/vo/pose_estimation/p3p.py:81-108]  error_fn: squared reprojection error
 * [ref: src/vo/algorithms/ransac.py:104-106]   inliers = error < thr (strict); count
 * X: N*3 landmarks, x: N*2 pixels, K: 3x3 row-major (HOST pointer in both forms),
 * samples: Hyp*4 indices into the N correspondences.  Outputs per hypothesis:
 * R Hyp*9, t Hyp*3, valid Hyp (0 = the reference's "model is None"), counts Hyp,
 * masks Hyp*ceil(N/64) 64-bit words (bit i%64 of word i/64 = point i is an inlier;
 * nullable).  vo_reproj_inliers scores one pose: mask N bytes and/or err N.     */
int vo_p3p_hypotheses(vo_ctx* ctx, const double* X, const double* x, int N, const double* K,
                      const int32_t* samples, int Hyp, double thr_sq, double* R, double* t,
                      uint8_t* valid, int32_t* counts, uint64_t* masks);
int vo_p3p_hypotheses_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N,
                          const double* K, const int32_t* d_samples, int Hyp, double thr_sq,
                          double* d_R, double* d_t, uint8_t* d_valid, int32_t* d_counts,
                          uint64_t* d_masks);
int vo_reproj_inliers(vo_ctx* ctx, const double* X, const double* x, int N, const double* K,
                      const double* R, const double* t, double thr_sq, uint8_t* mask,
                      double* err);
int vo_reproj_inliers_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N,
                          const double* K, const double* d_Rt, double thr_sq,
                          uint8_t* d_mask, double* d_err);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
