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

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
