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
  VO_ECAPACITY = -4,     /* an internal candidate list overflowed its capacity  */
  VO_ETRACKING = -5      /* fewer than 4 triangulated tracks survive: no pose   */
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
  VO_K_GATHER = 13,
  VO_K_NMS_ROUND = 14,
  VO_K_NMS_COLLECT = 15,
  VO_K_NMS_RANK = 16,
  VO_K_NMS_EMIT = 17,
  VO_K_SIFT_SCALESPACE = 18,
  VO_K_SIFT_DETECT = 19,
  VO_K_SIFT_DESCRIBE = 20,
  VO_K_REFINE = 21,
  VO_K_STATE_CANDIDATES = 22,
  VO_K_STATE_REGROUP = 23,
  VO_K_RANSAC_REPLAY = 24,
  VO_K_STATE_LANDMARKS = 25,
  VO_K_EXPORT = 26,
  VO_K_COUNT = 32
};
int vo_prof_enable(vo_ctx* ctx, int kernel_id);
/* bracket only every n-th launch of a profiled kernel (default 1): the event pair itself costs
 * the stream a few microseconds, vo_prof_read then reports the sampled launches           */
int vo_prof_set_sampling(vo_ctx* ctx, int every);
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
/* The detector on S frames of one size in one set of launches (the sequence is the grid's extra dimension: several
 * sequences per GPU advance together).  imgs: S*H*W bytes; kp_xy: S*N*2; scores: S*H*W or NULL.
 * Results are those of S calls of vo_harris_keypoints.                                                     */
int vo_harris_keypoints_batch(vo_ctx* ctx, const uint8_t* imgs, int S, int H, int W, int patch, double kappa, int N,
                              int r, double* kp_xy, double* scores /* nullable */);
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
 * window).  The _dev form takes the images and, for each, the buffer that
 * vo_pyramid_build_dev filled (vo_pyramid_bytes): an opaque layout holding every
 * level, level 0 included, with a reflect-101 border so the tracker's blocks are
 * plain in-bounds reads.  The tracker reads the levels from those buffers only. */
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

/* ---- two-view bootstrap ----------------------------------------------------------
 * [ref: src/vo/landmarks/triangulation.py:110-350, src/vo/helpers.py:31-54; called once per sequence, src/main.py:204-230]
 * vo_fundamental_hypotheses: the model_fn / error_fn pair of _find_fundamental_matrix_ransac (:134-145) for Hyp samples
 *   at once -- samples: Hyp*8 indices into the N correspondences p1, p2 (N*2 each; the reference hands the loop
 *   Hartley-normalised points, :147-149).  Per sample: Kronecker rows (:203-205), null vector of the 8x9 system
 *   (:209-212), rank-2 projection (:214-217) -> F Hyp*9 row-major.  normalize_samples != 0: the sample's own Hartley
 *   normalisation around the fit (_find_fundamental_matrix(is_normalized=False), :195-198, :220-221).  error_kind 0:
 *   (p2^T F p1)^2 (:140-145); 1: the larger squared distance to the epipolar lines of the two images.  inlier =
 *   error < threshold (src/vo/algorithms/ransac.py:104-106); counts Hyp, masks Hyp*ceil(N/64) words (nullable).
 *   The sequential accept / adapt rule over (counts) is vo_ransac_replay's (every sample has a model: valid = 1).
 * vo_fundamental_fit: _find_fundamental_matrix (:165-222) over the correspondences whose mask byte is set (NULL: all):
 *   what RANSAC's closing model_fn(population[inliers]) computes (ransac.py:123-127).  normalize as above.
 * vo_essential_decompose: _decompose_essential_matrix (:245-277): E 3x3 -> M4 4*12, [R_j | (-1)^i T] at index 2i+j.
 *   The four candidates are the reference's set; which of the two rotations is "R_0" and which sign of T comes first
 *   depends on the signs the SVD routine picks (LAPACK's in the reference) and may differ.
 * vo_relative_pose: _find_relative_pose (:279-350) given F: E = K2^T F K1, the four candidates, the cheirality votes
 *   over the correspondences whose inliers byte is set (NULL: all) by DLT against K1 [I|0] (:313-332, strict `>`),
 *   the winner's triangulation of ALL N correspondences -> M 12 (3x4), X N*3 (frame 1), mask_out N (nullable):
 *   inliers & in front of both cameras (:341-348), M4 48 (nullable).                                              */
int vo_fundamental_hypotheses(vo_ctx* ctx, const double* p1, const double* p2, int N, const int32_t* samples, int Hyp,
                              int normalize_samples, int error_kind, double threshold, double* F, int32_t* counts,
                              uint64_t* masks);
int vo_fundamental_fit(vo_ctx* ctx, const double* p1, const double* p2, int N, const uint8_t* mask, int normalize,
                       double* F);
int vo_essential_decompose(vo_ctx* ctx, const double* E, double* M4);
int vo_relative_pose(vo_ctx* ctx, const double* x1, const double* x2, int N, const uint8_t* inliers, const double* K1,
                     const double* K2, const double* F, double* M, double* X, uint8_t* mask_out, double* M4);

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
/* The frame loop's hypothesis kernel takes the decision `error < thr_sq` (ransac.py:104-106) on the sum of squares
 * s, where error = fl(fl(sqrt(s))^2) is what error_fn computes (p3p.py:81-108: norm, then square):
 * vo_inlier_sum_sq_limit returns the largest double s whose error is below thr_sq (-1: there is none), and
 * `s <= limit` is the same decision because rounding and sqrt are monotone.  Host arithmetic only.          */
double vo_inlier_sum_sq_limit(double thr_sq);

/* ---- pose refinement --------------------------------------------------------------
 * [ref: src/vo/pose_estimation/p3p.py:188-213 _nonlinear_refinement; helpers.py:86-142]
 * Minimises sum_i |x_i - proj(K, R X_i + t)|^2 over the pose, from (R0, t0), over the points
 * whose mask byte is non-zero (all points if the mask is NULL) -- the objective the reference
 * hands to scipy.optimize.least_squares.  Gauss-Newton on the left-multiplied increment with
 * the analytic Jacobian, at most max_iter (<= 100) accepted steps, fp64; converges to the
 * minimiser itself (SciPy's default tolerances stop within ~1e-4 of it).  Outputs R (9,
 * row-major), t (3), the number of accepted steps and the final cost (sum of squared pixel
 * errors).  The _dev form takes Rt0 = R0 then t0 (12 doubles) and writes 14 doubles:
 * R, t, iterations, cost.                                                              */
int vo_refine_pose(vo_ctx* ctx, const double* X, const double* x, int N, const double* K,
                   const uint8_t* inlier_mask, const double* R0, const double* t0, int max_iter,
                   double* R, double* t, int32_t* iterations, double* cost);
int vo_refine_pose_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const double* K,
                       const uint8_t* d_mask, const double* d_Rt0, int max_iter, double* d_out14);

/* ---- descriptor matching ---------------------------------------------------------
 * [ref: src/vo/features/harris.py:246-262, src/vo/features/sift.py:38-54]
 * cv2.BFMatcher().knnMatch(q, t, k=2) + ratio test + first-come uniqueness on the train
 * index, queries in order.  q: nq*D, t: nt*D float32; pairs: up to nq (query, train)
 * rows; n_pairs: rows written.  Distances are exact for integer-valued descriptors.   */
int vo_match_knn2_ratio(vo_ctx* ctx, const float* q, int nq, const float* t, int nt, int D,
                        double ratio, int32_t* pairs, int32_t* n_pairs);
int vo_knn2_dev(vo_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt, int D,
                int32_t* d_best, double* d_d2);

/* ---- Shi-Tomasi corners -----------------------------------------------------------
 * [ref: src/vo/features/klt.py:98]  cv2.goodFeaturesToTrack(img, mask, maxCorners,
 * qualityLevel, minDistance, blockSize).  xy: max_corners*2 float32 (or H*W/4*2 when
 * max_corners <= 0); n: corners found.  vo_min_eigen_map exposes the H*W float32 map.
 * All stages run on the device: eigenvalue map, thresholded 3x3 maxima, descending order
 * (value, then address) and the greedy minimum-distance walk over OpenCV's cell grid.     */
int vo_good_features(vo_ctx* ctx, const uint8_t* img, int H, int W, const uint8_t* mask,
                     int max_corners, double quality, double min_dist, int block, float* xy,
                     int32_t* n);
int vo_min_eigen_map(vo_ctx* ctx, const uint8_t* img, int H, int W, int block, float* eig);

/* ---- SIFT ---------------------------------------------------------------------------
 * [ref: src/vo/features/sift.py:10,17]  cv2.SIFT_create().detectAndCompute(image, None)
 * with OpenCV's defaults (3 layers/octave, contrast 0.04, edge 10, sigma 1.6, image
 * doubled first).  kp: cap*6 float32 (x, y, size, angle, response, octave) in image
 * coordinates, ordered as KeyPointsFilter::removeDuplicatedSorted leaves them; desc:
 * cap*128 float32 (integer-valued 0..255).  If more than `cap` keypoints are found the
 * `cap` strongest by response are kept (KeyPointsFilter::retainBest).  cap <= 0: keep every
 * keypoint, as the reference does (nfeatures = 0); kp / desc must then hold
 * vo_sift_capacity(H, W) rows, the library's own list capacity for that image size --
 * more keypoints than that is VO_ECAPACITY, never a silent truncation.                  */
int vo_sift_capacity(int H, int W);
int vo_sift(vo_ctx* ctx, const uint8_t* img, int H, int W, int cap, float* kp, float* desc,
            int32_t* n);
/* The same with the frame and the results in device memory, enqueued on the context's stream (no synchronisation):
 * d_kp cap*6 float32, d_desc cap*128 float32 and / or d_desc_u8 cap*128 bytes (the descriptor values are whole numbers
 * 0..255; either may be NULL), d_n one int32.  1 <= cap <= 4000: the final order, the duplicate filter and the cap run
 * on the device too (one workgroup sorts the cap + ties described rows).                                          */
int vo_sift_dev(vo_ctx* ctx, const uint8_t* d_img, int H, int W, int cap, float* d_kp, float* d_desc, uint8_t* d_desc_u8,
                int32_t* d_n);

/* ---- RANSAC control (host-side, bit-compatible with the reference) --------------
 * [ref: src/vo/algorithms/ransac.py:52, 92-94]  the sample stream of
 *   np.random.default_rng(2023).choice(np.arange(pop), replace=False, size=s)
 * vo_pcg64 is NumPy's PCG64 bit-generator state (Generator.bit_generator.state:
 * 128-bit state and increment split into 64-bit halves, plus the buffered 32-bit
 * half).  vo_rng_choice draws `count` samples of s indices and advances the state.
 * [ref: src/vo/algorithms/ransac.py:58-67, 90-121]  vo_ransac_replay walks
 * pre-computed hypotheses (valid, inlier count) through the reference's sequential
 * accept / adaptive-iteration rule; vo_ransac_state holds the fields that persist
 * on the reference's RANSAC object between calls.  max_iterations < 0 = unbounded. */
typedef struct vo_pcg64 {
  uint64_t state_hi, state_lo, inc_hi, inc_lo;
  uint32_t has_uint32, uinteger;
} vo_pcg64;
typedef struct vo_ransac_state {
  double outlier_ratio, confidence;
  int64_t max_iterations, n_iterations;
  int32_t s, adaptive;
} vo_ransac_state;
int vo_rng_choice(vo_pcg64* rng, int pop, int s, int count, int32_t* out);
int64_t vo_ransac_num_iterations(double confidence, double outlier_ratio, int s);
int vo_ransac_replay(vo_ransac_state* st, const uint8_t* valid, const int32_t* counts, int B,
                     int N, int64_t* n_done, int32_t* best_count, int32_t* best_idx,
                     int idx_offset, int* consumed, int* finished);

/* ---- device-resident frame pipeline ------------------------------------------------
 * The steady-state loop of the reference driver [ref: src/main.py:248-286], KLT tracker mode
 * [ref: src/vo/features/tracker.py:56-57], with everything the reference carries from frame to
 * frame kept in HBM: the Features arrays (keypoints, state codes, landmarks, track starts, track
 * start poses, candidate mask) [ref: src/vo/primitives/features.py:4-54], State's current and
 * previous pose [ref: src/vo/primitives/state.py:9-15], the estimator's RANSAC fields and its
 * position in the generator's output stream [ref: src/vo/algorithms/ransac.py:42-56], and
 * KLTTracker._num_features.  The pipeline takes IMAGES ONLY; one step = one frame:
 *   side streams: pyramid(next); Harris response + greedy NMS on next (the detector whose
 *                 keypoints the next step appends when too few tracks survive; launched for
 *                 every frame, executed per sequence within detect_margin of that limit)
 *   re-detect     [ref: src/vo/features/klt.py:207-230, 117-189]  length < 0.8 * _num_features:
 *                 the detector's keypoints of `prev` are appended as unmatched features
 *   KLT           [ref: klt.py:233-249]  every feature prev -> next, keep status & err < thr
 *   Matches       [ref: src/vo/primitives/matches.py:26-212]  regroup into [triangulated |
 *                 matched | newly matched], landmarks / track starts / start poses carried over
 *   estimate_pose [ref: src/vo/pose_estimation/p3p.py:123-186, use_opencv=False]  `hyp` samples
 *                 drawn with the reference's generator, solved and scored; the sequential
 *                 accept / adapt rule [ref: ransac.py:90-121] replayed on the device; refinement
 *                 over the inliers [ref: p3p.py:188-213]
 *   State         [ref: src/vo/primitives/state.py:38-50, 162-172, 135-160, 174-219]  pose,
 *                 reset_outliers, compute_candidates (bearing angle >= threshold)
 *   triangulate_candidates [ref: src/vo/landmarks/triangulation.py:38-86]  one start pose per
 *                 track; update_with_world_landmarks + _check_landmarks [ref: state.py:69-107]
 * No stage waits for the host: a step is a chain of launches, its result a record the last
 * kernel writes to host-visible memory.  The two-view bootstrap [ref: main.py:204-230] runs on
 * the host (vo/driver.py) and hands its Features / poses over with vo_pipeline_set_state.      */
typedef struct vo_pipeline vo_pipeline;
typedef struct vo_pipeline_config {
  int32_t H, W, n_frames;        /* n_frames: frame store in HBM (vo_pipeline_set_frame slots)   */
  int32_t n_keypoints, harris_patch, nms_radius;      /* detector: harris.py:16-25               */
  double harris_kappa;
  int32_t klt_win, klt_max_level, klt_max_iter, hyp;
  double klt_eps, klt_min_eig, klt_err_threshold;      /* klt.py:29-39                            */
  double p3p_thr_sq, ransac_outlier_ratio, ransac_confidence;
  int64_t ransac_max_iterations;                       /* < 0: unbounded                          */
  double K[9];
  double Kinv[9];                /* inverse intrinsics as the caller computes them (the reference:
                                    np.linalg.inv, camera.py:88); all zero = computed here        */
  int32_t refine_iters;          /* > 0: refine the accepted pose over its inliers (vo_refine_pose) */
  int32_t feature_cap;           /* capacity of the Features arrays; 0 = 2 * n_keypoints          */
  double bearing_threshold;      /* State(bearing_threshold), state.py:8; 0 = 0.0075             */
  double redetect_fraction;      /* klt.py:212; 0 = 0.8                                           */
  int32_t debug_fault_every;     /* test hook: every n-th step takes the host recovery path      */
  int32_t redetect_start_pose;   /* start pose of re-detected keypoints: 0 = np.eye(4), as the reference's
                                    update_features writes it (klt.py:148-153) -- away from the origin such
                                    tracks triangulate against the wrong baseline; 1 = the current pose,
                                    what State.reset_outliers gives a restarted track (state.py:170-172) */
  double detect_margin;          /* the detector chain is launched for every frame (its keypoints are what the NEXT
                                    step appends when fewer than redetect_fraction of the tracks are left, and that
                                    count is known one step too late for a launch without a host turn); a sequence
                                    sits it out unless its count, extrapolated by detect_losses times the last
                                    step's loss, is below (redetect_fraction + detect_margin) * num_features.  A
                                    sequence that falls through all of that in one frame is finished through the
                                    host path (detector run then).  0 = 0.01; < 0 = the detector runs on every
                                    frame.  The reference detects only below the limit itself.             */
  int32_t debug_never_detect;    /* test hook: the detector runs only when forced (state hand-over, host path)  */
  double detect_losses;          /* how many of the last step's losses the count is extrapolated by in the detector's
                                    decision (see detect_margin).  0 = 2.5                                          */
  int32_t sequences;             /* S independent frame streams advancing in lock step through the same
                                    launches (the sequence is the grid's extra dimension; SURVEY.md 8e: streams
                                    are independent, so they batch).  0 = 1.  Every sequence has its own frame
                                    store, Features, State, RANSAC object and generator; the plain entry points
                                    address sequence 0, the _seq forms any of them.                      */
  int32_t tracker_mode;          /* 0: KLT tracker with the Harris detector (everything above); 1: SIFT
                                    [ref: src/vo/features/tracker.py:60-61, src/vo/features/sift.py:23-56] -- per frame
                                    detect + describe (the sift_cap strongest keypoints; the reference keeps all of them),
                                    2-NN + ratio + first-come uniqueness against the descriptors the current Features
                                    carry, Matches regroup from the pair list with the descriptors following their
                                    keypoints [ref: src/vo/primitives/matches.py:51-58, 134-141], then the same pose
                                    estimation and State bookkeeping.  One sequence per pipeline; the state handed over
                                    needs its descriptors too (vo_pipeline_set_descriptors).
                                    2: Harris [ref: tracker.py:58-59, src/vo/features/harris.py:50-84, 196-264] -- per
                                    frame the detector's n_keypoints keypoints (response + greedy NMS, every frame),
                                    their raw 19x19 patches (descriptor_radius 9) as bytes, 2-NN + 0.85 ratio +
                                    uniqueness on the matrix cores (361 values padded to 384), then as mode 1.      */
  int32_t sift_cap;              /* keypoints kept per frame in SIFT mode (0 = n_keypoints; <= 4000, <= feature_cap) */
  double match_ratio;            /* 0 = the reference's: 0.8 in SIFT mode (sift.py:49), 0.85 in Harris mode (harris.py:255) */
} vo_pipeline_config;
typedef struct vo_step_result {
  double R[9], t[3];            /* world -> camera pose of `next` (best hypothesis)   */
  int32_t n_tracked;            /* features of the new frame (survivors of the KLT filter) */
  int32_t n_inliers;            /* inliers of the returned pose                       */
  int32_t best_index;           /* index of the accepted hypothesis                   */
  int32_t hyp_valid;            /* hypotheses with a P3P solution among those scored  */
  int64_t ransac_iterations;    /* iterations the reference loop would have counted   */
  int32_t draws_consumed;       /* samples consumed from the generator                */
  int32_t refine_iterations;    /* accepted refinement steps; -1: not refined         */
  double R_refined[9], t_refined[3];   /* refined pose (= R, t when not refined)      */
  double refine_cost;           /* sum of squared inlier reprojection errors after it */
  int32_t n_features_in;        /* features handed to the tracker (after a re-detect) */
  int32_t redetected;           /* 1: the detector's keypoints were appended          */
  int32_t n_triangulated;       /* tracked features with a landmark: the P3P population */
  int32_t n_candidates;         /* tracks that passed the bearing test and were triangulated */
  int32_t n_dropped;            /* landmarks the cheirality check removed             */
  int32_t n_landmarks;          /* features in state 2 after the step                 */
  int32_t fault;                /* internal: reason the step left the device-only path */
  int32_t recovered;            /* 1: the step was finished through the host path     */
  int32_t detector_ran;         /* 1: the detector was executed on this step's `prev` frame (detect_margin) */
  int32_t reserved;             /* recovered steps: the reason as fault bits (1 few landmarks, 2 a draw NumPy might have
                                   rejected, 4 rule not done after `hyp` samples, 8 capacity, 16 forced, 32 detector skipped) */
  uint64_t raw_pos;             /* generator outputs consumed so far (32-bit words)   */
  double T_wc[12];              /* camera -> world pose after the step, rows 0..2 (State.curr_pose) */
  uint64_t ts[8];               /* device clock (100 MHz ticks) at the start of: tracker, regroup, hypotheses, pose,
                                   landmark stage, at the end of the step (the record's last write); [6], [7]:
                                   inside the pose kernel, RANSAC replay done / refinement done               */
  uint32_t seq_head, seq_tail;  /* internal; the last two words.  In the mapped record: seq_head = the step's sequence
                                   number XOR every other 32-bit word of the record, seq_tail = the number + a
                                   position-weighted sum of those words, so that a copy taken while some of its lines
                                   were still on their way is recognised (vo_record_check); in what the collect calls
                                   return both equal the number                                                    */
} vo_step_result;
/* The self-check of a result record in mapped host memory (the GPU's stores to host memory arrive line by line, in no
 * particular order): _seal closes a record for step `seq` the way the device does, _check returns 1 when the copy is one
 * whole record of that step.  Host arithmetic only.                                                              */
void vo_record_seal(vo_step_result* rec, unsigned seq);
int vo_record_check(const vo_step_result* rec, unsigned seq);
int vo_pipeline_create(vo_ctx* ctx, const vo_pipeline_config* cfg, vo_pipeline** out);
void vo_pipeline_destroy(vo_pipeline* p);
/* A closed pipeline's two side streams and their workspace are kept for the next pipeline of the same configuration (creating
 * and destroying them per pipeline stalled inside the runtime about once in 400 cycles); this destroys what is kept.  Nothing in
 * the reference corresponds to it (its objects are garbage-collected).                                                      */
void vo_pipeline_release_cached(void);
/* 1 when a binding should call vo_pipeline_release_cached from its own exit hook (the library's own policy: under a profiler,
 * or VO_SIDE_POOL_ATEXIT=1), 0 when what is kept is left to the end of the process.                                       */
int vo_pipeline_release_cached_at_exit(void);
/* frame store: copies a host image into slot idx of the frame store.  The caller's buffer is free on return (it is
 * copied into a pinned staging buffer of that slot); the transfer itself is queued in front of the pyramid that reads
 * the slot and the call does not wait for it.  A slot that a step in flight reads is refused (VO_EINVAL), and so is the
 * slot of the frame submitted last once a state exists: the next step tracks FROM that image.                       */
int vo_pipeline_set_frame(vo_pipeline* p, int idx, const uint8_t* img);
int vo_pipeline_seed(vo_pipeline* p, const vo_pcg64* rng);
int vo_pipeline_get_rng(vo_pipeline* p, vo_pcg64* rng);      /* estimator generator state after the last collected step */
/* Hands over the Features of frame idx (the reference's state.curr_frame.features after the
 * bootstrap) and State's poses: n features -- kp n*2 float64 (the tracker reads them rounded to
 * float32, as cv2.calcOpticalFlowPyrLK takes them), state n bytes (0/1/2), landmarks
 * n*3, tracks n*2, poses n*16 (4x4 row-major, camera-to-world; NaN rows where the reference holds
 * NaN) -- plus curr / prev pose as 4x4 camera-to-world AND world-to-camera matrices (the
 * reference forms the latter with np.linalg.inv; passing both keeps every later product the
 * same), and KLTTracker._num_features.  The pyramid and the detector's keypoints of frame idx are
 * made by the first submit after it.                                                            */
int vo_pipeline_set_state(vo_pipeline* p, int idx, int n, const double* kp, const uint8_t* state,
                          const double* landmarks, const double* tracks, const double* poses,
                          const double* T_wc, const double* T_cw, const double* T_wc_prev,
                          const double* T_cw_prev, int num_features);
/* A stream that is walked more than once (bench.py: 100 resident frames, pass after pass): _checkpoint keeps a copy of
 * every sequence's Features / State as they are now -- nothing in flight -- in HBM; _rewind puts that copy back as the
 * state of the frame it was taken at and queues that frame's pyramid and detection, all asynchronously on the pipeline's
 * streams (nothing in flight; no host synchronisation).  What lives on the reference's estimator object across frames
 * (RANSAC.n_iterations / outlier_ratio [ref: src/vo/algorithms/ransac.py:47-56], the generator) is NOT rewound.      */
/* Descriptor tracker modes: the descriptors (n x 128 float32 in SIFT mode, n x 361 in Harris mode; whole numbers
 * 0..255) of the features handed over by the last vo_pipeline_set_state, in the same order.                      */
int vo_pipeline_set_descriptors(vo_pipeline* p, const float* desc, int n);
int vo_pipeline_checkpoint(vo_pipeline* p);
int vo_pipeline_rewind(vo_pipeline* p);
/* Downloads the current Features (arrays sized to the capacity vo_pipeline_feature_cap returns;
 * any pointer may be NULL); n_out: feature count.  Nothing may be in flight.                  */
int vo_pipeline_feature_cap(vo_pipeline* p);
int vo_pipeline_get_state(vo_pipeline* p, int32_t* n_out, double* kp, uint8_t* state, uint8_t* candidate_mask,
                          double* landmarks, double* tracks, double* poses, double* T_wc, double* T_wc_prev,
                          vo_ransac_state* rs, int32_t* num_features);
/* keypoints the detector found on the frame submitted last (n_keypoints*2 float64)              */
int vo_pipeline_get_detection(vo_pipeline* p, double* kp_xy);
/* One frame.  submit enqueues all GPU work of the step prev_idx -> next_idx and returns; collect
 * waits for the oldest submitted step's record.  At most two steps may be in flight (the frame
 * store and the per-frame buffers rotate over three slots); with submit(k+1) before collect(k)
 * the host's launches overlap the GPU's work.  step = submit + collect.                        */
int vo_pipeline_step(vo_pipeline* p, int prev_idx, int next_idx, vo_step_result* out);
int vo_pipeline_submit(vo_pipeline* p, int prev_idx, int next_idx);
int vo_pipeline_collect(vo_pipeline* p, vo_step_result* out);
/* Test / integration entry: the bookkeeping of one frame with everything the estimators would
 * produce given by the caller -- Matches(frame1 = current features, frame2 = Features(new_kp),
 * pairs) [ref: matches.py:11-212], update_with_world_pose(T), outliers[triangulate_inliers] =
 * ~p3p_inliers, reset_outliers, compute_candidates [phase 1]; triangulate_candidates,
 * update_with_world_landmarks [phase 2].  phases: 1, 2 or 3.  Synchronous.                     */
int vo_pipeline_bookkeeping(vo_pipeline* p, int phases, const double* new_kp, int n2, const int32_t* pairs,
                            int M, const double* T_wc, const double* T_cw, const uint8_t* p3p_inliers);
/* vo_prof_read / vo_prof_reset over all of the pipeline's streams                              */
int vo_pipeline_prof_read(vo_pipeline* p, int kernel_id, double* total_ms, int64_t* launches);
int vo_pipeline_prof_reset(vo_pipeline* p);
/* Shared-map record of the last collected step -> DEVICE memory (async):
 * [T_cw 4x4 row-major (16, the refined pose) | n (1) | n landmarks x 3 (the step's P3P
 * population, n <= cap)], all f64, 17 + 3*cap doubles.  The caller all-gathers records over RCCL
 * (bench.py), one or several frames per collective:
 *   _post  queues the record on the pipeline's stream and returns at once;
 *   _join  orders the pipeline's stream and `consumer` (hipStream_t; NULL = the context's
 *          stream) both ways: work enqueued on `consumer` after the call sees every record
 *          posted so far, and records posted after the call are written after everything
 *          `consumer` held at the time of the call (an exchange still reading the buffer).   */
int vo_pipeline_export_state_post(vo_pipeline* p, const vo_step_result* r, int cap, double* d_record);
int vo_pipeline_export_state_post_seq(vo_pipeline* p, int seq, const vo_step_result* r, int cap, double* d_record);
int vo_pipeline_export_state_join(vo_pipeline* p, void* consumer);
/* the ransac.py:58-67 iteration bound through the pipeline's threshold table (what the device
 * evaluates); equals vo_ransac_num_iterations clipped to max_iterations                         */
int64_t vo_pipeline_ransac_bound(vo_pipeline* p, double outlier_ratio);
/* Several sequences per pipeline (vo_pipeline_config.sequences = S): the sequences step together --
 * vo_pipeline_submit enqueues frame slot prev_idx -> next_idx of EVERY sequence in one set of launches,
 * vo_pipeline_collect_all waits for all S records (outs: S entries; vo_pipeline_collect returns sequence
 * 0's).  A sequence whose step leaves the device-only path is redone alone through the host path; the
 * others are not held up on the device.  vo_pipeline_seed gives every sequence's estimator the same
 * generator state (each then advances by its own draws).                                        */
int vo_pipeline_sequences(vo_pipeline* p);
int vo_pipeline_set_frame_seq(vo_pipeline* p, int seq, int idx, const uint8_t* img);
/* The same from PINNED host memory (vo_host_alloc; a frame grabber's or a decoder's output buffer): no staging copy on the
 * host, and the DMA runs on a stream of its own beside the pipeline's kernels instead of in front of the frame's pyramid --
 * a frame uploaded while the previous step is still in flight costs the step nothing.  The buffer must not change until
 * the upload is over: vo_pipeline_frame_uploaded(idx, wait) -- 1 when it is (wait != 0: blocks until then), 0 when not --
 * or the collect of a step that read the slot.  Replaces the image hand-over of the reference's per-frame loop
 * (src/main.py:248-251: frame.image goes into Tracker.track as a host array).                                         */
int vo_pipeline_set_frame_pinned(vo_pipeline* p, int seq, int idx, const uint8_t* pinned_img);
int vo_pipeline_frame_uploaded(vo_pipeline* p, int idx, int wait);
/* Hint: frame slot idx will be the `next_idx` of the coming vo_pipeline_submit.  Its pyramid (klt.py:233-249 builds it
 * inside cv2.calcOpticalFlowPyrLK, per call) is built now, behind the tracker of the step submitted last, so the coming
 * step's tracker does not start behind a pyramid that its own, later, submit enqueues.  Results do not depend on the hint;
 * a wrong one costs one wasted pyramid.  KLT tracker mode; a no-op otherwise and before the first step.                 */
int vo_pipeline_prepare(vo_pipeline* p, int idx);
int vo_host_alloc(vo_ctx* ctx, size_t bytes, void** out);
int vo_host_free(vo_ctx* ctx, void* p);          /* ctx may be NULL */
int vo_pipeline_set_state_seq(vo_pipeline* p, int seq, int idx, int n, const double* kp, const uint8_t* state,
                              const double* landmarks, const double* tracks, const double* poses,
                              const double* T_wc, const double* T_cw, const double* T_wc_prev,
                              const double* T_cw_prev, int num_features);
int vo_pipeline_get_state_seq(vo_pipeline* p, int seq, int32_t* n_out, double* kp, uint8_t* state,
                              uint8_t* candidate_mask, double* landmarks, double* tracks, double* poses,
                              double* T_wc, double* T_wc_prev, vo_ransac_state* rs, int32_t* num_features);
int vo_pipeline_get_rng_seq(vo_pipeline* p, int seq, vo_pcg64* rng);
int vo_pipeline_collect_all(vo_pipeline* p, vo_step_result* outs);

/* ---- shared map over RCCL ------------------------------------------------------------------
 * The reference is one process and one thread (README.md:49); frame streams shard at sequence granularity (one
 * process per GPU) and the ranks share {pose, landmarks} through ONE collective: an all-gather of a
 * fixed-size record per rank, [T_cw 4x4 (16) | n (1) | n landmarks x 3, n <= cap] float64 = 17 + 3 cap doubles
 * (what vo_pipeline_export_state_post writes).  A host that is not PyTorch (bench.py drives the same exchange through
 * torch.distributed) makes a communicator from an id rank 0 creates and hands to the other ranks by its own means
 * (MPI, a file, a socket):
 *   vo_comm_unique_id   128 bytes (ncclUniqueId)
 *   vo_comm_create      ncclCommInitRank on the context's device; collective: every rank calls it
 *   vo_allgather_state_dev  ncclAllGather of doubles_per_rank doubles per rank, device pointers, asynchronous on
 *                       `stream` (NULL: the context's).  One or several frames' records per call: records of k frames
 *                       posted back to back are one message of k (17 + 3 cap) doubles.
 *   vo_allgather_state  host arrays, one record, synchronous: pose16 (T_cw), landmarks n*3 -> all: world records.
 * RCCL is opened at the first call (the process's own copy if it has one), not linked.                          */
#define VO_COMM_ID_BYTES 128
typedef struct vo_comm vo_comm;
int vo_comm_unique_id(vo_ctx* ctx, void* id128);
int vo_comm_create(vo_ctx* ctx, int world, int rank, const void* id128, vo_comm** out);
void vo_comm_destroy(vo_comm* c);
int vo_comm_world(const vo_comm* c);
int vo_allgather_state_dev(vo_ctx* ctx, vo_comm* c, const double* d_records, size_t doubles_per_rank, double* d_all,
                           void* stream);
int vo_allgather_state(vo_ctx* ctx, vo_comm* c, const double* pose16, const double* landmarks, int n, int cap,
                       double* all);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
