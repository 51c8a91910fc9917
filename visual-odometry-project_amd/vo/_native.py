"""ctypes binding of libvo_hip.so (include/vo_hip.h).

There is no CPU fallback: if the library is missing, or no GPU is present when a
context is created, the caller gets an exception, never a silent slow path.
"""
import ctypes as C
import atexit
import os
import threading
import weakref

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvo_hip.so")

VO_OK, VO_EINVAL, VO_ENOMEM, VO_EHIP, VO_ECAPACITY = 0, -1, -2, -3, -4
_CODES = {VO_EINVAL: "VO_EINVAL", VO_ENOMEM: "VO_ENOMEM", VO_EHIP: "VO_EHIP", VO_ECAPACITY: "VO_ECAPACITY"}

# kernel ids (vo_hip.h)
K_HARRIS_RESPONSE, K_NMS_CANDIDATES, K_NMS_THRESHOLD, K_NMS_COMPACT, K_NMS_SELECT = 0, 1, 2, 3, 4
K_PATCH_DESC, K_PYR_DOWN, K_KLT_TRACK, K_DLT, K_P3P_SOLVE, K_P3P_SCORE, K_REPROJ, K_MATCH = 5, 6, 7, 8, 9, 10, 11, 12
K_COUNT = 32


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_CODES.get(code, "VO_E?"), code, msg))
        self.code = code


_lib = None
_lib_lock = threading.Lock()


class Pcg64(C.Structure):
    """NumPy PCG64 bit-generator state (vo_pcg64)."""
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64),
                ("has_uint32", C.c_uint32), ("uinteger", C.c_uint32)]

    @classmethod
    def from_generator(cls, gen):
        st = gen.bit_generator.state
        assert st["bit_generator"] == "PCG64"
        s, inc = st["state"]["state"], st["state"]["inc"]
        m = (1 << 64) - 1
        return cls(s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"])

    def to_generator(self, gen):
        st = gen.bit_generator.state
        st["state"]["state"] = (self.state_hi << 64) | self.state_lo
        st["state"]["inc"] = (self.inc_hi << 64) | self.inc_lo
        st["has_uint32"] = int(self.has_uint32)
        st["uinteger"] = int(self.uinteger)
        gen.bit_generator.state = st


class RansacState(C.Structure):
    _fields_ = [("outlier_ratio", C.c_double), ("confidence", C.c_double), ("max_iterations", C.c_int64),
                ("n_iterations", C.c_int64), ("s", C.c_int32), ("adaptive", C.c_int32)]


class PipelineConfig(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("n_frames", C.c_int32),
                ("n_keypoints", C.c_int32), ("harris_patch", C.c_int32), ("nms_radius", C.c_int32),
                ("harris_kappa", C.c_double),
                ("klt_win", C.c_int32), ("klt_max_level", C.c_int32), ("klt_max_iter", C.c_int32), ("hyp", C.c_int32),
                ("klt_eps", C.c_double), ("klt_min_eig", C.c_double), ("klt_err_threshold", C.c_double),
                ("p3p_thr_sq", C.c_double), ("ransac_outlier_ratio", C.c_double), ("ransac_confidence", C.c_double),
                ("ransac_max_iterations", C.c_int64),
                ("K", C.c_double * 9), ("Kinv", C.c_double * 9), ("refine_iters", C.c_int32), ("feature_cap", C.c_int32),
                ("bearing_threshold", C.c_double), ("redetect_fraction", C.c_double),
                ("debug_fault_every", C.c_int32), ("redetect_start_pose", C.c_int32), ("detect_margin", C.c_double),
                ("debug_never_detect", C.c_int32), ("detect_losses", C.c_double), ("sequences", C.c_int32),
                ("tracker_mode", C.c_int32), ("sift_cap", C.c_int32), ("match_ratio", C.c_double)]


class StepResult(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("n_tracked", C.c_int32), ("n_inliers", C.c_int32),
                ("best_index", C.c_int32), ("hyp_valid", C.c_int32), ("ransac_iterations", C.c_int64),
                ("draws_consumed", C.c_int32), ("refine_iterations", C.c_int32),
                ("R_refined", C.c_double * 9), ("t_refined", C.c_double * 3), ("refine_cost", C.c_double),
                ("n_features_in", C.c_int32), ("redetected", C.c_int32), ("n_triangulated", C.c_int32),
                ("n_candidates", C.c_int32), ("n_dropped", C.c_int32), ("n_landmarks", C.c_int32),
                ("fault", C.c_int32), ("recovered", C.c_int32), ("detector_ran", C.c_int32), ("reserved", C.c_int32),
                ("raw_pos", C.c_uint64), ("T_wc", C.c_double * 12),
                ("ts", C.c_uint64 * 8), ("seq_head", C.c_uint32), ("seq_tail", C.c_uint32)]

    def pose_world_cam(self):
        """State.curr_pose after the step: camera-to-world, 4x4."""
        return np.vstack([np.array(self.T_wc).reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])


_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_SIGS = {
    "vo_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "vo_destroy": (None, [_vp]),
    "vo_last_error": (C.c_char_p, [_vp]),
    "vo_version": (_i, []),
    "vo_sync": (_i, [_vp]),
    "vo_stream": (_vp, [_vp]),
    "vo_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vo_dev_free": (_i, [_vp, _vp]),
    "vo_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "vo_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "vo_prof_enable": (_i, [_vp, _i]),
    "vo_prof_set_sampling": (_i, [_vp, _i]),
    "vo_prof_disable": (_i, [_vp]),
    "vo_prof_read": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_int64)]),
    "vo_prof_reset": (_i, [_vp]),
    "vo_kernel_name": (C.c_char_p, [_i]),
    "vo_harris_response": (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    "vo_harris_keypoints": (_i, [_vp, _vp, _i, _i, _i, _d, _i, _i, _vp, _vp]),
    "vo_nms_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vo_harris_keypoints_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _i, _i, _vp, _vp]),
    "vo_harris_response_dev": (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    "vo_nms_keypoints_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vo_patch_descriptors": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "vo_patch_descriptors_dev": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "vo_klt_num_levels": (_i, [_i, _i, _i, _i]),
    "vo_pyramid_bytes": (_sz, [_i, _i, _i]),
    "vo_pyr_down": (_i, [_vp, _vp, _i, _i, _vp]),
    "vo_pyramid_build_dev": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vo_klt_track": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp]),
    "vo_klt_track_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _d, _d, _vp, _vp, _vp]),
    "vo_triangulate_dlt": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "vo_triangulate_dlt_dev": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "vo_p3p_hypotheses": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "vo_p3p_hypotheses_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "vo_reproj_inliers": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp]),
    "vo_reproj_inliers_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _d, _vp, _vp]),
    "vo_refine_pose": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(_d)]),
    "vo_refine_pose_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "vo_match_knn2_ratio": (_i, [_vp, _vp, _i, _vp, _i, _i, _d, _vp, _vp]),
    "vo_knn2_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "vo_good_features": (_i, [_vp, _vp, _i, _i, _vp, _i, _d, _d, _i, _vp, _vp]),
    "vo_min_eigen_map": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vo_sift": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "vo_sift_capacity": (_i, [_i, _i]),
    "vo_sift_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "vo_fundamental_hypotheses": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _d, _vp, _vp, _vp]),
    "vo_fundamental_fit": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp]),
    "vo_essential_decompose": (_i, [_vp, _vp, _vp]),
    "vo_relative_pose": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vo_comm_unique_id": (_i, [_vp, _vp]),
    "vo_comm_create": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "vo_comm_destroy": (None, [_vp]),
    "vo_comm_world": (_i, [_vp]),
    "vo_allgather_state_dev": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "vo_allgather_state": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "vo_rng_choice": (_i, [_vp, _i, _i, _i, _vp]),
    "vo_ransac_num_iterations": (C.c_int64, [_d, _d, _i]),
    "vo_ransac_replay": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "vo_record_seal": (None, [_vp, C.c_uint]),
    "vo_record_check": (_i, [_vp, C.c_uint]),
    "vo_pipeline_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "vo_pipeline_set_descriptors": (_i, [_vp, _vp, _i]),
    "vo_pipeline_checkpoint": (_i, [_vp]),
    "vo_pipeline_rewind": (_i, [_vp]),
    "vo_pipeline_destroy": (None, [_vp]),
    "vo_pipeline_release_cached": (None, []),
    "vo_pipeline_release_cached_at_exit": (_i, []),
    "vo_pipeline_set_frame": (_i, [_vp, _i, _vp]),
    "vo_pipeline_seed": (_i, [_vp, _vp]),
    "vo_pipeline_get_rng": (_i, [_vp, _vp]),
    "vo_pipeline_set_state": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "vo_pipeline_feature_cap": (_i, [_vp]),
    "vo_pipeline_get_state": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vo_pipeline_get_detection": (_i, [_vp, _vp]),
    "vo_pipeline_step": (_i, [_vp, _i, _i, _vp]),
    "vo_pipeline_submit": (_i, [_vp, _i, _i]),
    "vo_pipeline_collect": (_i, [_vp, _vp]),
    "vo_pipeline_bookkeeping": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "vo_pipeline_export_state_post": (_i, [_vp, _vp, _i, _vp]),
    "vo_pipeline_export_state_join": (_i, [_vp, _vp]),
    "vo_pipeline_prof_read": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_int64)]),
    "vo_pipeline_prof_reset": (_i, [_vp]),
    "vo_pipeline_ransac_bound": (C.c_int64, [_vp, _d]),
    "vo_pipeline_sequences": (_i, [_vp]),
    "vo_pipeline_set_frame_seq": (_i, [_vp, _i, _i, _vp]),
    "vo_pipeline_set_frame_pinned": (_i, [_vp, _i, _i, _vp]),
    "vo_pipeline_frame_uploaded": (_i, [_vp, _i, _i]),
    "vo_pipeline_prepare": (_i, [_vp, _i]),
    "vo_host_alloc": (_i, [_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vo_host_free": (_i, [_vp, _vp]),
    "vo_pipeline_set_state_seq": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "vo_pipeline_get_state_seq": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vo_pipeline_get_rng_seq": (_i, [_vp, _i, _vp]),
    "vo_pipeline_collect_all": (_i, [_vp, _vp]),
    "vo_pipeline_export_state_post_seq": (_i, [_vp, _i, _vp, _i, _vp]),
}


def lib_path():
    return _LIB_PATH


def load():
    """Load libvo_hip.so; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise ImportError(
                    "libvo_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (there is no CPU fallback)" % _LIB_PATH)
            lib = C.CDLL(_LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Context:
    """One HIP stream + device workspace (vo_ctx).  Not thread-safe."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.vo_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != VO_OK:
            raise VoError(rc, "vo_create(device=%d) failed: no usable MI355X/HIP device (no CPU fallback)" % device)
        self._h = h
        self.device = device
        self._pipelines = weakref.WeakSet()      # pipelines built on this context: closed before it is
        self._pinned_ranges = []                 # live pinned_empty allocations: (first byte, one past the last, owner)

    def close(self):
        if getattr(self, "_h", None):
            for p in list(self._pipelines):
                p.close()
            self._lib.vo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != VO_OK:
            raise VoError(rc, self._lib.vo_last_error(self._h).decode("utf-8", "replace"))

    # ---- plumbing ----
    def sync(self):
        self._chk(self._lib.vo_sync(self._h))

    @property
    def stream(self):
        return self._lib.vo_stream(self._h)

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._lib.vo_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def free(self, p):
        self._chk(self._lib.vo_dev_free(self._h, C.c_void_p(p)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self._lib.vo_dev_upload(self._h, C.c_void_p(dptr), _ptr(arr), arr.nbytes))

    def download(self, dptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(self._lib.vo_dev_download(self._h, _ptr(out), C.c_void_p(dptr), out.nbytes))
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 1))
        self.upload(p, arr)
        return p

    def prof_enable(self, kernel_id=-1):
        self._chk(self._lib.vo_prof_enable(self._h, int(kernel_id)))

    def prof_set_sampling(self, every):
        self._chk(self._lib.vo_prof_set_sampling(self._h, int(every)))

    def prof_disable(self):
        self._chk(self._lib.vo_prof_disable(self._h))

    def prof_reset(self):
        self._chk(self._lib.vo_prof_reset(self._h))

    def prof_read(self, kernel_id):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self._lib.vo_prof_read(self._h, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_name(self, kernel_id):
        return self._lib.vo_kernel_name(int(kernel_id)).decode()

    # ---- Harris / NMS / descriptors (host arrays) ----
    def harris_response(self, img, patch=9, kappa=0.09):
        img = _c(img, np.uint8)
        assert img.ndim == 2
        H, W = img.shape
        out = np.empty((H, W), np.float64)
        self._chk(self._lib.vo_harris_response(self._h, _ptr(img), H, W, int(patch), float(kappa), _ptr(out)))
        return out

    def harris_keypoints(self, img, patch=9, kappa=0.09, num_keypoints=1000, r=5, want_scores=False):
        img = _c(img, np.uint8)
        assert img.ndim == 2
        H, W = img.shape
        kp = np.empty((num_keypoints, 2), np.float64)
        sc = np.empty((H, W), np.float64) if want_scores else None
        self._chk(self._lib.vo_harris_keypoints(self._h, _ptr(img), H, W, int(patch), float(kappa),
                                                int(num_keypoints), int(r), _ptr(kp), _ptr(sc)))
        return (kp, sc) if want_scores else kp

    def harris_keypoints_batch(self, imgs, patch=9, kappa=0.09, num_keypoints=1000, r=5, want_scores=False):
        """(S, H, W) uint8 -> (S, N, 2) keypoints [, (S, H, W) score maps]: S frames in one set of launches."""
        imgs = _c(imgs, np.uint8)
        assert imgs.ndim == 3
        S, H, W = imgs.shape
        kp = np.empty((S, num_keypoints, 2), np.float64)
        sc = np.empty((S, H, W), np.float64) if want_scores else None
        self._chk(self._lib.vo_harris_keypoints_batch(self._h, _ptr(imgs), S, H, W, int(patch), float(kappa),
                                                      int(num_keypoints), int(r), _ptr(kp), _ptr(sc)))
        return (kp, sc) if want_scores else kp

    def nms_keypoints(self, scores, num_keypoints, r):
        scores = _c(scores, np.float64)
        H, W = scores.shape
        kp = np.empty((num_keypoints, 2), np.float64)
        self._chk(self._lib.vo_nms_keypoints(self._h, _ptr(scores), H, W, int(num_keypoints), int(r), _ptr(kp)))
        return kp

    def patch_descriptors(self, img, kp_xy, r=9):
        img = _c(img, np.uint8)
        kp = _c(np.asarray(kp_xy).reshape(-1, 2), np.float64)
        H, W = img.shape
        n = kp.shape[0]
        d = (2 * r + 1) ** 2
        out = np.empty((n, d), np.float64)
        self._chk(self._lib.vo_patch_descriptors(self._h, _ptr(img), H, W, _ptr(kp), n, int(r), _ptr(out)))
        return out

    # ---- matching / corners ----
    def match_knn2_ratio(self, q, t, ratio):
        q, t = np.asarray(q), np.asarray(t)
        if len(q) == 0 or len(t) == 0:
            return np.empty((0, 2), np.int64)
        q = _c(q.reshape(len(q), -1), np.float32)
        t = _c(t.reshape(len(t), -1), np.float32)
        assert q.shape[1] == t.shape[1]
        pairs = np.empty((max(q.shape[0], 1), 2), np.int32)
        n = C.c_int32(0)
        self._chk(self._lib.vo_match_knn2_ratio(self._h, _ptr(q), q.shape[0], _ptr(t), t.shape[0],
                                                max(q.shape[1], 1), float(ratio), _ptr(pairs), C.byref(n)))
        return pairs[: n.value].astype(np.int64)

    def pinned_empty(self, shape, dtype=np.uint8):
        """A NumPy array in pinned host memory (vo_host_alloc): what Pipeline.set_frame(..., pinned=True) uploads from
        without a staging copy.  The memory lives as long as the array (and every view of it) does."""
        shape = tuple(int(v) for v in np.atleast_1d(shape))
        dt = np.dtype(dtype)
        nbytes = max(int(np.prod(shape)) * dt.itemsize, 1)
        q = C.c_void_p()
        self._chk(self._lib.vo_host_alloc(self._h, nbytes, C.byref(q)))
        buf = (C.c_uint8 * nbytes).from_address(q.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        lib, addr = self._lib, q.value
        weakref.finalize(buf, lambda: lib.vo_host_free(None, C.c_void_p(addr)))     # (the context may be gone by then)
        arr.flags.writeable = True
        self._pinned_ranges.append((addr, addr + nbytes, weakref.ref(buf)))
        return arr

    def is_pinned(self, arr):
        """True when arr's bytes lie inside a live pinned_empty allocation of this context."""
        a = arr.__array_interface__["data"][0]
        b = a + arr.nbytes
        self._pinned_ranges = [r for r in self._pinned_ranges if r[2]() is not None]
        return any(lo <= a and b <= hi for lo, hi, _ in self._pinned_ranges)

    def good_features(self, img, mask=None, max_corners=500, quality=0.01, min_distance=8, block_size=7):
        img = _c(img, np.uint8)
        H, W = img.shape
        m = None if mask is None else _c(mask, np.uint8)
        cap = max_corners if max_corners > 0 else (H * W + 3) // 4 + 64     # (the C side's candidate capacity)
        xy = np.empty((cap, 2), np.float32)
        n = C.c_int32(0)
        self._chk(self._lib.vo_good_features(self._h, _ptr(img), H, W, _ptr(m), int(max_corners), float(quality),
                                             float(min_distance), int(block_size), _ptr(xy), C.byref(n)))
        return xy[: n.value].copy()

    def sift(self, img, cap=None):
        """(kp (n, 6) float32: x, y, size, angle, response, octave; desc (n, 128) float32).
        cap=None keeps every keypoint, as cv2.SIFT_create() does; an integer keeps the strongest `cap`."""
        img = _c(img, np.uint8)
        H, W = img.shape
        rows = int(cap) if cap else self._lib.vo_sift_capacity(H, W)
        kp = np.empty((rows, 6), np.float32)
        desc = np.empty((rows, 128), np.float32)
        n = C.c_int32(0)
        self._chk(self._lib.vo_sift(self._h, _ptr(img), H, W, int(cap) if cap else 0, _ptr(kp), _ptr(desc), C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    def min_eigen_map(self, img, block_size=7):
        img = _c(img, np.uint8)
        H, W = img.shape
        out = np.empty((H, W), np.float32)
        self._chk(self._lib.vo_min_eigen_map(self._h, _ptr(img), H, W, int(block_size), _ptr(out)))
        return out

    # ---- KLT ----
    def klt_num_levels(self, H, W, win, max_level):
        return self._lib.vo_klt_num_levels(int(H), int(W), int(win), int(max_level))

    def pyramid_bytes(self, H, W, n_levels):
        return self._lib.vo_pyramid_bytes(int(H), int(W), int(n_levels))

    def pyr_down(self, img):
        img = _c(img, np.uint8)
        H, W = img.shape
        out = np.empty(((H + 1) // 2, (W + 1) // 2), np.uint8)
        self._chk(self._lib.vo_pyr_down(self._h, _ptr(img), H, W, _ptr(out)))
        return out

    def klt_track(self, prev, nxt, prev_xy, win=17, max_level=2, max_iter=10, eps=0.03, min_eig=1e-4):
        prev = _c(prev, np.uint8)
        nxt = _c(nxt, np.uint8)
        assert prev.shape == nxt.shape and prev.ndim == 2
        H, W = prev.shape
        pts = _c(np.asarray(prev_xy).reshape(-1, 2), np.float32)
        n = pts.shape[0]
        out = np.empty((n, 2), np.float32)
        status = np.empty(n, np.uint8)
        err = np.empty(n, np.float32)
        self._chk(self._lib.vo_klt_track(self._h, _ptr(prev), _ptr(nxt), H, W, _ptr(pts), n, int(win), int(max_level),
                                         int(max_iter), float(eps), float(min_eig), _ptr(out), _ptr(status), _ptr(err)))
        return out, status, err

    # ---- DLT ----
    def triangulate_dlt(self, x1, x2, C1, C2):
        x1 = _c(np.asarray(x1).reshape(-1, 2), np.float64)
        x2 = _c(np.asarray(x2).reshape(-1, 2), np.float64)
        n = x1.shape[0]
        C1 = _c(C1, np.float64)
        C2 = _c(C2, np.float64).reshape(3, 4)
        per_point = 1 if C1.ndim == 3 else 0
        assert C1.shape[-2:] == (3, 4) and (not per_point or C1.shape[0] == n)
        X = np.empty((n, 3), np.float64)
        self._chk(self._lib.vo_triangulate_dlt(self._h, _ptr(x1), _ptr(x2), n, _ptr(C1), per_point, _ptr(C2), _ptr(X)))
        return X

    # ---- two-view bootstrap ----
    def fundamental_hypotheses(self, p1, p2, samples, threshold, normalize_samples=False, error_kind=0, want_masks=False):
        """8-point F of every sample (samples (Hyp, 8)) and its inlier count over all correspondences:
        (F (Hyp, 3, 3), counts (Hyp,)[, masks (Hyp, N) bool]).  error_kind 0: (p2^T F p1)^2; 1: squared epipolar distance."""
        p1 = _c(np.asarray(p1).reshape(-1, 2), np.float64)
        p2 = _c(np.asarray(p2).reshape(-1, 2), np.float64)
        samples = _c(np.asarray(samples).reshape(-1, 8), np.int32)
        n, h = p1.shape[0], samples.shape[0]
        F = np.empty((h, 3, 3), np.float64)
        counts = np.empty(h, np.int32)
        words = (n + 63) // 64
        masks = np.empty((h, words), np.uint64) if want_masks else None
        self._chk(self._lib.vo_fundamental_hypotheses(self._h, _ptr(p1), _ptr(p2), n, _ptr(samples), h,
                                                      int(bool(normalize_samples)), int(error_kind), float(threshold),
                                                      _ptr(F), _ptr(counts), _ptr(masks)))
        if want_masks == "packed":          # (rows of 64-bit words; Context.unpack_mask(row, n) opens one)
            return F, counts, masks
        if want_masks:
            bits = np.unpackbits(masks.view(np.uint8).reshape(h, words * 8), axis=1, bitorder="little")[:, :n]
            return F, counts, bits.astype(bool)
        return F, counts

    @staticmethod
    def unpack_mask(row, n):
        return np.unpackbits(np.ascontiguousarray(row).view(np.uint8), bitorder="little")[:n].astype(bool)

    def fundamental_fit(self, p1, p2, mask=None, normalize=True):
        """The 8-point fit over all (masked) correspondences: F (3, 3)."""
        p1 = _c(np.asarray(p1).reshape(-1, 2), np.float64)
        p2 = _c(np.asarray(p2).reshape(-1, 2), np.float64)
        m = None if mask is None else _c(np.asarray(mask).reshape(-1), np.uint8)
        F = np.empty((3, 3), np.float64)
        self._chk(self._lib.vo_fundamental_fit(self._h, _ptr(p1), _ptr(p2), p1.shape[0], _ptr(m), int(bool(normalize)), _ptr(F)))
        return F

    def essential_decompose(self, E):
        E = _c(np.asarray(E).reshape(3, 3), np.float64)
        M4 = np.empty((4, 3, 4), np.float64)
        self._chk(self._lib.vo_essential_decompose(self._h, _ptr(E), _ptr(M4)))
        return M4

    def relative_pose(self, x1, x2, K1, K2, F, inliers=None):
        """(M (3, 4), X (N, 3), mask (N,) bool, M4 (4, 3, 4)): the cheirality vote over the four decompositions of
        E = K2^T F K1 and the winner's triangulation of all correspondences."""
        x1 = _c(np.asarray(x1).reshape(-1, 2), np.float64)
        x2 = _c(np.asarray(x2).reshape(-1, 2), np.float64)
        K1 = _c(np.asarray(K1).reshape(3, 3), np.float64)
        K2 = _c(np.asarray(K2).reshape(3, 3), np.float64)
        F = _c(np.asarray(F).reshape(3, 3), np.float64)
        inl = None if inliers is None else _c(np.asarray(inliers).reshape(-1), np.uint8)
        n = x1.shape[0]
        M, X, mask, M4 = np.empty((3, 4)), np.empty((n, 3)), np.empty(n, np.uint8), np.empty((4, 3, 4))
        self._chk(self._lib.vo_relative_pose(self._h, _ptr(x1), _ptr(x2), n, _ptr(inl), _ptr(K1), _ptr(K2), _ptr(F), _ptr(M),
                                             _ptr(X), _ptr(mask), _ptr(M4)))
        return M, X, mask.astype(bool), M4

    # ---- P3P ----
    def p3p_hypotheses(self, X, x, K, samples, thr_sq, want_masks=False):
        X = _c(np.asarray(X).reshape(-1, 3), np.float64)
        x = _c(np.asarray(x).reshape(-1, 2), np.float64)
        K = _c(K, np.float64).reshape(3, 3)
        samples = _c(np.asarray(samples).reshape(-1, 4), np.int32)
        n, h = X.shape[0], samples.shape[0]
        R = np.empty((h, 3, 3), np.float64)
        t = np.empty((h, 3), np.float64)
        valid = np.empty(h, np.uint8)
        counts = np.empty(h, np.int32)
        words = (n + 63) // 64
        masks = np.empty((h, words), np.uint64) if want_masks else None
        self._chk(self._lib.vo_p3p_hypotheses(self._h, _ptr(X), _ptr(x), n, _ptr(K), _ptr(samples), h, float(thr_sq),
                                              _ptr(R), _ptr(t), _ptr(valid), _ptr(counts), _ptr(masks)))
        if want_masks:
            bits = np.unpackbits(masks.view(np.uint8).reshape(h, words * 8), axis=1, bitorder="little")[:, :n]
            return R, t, valid, counts, bits.astype(bool)
        return R, t, valid, counts

    def refine_pose(self, X, x, K, R0, t0, inlier_mask=None, max_iter=20):
        """Least-squares pose from (R0, t0) over the (masked) correspondences: (R, t (3,), iterations, cost)
        -- the minimiser p3p.py:188-213 asks SciPy for (vo_refine_pose)."""
        X = _c(np.asarray(X).reshape(-1, 3), np.float64)
        x = _c(np.asarray(x).reshape(-1, 2), np.float64)
        K = _c(K, np.float64)
        R0 = _c(np.asarray(R0).reshape(3, 3), np.float64)
        t0 = _c(np.asarray(t0).reshape(3), np.float64)
        m = None if inlier_mask is None else _c(np.asarray(inlier_mask).reshape(-1), np.uint8)
        R, t = np.empty((3, 3)), np.empty(3)
        it, cost = C.c_int32(), C.c_double()
        self._chk(self._lib.vo_refine_pose(self._h, _ptr(X), _ptr(x), len(X), _ptr(K), _ptr(m), _ptr(R0), _ptr(t0),
                                           int(max_iter), _ptr(R), _ptr(t), C.byref(it), C.byref(cost)))
        return R, t, it.value, cost.value

    def reproj_inliers(self, X, x, K, R, t, thr_sq, want_err=False):
        X = _c(np.asarray(X).reshape(-1, 3), np.float64)
        x = _c(np.asarray(x).reshape(-1, 2), np.float64)
        K = _c(K, np.float64).reshape(3, 3)
        R = _c(R, np.float64).reshape(3, 3)
        t = _c(np.asarray(t).reshape(3), np.float64)
        n = X.shape[0]
        mask = np.empty(n, np.uint8)
        err = np.empty(n, np.float64) if want_err else None
        self._chk(self._lib.vo_reproj_inliers(self._h, _ptr(X), _ptr(x), n, _ptr(K), _ptr(R), _ptr(t), float(thr_sq),
                                              _ptr(mask), _ptr(err)))
        return (mask.astype(bool), err) if want_err else mask.astype(bool)

    # ---- device-pointer variants (async on the context's stream) ----
    def pyramid_build_dev(self, d_img, H, W, n_levels, d_pyr):
        self._chk(self._lib.vo_pyramid_build_dev(self._h, C.c_void_p(d_img), H, W, int(n_levels), C.c_void_p(d_pyr)))

    def klt_track_dev(self, d_prev, d_prev_pyr, d_next, d_next_pyr, H, W, n_levels, d_prev_xy, N, win, max_iter, eps,
                      min_eig, d_next_xy, d_status, d_err):
        self._chk(self._lib.vo_klt_track_dev(self._h, C.c_void_p(d_prev), C.c_void_p(d_prev_pyr), C.c_void_p(d_next),
                                             C.c_void_p(d_next_pyr), H, W, int(n_levels), C.c_void_p(d_prev_xy), int(N),
                                             int(win), int(max_iter), float(eps), float(min_eig),
                                             C.c_void_p(d_next_xy), C.c_void_p(d_status), C.c_void_p(d_err)))

    def triangulate_dlt_dev(self, d_x1, d_x2, n, d_C1, per_point, d_C2, d_X):
        self._chk(self._lib.vo_triangulate_dlt_dev(self._h, C.c_void_p(d_x1), C.c_void_p(d_x2), int(n),
                                                   C.c_void_p(d_C1), int(per_point), C.c_void_p(d_C2), C.c_void_p(d_X)))

    def p3p_hypotheses_dev(self, d_X, d_x, N, K, d_samples, Hyp, thr_sq, d_R, d_t, d_valid, d_counts, d_masks):
        K = _c(K, np.float64).reshape(3, 3)
        self._chk(self._lib.vo_p3p_hypotheses_dev(self._h, C.c_void_p(d_X), C.c_void_p(d_x), int(N), _ptr(K),
                                                  C.c_void_p(d_samples), int(Hyp), float(thr_sq), C.c_void_p(d_R),
                                                  C.c_void_p(d_t), C.c_void_p(d_valid), C.c_void_p(d_counts),
                                                  C.c_void_p(d_masks)))

    def reproj_inliers_dev(self, d_X, d_x, N, K, d_Rt, thr_sq, d_mask, d_err):
        K = _c(K, np.float64).reshape(3, 3)
        self._chk(self._lib.vo_reproj_inliers_dev(self._h, C.c_void_p(d_X), C.c_void_p(d_x), int(N), _ptr(K),
                                                  C.c_void_p(d_Rt), float(thr_sq), C.c_void_p(d_mask), C.c_void_p(d_err)))

    def harris_response_dev(self, d_img, H, W, patch, kappa, d_scores):
        self._chk(self._lib.vo_harris_response_dev(self._h, C.c_void_p(d_img), H, W, int(patch), float(kappa),
                                                   C.c_void_p(d_scores)))

    def nms_keypoints_dev(self, d_scores, H, W, N, r, d_kp):
        self._chk(self._lib.vo_nms_keypoints_dev(self._h, C.c_void_p(d_scores), H, W, int(N), int(r),
                                                 C.c_void_p(d_kp)))

    def patch_descriptors_dev(self, d_img, H, W, d_kp, N, r, d_desc):
        self._chk(self._lib.vo_patch_descriptors_dev(self._h, C.c_void_p(d_img), H, W, C.c_void_p(d_kp), int(N),
                                                     int(r), C.c_void_p(d_desc)))


class Comm:
    """RCCL communicator of the shared-map exchange for hosts without torch.distributed (vo_comm_*): rank 0 makes the
    id (`Comm.unique_id(ctx)`, 128 bytes) and hands it to the other ranks by its own means; every rank then constructs
    Comm(ctx, world, rank, id)."""

    @staticmethod
    def unique_id(ctx):
        buf = C.create_string_buffer(128)
        ctx._chk(ctx._lib.vo_comm_unique_id(ctx._h, buf))
        return buf.raw

    def __init__(self, ctx, world, rank, uid):
        assert len(uid) == 128
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        h = C.c_void_p()
        ctx._chk(ctx._lib.vo_comm_create(ctx._h, self.world, self.rank, C.c_char_p(uid), C.byref(h)))
        self._h = h

    def allgather_state(self, T_cw, landmarks, cap):
        """One record per rank, host arrays, synchronous: (world, 17 + 3 cap) float64."""
        T = _c(np.asarray(T_cw).reshape(16), np.float64)
        lm = _c(np.asarray(landmarks).reshape(-1, 3), np.float64)
        out = np.empty((self.world, 17 + 3 * int(cap)), np.float64)
        self.ctx._chk(self.ctx._lib.vo_allgather_state(self.ctx._h, self._h, _ptr(T), _ptr(lm), lm.shape[0], int(cap), _ptr(out)))
        return out

    def allgather_dev(self, d_records, doubles_per_rank, d_all, stream=None):
        self.ctx._chk(self.ctx._lib.vo_allgather_state_dev(self.ctx._h, self._h, C.c_void_p(d_records), int(doubles_per_rank),
                                                           C.c_void_p(d_all), C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.vo_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rng_choice(pcg, pop, s, count):
    """`count` draws of Generator.choice(arange(pop), replace=False, size=s); advances `pcg` (host only)."""
    out = np.empty((count, s), np.int32)
    rc = load().vo_rng_choice(C.byref(pcg), int(pop), int(s), int(count), _ptr(out))
    if rc != VO_OK:
        raise VoError(rc, "vo_rng_choice(pop=%d, s=%d)" % (pop, s))
    return out


def ransac_num_iterations(confidence, outlier_ratio, s):
    return int(load().vo_ransac_num_iterations(float(confidence), float(outlier_ratio), int(s)))


from vo._pipeline import Pipeline  # noqa: E402,F401  (device-resident frame loop, vo_pipeline_*)


_default_ctx = None


def release_cached():
    """vo_pipeline_release_cached: destroy the side streams / workspace kept from closed pipelines."""
    if _lib is not None:
        _lib.vo_pipeline_release_cached()


def _release_cached_at_exit():
    if _lib is not None and _lib.vo_pipeline_release_cached_at_exit():
        _lib.vo_pipeline_release_cached()


atexit.register(_release_cached_at_exit)        # (before the interpreter and the HIP runtime are torn down)


def set_default_context(ctx):
    """Make `ctx` the context the drop-in classes use when none is passed to them."""
    global _default_ctx
    _default_ctx = ctx


def default_context():
    """Process-wide context on the device selected by LOCAL_RANK (or device 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("VO_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx
