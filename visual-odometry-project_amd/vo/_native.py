"""ctypes binding of libvo_hip.so (include/vo_hip.h).

There is no CPU fallback: if the library is missing, or no GPU is present when a
context is created, the caller gets an exception, never a silent slow path.
"""
import ctypes as C
import os
import threading

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
    "vo_prof_disable": (_i, [_vp]),
    "vo_prof_read": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_int64)]),
    "vo_prof_reset": (_i, [_vp]),
    "vo_kernel_name": (C.c_char_p, [_i]),
    "vo_harris_response": (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    "vo_harris_keypoints": (_i, [_vp, _vp, _i, _i, _i, _d, _i, _i, _vp, _vp]),
    "vo_nms_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vo_harris_response_dev": (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    "vo_nms_keypoints_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vo_patch_descriptors": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "vo_patch_descriptors_dev": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
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

    def close(self):
        if getattr(self, "_h", None):
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

    # ---- device-pointer variants (async on the context's stream) ----
    def harris_response_dev(self, d_img, H, W, patch, kappa, d_scores):
        self._chk(self._lib.vo_harris_response_dev(self._h, C.c_void_p(d_img), H, W, int(patch), float(kappa),
                                                   C.c_void_p(d_scores)))

    def nms_keypoints_dev(self, d_scores, H, W, N, r, d_kp):
        self._chk(self._lib.vo_nms_keypoints_dev(self._h, C.c_void_p(d_scores), H, W, int(N), int(r),
                                                 C.c_void_p(d_kp)))

    def patch_descriptors_dev(self, d_img, H, W, d_kp, N, r, d_desc):
        self._chk(self._lib.vo_patch_descriptors_dev(self._h, C.c_void_p(d_img), H, W, C.c_void_p(d_kp), int(N),
                                                     int(r), C.c_void_p(d_desc)))


_default_ctx = None


def default_context():
    """Process-wide context on the device selected by LOCAL_RANK (or device 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("VO_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx
