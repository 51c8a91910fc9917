"""ctypes access to the C oracle (oracle/csrc -> oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.run(["make"], cwd=os.path.join(_HERE, "csrc"), check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "csrc")
        newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def p3p_solve(X4, x4, K):
    X4 = np.ascontiguousarray(X4, np.float64).reshape(4, 3)
    x4 = np.ascontiguousarray(x4, np.float64).reshape(4, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    f = lib().oracle_p3p_solve
    f.restype = C.c_int
    ok = f(_p(X4), _p(x4), _p(K), _p(R), _p(t))
    return (R, t) if ok else None


def reproj_errors(X, x, K, R, t):
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    err = np.zeros(X.shape[0])
    lib().oracle_reproj_errors(_p(X), _p(x), C.c_int(X.shape[0]), _p(K), _p(R), _p(t), _p(err))
    return err


def p3p_hypotheses(X, x, K, samples, thr, want_masks=False):
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 4)
    n, h = X.shape[0], samples.shape[0]
    R = np.zeros((h, 3, 3))
    t = np.zeros((h, 3))
    valid = np.zeros(h, np.uint8)
    counts = np.zeros(h, np.int32)
    masks = np.zeros((h, n), np.uint8) if want_masks else None
    lib().oracle_p3p_hypotheses(_p(X), _p(x), C.c_int(n), _p(K), _p(samples), C.c_int(h), C.c_double(thr),
                                _p(R), _p(t), _p(valid), _p(counts), _p(masks) if want_masks else None)
    return R, t, valid, counts, masks


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros(((H + 1) // 2, (W + 1) // 2), np.uint8)
    lib().oracle_pyr_down(_p(img), C.c_int(H), C.c_int(W), _p(out))
    return out


def klt_num_levels(H, W, win, max_level):
    f = lib().oracle_klt_num_levels
    f.restype = C.c_int
    return f(C.c_int(H), C.c_int(W), C.c_int(win), C.c_int(max_level))


def klt_track(prev, nxt, prev_xy, win=17, max_level=2, max_iter=10, eps=0.03, min_eig=1e-4):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    H, W = prev.shape
    pts = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    lib().oracle_klt_track(_p(prev), _p(nxt), C.c_int(H), C.c_int(W), _p(pts), C.c_int(n), C.c_int(win),
                           C.c_int(max_level), C.c_int(max_iter), C.c_double(eps), C.c_double(min_eig),
                           _p(out), _p(status), _p(err))
    return out, status, err


def match_knn2_ratio(q, t, ratio):
    q = np.ascontiguousarray(np.asarray(q).reshape(len(q), -1), np.float32)
    t = np.ascontiguousarray(np.asarray(t).reshape(len(t), -1), np.float32)
    nq, nt, D = q.shape[0], t.shape[0], q.shape[1]
    pairs = np.zeros((max(nq, 1), 2), np.int32)
    best = np.zeros((max(nq, 1), 2), np.int32)
    d2 = np.zeros((max(nq, 1), 2), np.float64)
    used = np.zeros(max(nt, 1), np.uint8)
    f = lib().oracle_match_knn2_ratio
    f.restype = C.c_int
    n = f(_p(q), C.c_int(nq), _p(t), C.c_int(nt), C.c_int(D), C.c_double(ratio), _p(pairs), _p(best), _p(d2), _p(used))
    return pairs[:n].astype(np.int64), best, d2


def min_eigen_map(img, block=7):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros((H, W), np.float32)
    lib().oracle_min_eigen_map(_p(img), C.c_int(H), C.c_int(W), C.c_int(block), _p(out))
    return out


def good_features(img, mask=None, max_corners=500, quality=0.01, min_distance=8, block=7):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    cap = max_corners if max_corners > 0 else H * W
    xy = np.zeros((cap, 2), np.float32)
    f = lib().oracle_good_features
    f.restype = C.c_int
    n = f(_p(img), C.c_int(H), C.c_int(W), _p(m) if m is not None else None, C.c_int(max_corners),
          C.c_double(quality), C.c_double(min_distance), C.c_int(block), _p(xy))
    return xy[:n].copy()


def sift(img, cap=20000):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    kp = np.zeros((cap, 6), np.float32)
    desc = np.zeros((cap, 128), np.float32)
    f = lib().oracle_sift
    f.restype = C.c_int
    n = f(_p(img), C.c_int(H), C.c_int(W), C.c_int(cap), _p(kp), _p(desc))
    return kp[:n].copy(), desc[:n].copy()
