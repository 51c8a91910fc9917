"""NumPy oracle: Harris response, greedy NMS, raw-patch descriptors.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
``HarrisCornerDetector.extractKeypoints`` / ``extractDescriptors``
(reference src/vo/features/harris.py:86-194).  Pinned bit-exactly by
tests/golden/harris_*.npz.
"""
import numpy as np


def _box_sum_valid(a: np.ndarray, p: int) -> np.ndarray:
    """p x p all-ones 'valid' box sum of an int64 array (exact integers).

    Reference: three convolve2d(np.ones((p, p)), X, 'valid') calls at
    harris.py:115-120.  The sums are integers < 2**53, so a float64 direct sum
    in any order equals this integer sum exactly.
    """
    c = np.zeros((a.shape[0] + 1, a.shape[1] + 1), dtype=np.int64)
    np.cumsum(np.cumsum(a, axis=0), axis=1, out=c[1:, 1:])
    return c[p:, p:] - c[:-p, p:] - c[p:, :-p] + c[:-p, :-p]


def harris_scores(img: np.ndarray, patch_size: int = 9, kappa: float = 0.09) -> np.ndarray:
    """(H, W) float64 Harris response in image coordinates.

    harris.py:103-137.  ``signal.convolve2d(sobel, img, 'valid')`` is a TRUE
    convolution (kernel flipped) and yields int64 for a uint8 image:
        Ix[i,j] = sum_a w[a] * (img[i+a, j] - img[i+a, j+2]),  w = (1, 2, 1)
        Iy[i,j] = sum_b w[b] * (img[i, j+b] - img[i+2, j+b])
    """
    assert img.ndim == 2 and img.dtype == np.uint8
    g = img.astype(np.int64)
    H, W = g.shape
    dx = g[:, :-2] - g[:, 2:]                      # (H, W-2)
    ix = dx[:-2] + 2 * dx[1:-1] + dx[2:]           # (H-2, W-2)
    dy = g[:-2, :] - g[2:, :]                      # (H-2, W)
    iy = dy[:, :-2] + 2 * dy[:, 1:-1] + dy[:, 2:]  # (H-2, W-2)

    sxx = _box_sum_valid(ix * ix, patch_size).astype(np.float64)
    syy = _box_sum_valid(iy * iy, patch_size).astype(np.float64)
    sxy = _box_sum_valid(ix * iy, patch_size).astype(np.float64)

    # harris.py:123-127 -- three IEEE roundings, in this order, no FMA
    trace = sxx + syy
    det = sxx * syy - sxy * sxy
    r = det - kappa * (trace * trace)
    r[r < 0] = 0

    pad = patch_size // 2 + 1                      # harris.py:129-137
    out = np.zeros((r.shape[0] + 2 * pad, r.shape[1] + 2 * pad), dtype=np.float64)
    out[pad:pad + r.shape[0], pad:pad + r.shape[1]] = r
    return out


def nms_keypoints(scores: np.ndarray, num_keypoints: int, r: int) -> np.ndarray:
    """Greedy argmax non-maximum suppression, (N, 2, 1) float64 of (x, y).

    harris.py:139-152, including its slicing semantics: a negative slice start
    (y - r < 0 or x - r < 0) selects an empty range, so nothing is suppressed
    and the same pixel is returned for every remaining slot; with all scores 0
    the remaining keypoints are (0, 0).
    """
    s = scores.copy()
    h, w = s.shape
    kp = np.zeros((num_keypoints, 2, 1))
    for i in range(num_keypoints):
        flat = int(np.argmax(s))
        y, x = flat // w, flat % w
        s[y - r:y + r + 1, x - r:x + r + 1] = 0
        kp[i, 0, 0] = x
        kp[i, 1, 0] = y
    return kp


def nms_keypoints_fast(scores: np.ndarray, num_keypoints: int, r: int) -> np.ndarray:
    """Same result as nms_keypoints, via a sorted candidate walk (used for
    full-size checks where 2*N argmax passes would take minutes)."""
    h, w = scores.shape
    flat = scores.ravel()
    cand = np.flatnonzero(flat > 0)
    order = np.lexsort((cand, -flat[cand]))       # score desc, index asc
    cand = cand[order]
    alive = np.ones((h, w), dtype=bool)
    kp = np.zeros((num_keypoints, 2, 1))
    n = 0
    for c in cand:
        if n >= num_keypoints:
            break
        y, x = divmod(int(c), w)
        if not alive[y, x]:
            continue
        kp[n, 0, 0], kp[n, 1, 0] = x, y
        n += 1
        if y - r < 0 or x - r < 0:                # empty slice: no suppression
            kp[n:, 0, 0], kp[n:, 1, 0] = x, y
            n = num_keypoints
            break
        alive[y - r:y + r + 1, x - r:x + r + 1] = False
    return kp


def patch_descriptors(img: np.ndarray, keypoints: np.ndarray, r: int = 9) -> np.ndarray:
    """(N, (2r+1)^2, 1) float64 raw patches from the zero-padded image.

    harris.py:176-192.
    """
    n = keypoints.shape[0]
    d = 2 * r + 1
    padded = np.zeros((img.shape[0] + 2 * r, img.shape[1] + 2 * r), dtype=img.dtype)
    padded[r:r + img.shape[0], r:r + img.shape[1]] = img
    desc = np.zeros((n, d * d, 1))
    for k in range(n):
        x = int(keypoints[k, 0, 0])
        y = int(keypoints[k, 1, 0])
        desc[k, :, 0] = padded[y:y + d, x:x + d].reshape(-1)
    return desc
