"""Oracle checks without a GPU: DLT against the reference golden; P3P and KLT
oracles (parity unpinned vs OpenCV) against analytic ground truth."""
import os

import numpy as np
import pytest

from oracle import dlt_np, native

G = os.path.join(os.path.dirname(__file__), "golden")


def test_dlt_oracle_matches_reference_golden():
    g = np.load(os.path.join(G, "dlt_cameras.npz"))
    X = dlt_np.linear_triangulation(g["x1"][:, :, 0], g["x2"][:, :, 0], g["C1"], g["C2"])
    assert np.allclose(X, g["X_clean"][:, :, 0], rtol=1e-9, atol=1e-9)
    assert np.allclose(X, g["X_true"][:, :, 0], atol=1e-4)          # tests/test_triangulation.py:229
    Xn = dlt_np.linear_triangulation(g["x1n"][:, :, 0], g["x2n"][:, :, 0], g["C1"], g["C2"])
    assert np.allclose(Xn, g["X_noisy"][:, :, 0], rtol=1e-9, atol=1e-9)
    m = g["cand_mask"]
    P1, P2 = dlt_np.candidate_projections(g["K"], g["cand_poses"][m], g["cand_current_pose"])
    Xc = dlt_np.linear_triangulation(g["cand_tracks"][m][:, :, 0], g["cand_keypoints"][m][:, :, 0], P1, P2)
    assert np.allclose(Xc, g["X_cand"][:, :, 0], rtol=1e-9, atol=1e-9)


def scene(n=1000, seed=2023):
    rng = np.random.default_rng(seed)
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    th1, th2 = np.pi / 8, np.pi / 32
    R = np.array([[np.cos(th1), -np.sin(th1), 0], [np.sin(th1), np.cos(th1), 0], [0, 0, 1]]) @ np.array(
        [[np.cos(th2), 0, np.sin(th2)], [0, 1, 0], [-np.sin(th2), 0, np.cos(th2)]])
    t = np.array([1.0, 1.0, -1.0])
    X = rng.uniform(-1, 1, size=(n, 3))
    X[:, 2] = X[:, 2] * 5 + 10
    Xc = X @ R.T + t
    x = Xc @ K.T
    return rng, K, R, t, X, x[:, :2] / x[:, 2:]


def test_p3p_oracle_recovers_ground_truth():
    rng, K, R, t, X, x = scene()
    errs = []
    for _ in range(2000):
        idx = rng.choice(len(X), 4, replace=False)
        res = native.p3p_solve(X[idx], x[idx], K)
        assert res is not None, "noise-free sample must yield a pose"
        errs.append(max(np.abs(res[0] - R).max(), np.abs(res[1] - t).max()))
    errs = np.array(errs)
    assert np.median(errs) < 1e-10
    assert np.mean(errs < 1e-6) > 0.99
    assert errs.max() < 1e-3                                        # tests/test_p3p.py:93-98 bar


def test_p3p_oracle_scoring_and_masks():
    rng, K, R, t, X, x = scene()
    xn = x + rng.normal(0, 0.3, size=x.shape)
    xn[:150] += rng.uniform(-40, 40, size=(150, 2))
    samples = np.stack([rng.choice(len(X), 4, replace=False) for _ in range(300)]).astype(np.int32)
    Rh, th, valid, counts, masks = native.p3p_hypotheses(X, xn, K, samples, 1.0, want_masks=True)
    assert valid.sum() > 200
    b = int(np.argmax(counts))
    err = native.reproj_errors(X, xn, K, Rh[b], th[b])
    assert np.array_equal(err < 1.0, masks[b].astype(bool)) and counts[b] == masks[b].sum()
    # error definition: (sqrt(dx^2+dy^2))^2 after x' = X'*(1/Z'), u = x'*fx + cx
    Xc = X @ Rh[b].T + th[b]
    iz = 1.0 / Xc[:, 2]
    u = (Xc[:, 0] * iz) * K[0, 0] + K[0, 2]
    v = (Xc[:, 1] * iz) * K[1, 1] + K[1, 2]
    ref = np.sqrt((xn[:, 0] - u) ** 2 + (xn[:, 1] - v) ** 2) ** 2
    assert np.allclose(err, ref, rtol=1e-12, atol=1e-12)
    assert counts[b] > 600 and np.abs(Rh[b] - R).max() < 5e-3


def shift_image(h, w, seed, dx, dy):
    """Smooth random texture and a sub-pixel shifted copy (analytic flow = (dx, dy))."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img0 = np.zeros((h, w))
    img1 = np.zeros((h, w))
    for _ in range(40):
        fx, fy = rng.uniform(0.02, 0.25, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        amp = rng.uniform(5, 20)
        img0 += amp * np.sin(fx * xx + fy * yy + ph)
        img1 += amp * np.sin(fx * (xx - dx) + fy * (yy - dy) + ph)
    lo, hi = img0.min(), img0.max()
    q = lambda a: np.clip(np.rint((a - lo) / (hi - lo) * 255), 0, 255).astype(np.uint8)
    return q(img0), q(img1)


@pytest.mark.parametrize("dx,dy", [(1.3, -0.7), (5.6, 3.2), (-9.5, 6.25)])
def test_klt_oracle_recovers_analytic_flow(dx, dy):
    prev, nxt = shift_image(240, 320, 5, dx, dy)
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(40, 280, 200), rng.uniform(40, 200, 200)], axis=1).astype(np.float32)
    out, status, err = native.klt_track(prev, nxt, pts, win=17, max_level=2)
    good = status.astype(bool) & (err < 100)
    assert good.mean() > 0.95
    flow = out[good] - pts[good]
    assert np.abs(np.median(flow[:, 0]) - dx) < 0.05 and np.abs(np.median(flow[:, 1]) - dy) < 0.05
    assert np.percentile(np.abs(flow - [dx, dy]).max(axis=1), 90) < 0.15


def test_pyr_down_oracle_properties():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 51)).astype(np.uint8)
    d = native.pyr_down(img)
    assert d.shape == (19, 26)
    assert np.array_equal(native.pyr_down(np.full((20, 30), 77, np.uint8)), np.full((10, 15), 77, np.uint8))
    # interior pixel against the direct 5x5 binomial definition
    w = np.array([1, 4, 6, 4, 1])
    k = np.outer(w, w)
    y, x = 5, 7
    ref = (int((img[2 * y - 2:2 * y + 3, 2 * x - 2:2 * x + 3].astype(int) * k).sum()) + 128) >> 8
    assert d[y, x] == ref
    assert native.klt_num_levels(1241, 1376, 17, 2) == 3 and native.klt_num_levels(40, 40, 17, 3) == 2
