"""-m gpu: HIP DLT / P3P / reprojection / pyramid / KLT kernels through the C ABI
against the oracle (and, for DLT, the reference golden)."""
import os

import numpy as np
import pytest

from oracle import dlt_np, native
from test_oracle_geometry import scene, shift_image

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from vo import _native
    c = _native.Context(0)
    yield c
    c.close()


# ---------------- DLT (parity by tolerance: SVD route differs from LAPACK) ----------------
def test_dlt_matches_reference_golden(ctx):
    g = np.load(os.path.join(G, "dlt_cameras.npz"))
    X = ctx.triangulate_dlt(g["x1"], g["x2"], g["C1"], g["C2"])
    assert np.allclose(X, g["X_clean"][:, :, 0], rtol=1e-9, atol=1e-9)
    assert np.allclose(X, g["X_true"][:, :, 0], atol=1e-4)                  # reference bar, test_triangulation.py:229
    Xn = ctx.triangulate_dlt(g["x1n"], g["x2n"], g["C1"], g["C2"])
    assert np.allclose(Xn, g["X_noisy"][:, :, 0], rtol=1e-8, atol=1e-8)
    m = g["cand_mask"]
    P1, P2 = dlt_np.candidate_projections(g["K"], g["cand_poses"][m], g["cand_current_pose"])
    Xc = ctx.triangulate_dlt(g["cand_tracks"][m], g["cand_keypoints"][m], P1, P2)
    assert np.allclose(Xc, g["X_cand"][:, :, 0], rtol=1e-8, atol=1e-8)


def test_dlt_candidates_with_per_track_poses_match_reference_golden(ctx):
    """triangulate_candidates at the per-frame loop's shape: 2000 tracks, each with its own start pose
    (triangulation.py:38-86; golden from the reference's own code)."""
    g = np.load(os.path.join(G, "dlt_candidates.npz"))
    m = g["mask"]
    P1, P2 = dlt_np.candidate_projections(g["K"], g["start_poses"][g["start_index"]][m], g["current_pose"])
    X = ctx.triangulate_dlt(g["tracks"][m], g["keypoints"][m], P1, P2)
    assert X.shape == (2000, 3)
    assert np.allclose(X, g["X_cand"][:, :, 0], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("n", [1, 7, 2000, 8000])
def test_dlt_matches_oracle_seeded(ctx, n):
    rng, K, R, t, X, x2 = scene(n, seed=11 + n)
    x1 = X @ K.T
    x1 = x1[:, :2] / x1[:, 2:]
    x1 += rng.normal(0, 0.4, x1.shape)
    C1 = K @ np.eye(4)[:3]
    C2 = K @ np.hstack([R, t[:, None]])
    ref = dlt_np.linear_triangulation(x1, x2, C1, C2)
    got = ctx.triangulate_dlt(x1, x2, C1, C2)
    assert np.allclose(got, ref, rtol=1e-8, atol=1e-8)


def test_dlt_empty(ctx):
    assert ctx.triangulate_dlt(np.zeros((0, 2)), np.zeros((0, 2)), np.eye(4)[:3], np.eye(4)[:3]).shape == (0, 3)


# ---------------- P3P hypotheses + scoring ----------------
@pytest.mark.parametrize("n,hyp,noise", [(1000, 1000, 0.0), (2000, 1000, 0.4), (37, 64, 0.2), (8000, 4000, 0.5)])
def test_p3p_hypotheses_match_oracle(ctx, n, hyp, noise):
    rng, K, R, t, X, x = scene(n, seed=5 + n)
    xn = x + rng.normal(0, noise, x.shape) if noise else x.copy()
    k = n // 6
    xn[:k] += rng.uniform(-40, 40, size=(k, 2))                              # gross outliers
    samples = np.stack([rng.choice(n, 4, replace=False) for _ in range(hyp)]).astype(np.int32)
    thr = 1.0
    Rr, tr, vr, cr, mr = native.p3p_hypotheses(X, xn, K, samples, thr, want_masks=True)
    Rg, tg, vg, cg, mg = ctx.p3p_hypotheses(X, xn, K, samples, thr, want_masks=True)
    assert np.array_equal(vg, vr), "valid flags (reference: model is None) differ"
    assert np.allclose(Rg, Rr, rtol=0, atol=1e-9) and np.allclose(tg, tr, rtol=0, atol=1e-9)
    # inlier masks and counts: bit-exact
    assert np.array_equal(cg, cr)
    assert np.array_equal(mg, mr.astype(bool))
    exact = np.array_equal(Rg, Rr) and np.array_equal(tg, tr)
    print("p3p poses bit-identical to oracle:", exact, "valid:", int(vg.sum()), "/", hyp)
    b = int(np.argmax(cg))
    assert np.abs(Rg[b] - R).max() < 1e-2


def test_reproj_inliers_match_oracle(ctx):
    rng, K, R, t, X, x = scene(3000, seed=9)
    xn = x + rng.normal(0, 0.7, x.shape)
    for thr in (0.25, 1.0, 1.5625):
        mask, err = ctx.reproj_inliers(X, xn, K, R, t, thr, want_err=True)
        ref = native.reproj_errors(X, xn, K, R, t)
        assert np.array_equal(err, ref), "squared reprojection errors not bit-identical"
        assert np.array_equal(mask, ref < thr)


# ---------------- pyramid + KLT ----------------
@pytest.mark.parametrize("shape", [(37, 51), (240, 320), (1241, 1376)])
def test_pyr_down_bit_exact(ctx, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=shape).astype(np.uint8)
    assert np.array_equal(ctx.pyr_down(img), native.pyr_down(img))


@pytest.mark.parametrize("shape,levels", [((240, 320), 3), ((200, 264), 3), ((203, 262), 3), ((1241, 1376), 3),
                                          ((621, 700), 4), ((130, 132), 2)])
def test_bordered_pyramid_bit_exact(ctx, shape, levels):
    """vo_pyramid_build_dev: every level with its 32-pixel reflect-101 border, as the tracker reads it.  Widths that
    are multiples of 4 take the word-wide tiled kernel (levels 0..2 in one launch), the others the byte kernels;
    both must equal pyrDown level by level, and the border must mirror the interior (np.pad 'reflect')."""
    run_pyramid_case(ctx, shape, levels)


def test_bordered_pyramid_large_tiles_bit_exact():
    """The 64x32-tile variant the batched launches take (VO_PYR_TILE is read once per process: own process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = %r; import test_gpu_geometry as t; from vo import _native; c = _native.Context(0); "
            "[t.run_pyramid_case(c, s, 3) for s in ((240, 320), (200, 264), (1241, 1376), (621, 700))]; c.close(); print('ok')"
            % [p for p in sys.path if p])
    env = dict(os.environ, VO_PYR_TILE="64")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def run_pyramid_case(ctx, shape, levels):
    PAD = 32
    H, W = shape
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, size=shape).astype(np.uint8)
    nbytes = ctx.pyramid_bytes(H, W, levels)
    d_img = ctx.to_device(img)
    d_pyr = ctx.to_device(np.full(nbytes, 0xAB, np.uint8))
    ctx.pyramid_build_dev(d_img, H, W, levels, d_pyr)
    ctx.sync()
    got = ctx.download(d_pyr, (nbytes,), np.uint8)
    ctx.free(d_img)
    ctx.free(d_pyr)
    off, lvl, h, w = 0, img, H, W
    for l in range(levels):
        pitch = (w + 2 * PAD + 3) & ~3
        rows = h + 2 * PAD
        g = got[off:off + rows * pitch].reshape(rows, pitch)[:, :w + 2 * PAD]
        ref = np.pad(lvl, PAD, mode="reflect")
        assert np.array_equal(g[PAD:PAD + h, PAD:PAD + w], lvl), "level %d interior" % l
        assert np.array_equal(g, ref), "level %d border" % l
        off += (rows * pitch + 255) & ~255
        lvl = native.pyr_down(lvl)
        h, w = (h + 1) // 2, (w + 1) // 2


@pytest.mark.parametrize("dx,dy,win,lvl", [(1.3, -0.7, 17, 2), (5.6, 3.2, 17, 2), (-9.5, 6.25, 15, 2),
                                           (2.0, 1.0, 21, 3), (0.4, 0.2, 9, 0), (1.3, -0.7, 15, 2),
                                           (1.3, -0.7, 21, 2), (-0.37, 0.81, 17, 3)])
def test_klt_matches_oracle_and_flow(ctx, dx, dy, win, lvl):
    prev, nxt = shift_image(240, 320, 5, dx, dy)
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(-5, 325, 400), rng.uniform(-5, 245, 400)], axis=1).astype(np.float32)
    ro, rs, re = native.klt_track(prev, nxt, pts, win=win, max_level=lvl)
    go, gs, ge = ctx.klt_track(prev, nxt, pts, win=win, max_level=lvl)
    assert np.array_equal(gs, rs), "status flags differ"
    assert np.array_equal(go, ro), "tracked points not bit-identical to the oracle"
    assert np.array_equal(ge, re), "error measures not bit-identical"
    good = gs.astype(bool) & (ge < 100) & (pts[:, 0] > 40) & (pts[:, 0] < 280) & (pts[:, 1] > 40) & (pts[:, 1] < 200)
    flow = go[good] - pts[good]
    assert np.abs(np.median(flow[:, 0]) - dx) < 0.1 and np.abs(np.median(flow[:, 1]) - dy) < 0.1


@pytest.mark.parametrize("dx,dy,win,lvl", [(7.3, -5.6, 15, 2), (12.5, 9.25, 15, 3), (-9.4, 6.1, 17, 2), (5.5, 5.5, 21, 1)])
def test_klt_large_motion_restages(ctx, dx, dy, win, lvl):
    """Motions beyond the slack staged around the search window: the window walks out of the region the
    row kernels requested ahead of time and they stage again; still bit-identical to the oracle."""
    prev, nxt = shift_image(240, 320, 6, dx, dy)
    rng = np.random.default_rng(7)
    pts = np.stack([rng.uniform(-20, 340, 500), rng.uniform(-20, 260, 500)], axis=1).astype(np.float32)
    ro, rs, re = native.klt_track(prev, nxt, pts, win=win, max_level=lvl)
    go, gs, ge = ctx.klt_track(prev, nxt, pts, win=win, max_level=lvl)
    assert np.array_equal(gs, rs) and np.array_equal(go, ro) and np.array_equal(ge, re)


def test_klt_full_size(ctx):
    dx, dy = 3.25, -1.5
    prev, nxt = shift_image(1241, 1376, 8, dx, dy)
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(30, 1340, 2000), rng.uniform(30, 1200, 2000)], axis=1).astype(np.float32)
    go, gs, ge = ctx.klt_track(prev, nxt, pts, win=15, max_level=2)
    ro, rs, re = native.klt_track(prev, nxt, pts, win=15, max_level=2)
    assert np.array_equal(gs, rs) and np.array_equal(go, ro) and np.array_equal(ge, re)
    good = gs.astype(bool) & (ge < 100)
    assert good.mean() > 0.95
    assert np.percentile(np.abs(go[good] - pts[good] - [dx, dy]).max(axis=1), 90) < 0.2


def test_klt_edge_cases(ctx):
    prev, nxt = shift_image(64, 80, 3, 0.5, 0.5)
    go, gs, ge = ctx.klt_track(prev, nxt, np.zeros((0, 2), np.float32))
    assert go.shape == (0, 2)
    flat = np.full((64, 80), 128, np.uint8)                                   # textureless: min-eig test fails
    go, gs, ge = ctx.klt_track(flat, flat, np.array([[30.0, 30.0]], np.float32))
    ro, rs, re = native.klt_track(flat, flat, np.array([[30.0, 30.0]], np.float32))
    assert gs[0] == 0 and rs[0] == 0
    far = np.array([[-100.0, 20.0], [500.0, 20.0]], np.float32)               # window outside the image
    go, gs, ge = ctx.klt_track(prev, nxt, far)
    ro, rs, re = native.klt_track(prev, nxt, far)
    assert np.array_equal(gs, rs) and not gs.any() and np.array_equal(go, ro)


@pytest.mark.parametrize("n,noise,masked", [(1000, 0.5, False), (2000, 0.5, True), (40, 1.0, False), (5, 0.3, False)])
def test_refine_pose_matches_oracle(ctx, n, noise, masked):
    """vo_refine_pose (p3p.py:188-213) against the Gauss-Newton oracle: same algorithm, different
    summation order and libm -> 1e-8, not bits; and it lands near the pose the data came from."""
    from oracle import refine_np
    from test_oracle_refine import K as KR, scene as rscene
    rng = np.random.default_rng(n)
    X, x, R, t, R0, t0 = rscene(rng, n, noise)
    mask = None
    if masked:
        mask = rng.uniform(size=n) < 0.6
        x = x.copy()
        x[~mask] += rng.normal(0, 40, ((~mask).sum(), 2))     # gross outliers the mask must keep out
    sel = slice(None) if mask is None else mask
    Ro, to, ito, costo = refine_np.refine_pose(X[sel], x[sel], KR, R0, t0)
    Rg, tg, itg, costg = ctx.refine_pose(X, x, KR, R0, t0, inlier_mask=mask)
    if n >= 40:
        assert abs(itg - ito) <= 1        # the last step sits at the rounding floor: it may or may not count
    tol = 1e-8 if n >= 40 else 1e-6       # five points leave the pose barely determined
    assert np.abs(Rg - Ro).max() < tol and np.abs(tg - to).max() < tol
    assert abs(costg - costo) <= tol * max(costo, 1.0)
    if n >= 40:
        assert np.abs(Rg - R).max() < 5e-3 and np.abs(tg - t).max() < 0.1


def test_refine_pose_degenerate(ctx):
    from test_oracle_refine import K as KR
    R0, t0 = np.eye(3), np.array([0.1, 0.2, 0.3])
    Rg, tg, it, cost = ctx.refine_pose(np.zeros((0, 3)), np.zeros((0, 2)), KR, R0, t0)
    assert it == 0 and np.array_equal(Rg, R0) and np.array_equal(tg, t0)
