"""-m gpu: the device-resident frame pipeline (vo_pipeline_*) against the same
stages composed from the CPU oracles, frame by frame."""
import numpy as np
import pytest

from oracle import dlt_np, harris_np, native, ransac_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from vo import _native
    c = _native.Context(0)
    yield c
    c.close()


def oracle_step(stream, prev, nxt, kp_prev, rs, cfg):
    """One frame of the front-end with CPU oracles; rs = persistent oracle Ransac object."""
    K = stream.K
    out, status, err = native.klt_track(stream.image(prev), stream.image(nxt), kp_prev.astype(np.float32),
                                        win=cfg["win"], max_level=cfg["lvl"])
    keep = status.astype(bool) & (err < 100.0)
    p_c, n_c = kp_prev[keep], out[keep].astype(np.float64)
    z = stream.depth(prev)[p_c[:, 1].astype(int), p_c[:, 0].astype(int)].astype(np.float64)
    xc = (p_c[:, 0] - K[0, 2]) / K[0, 0] * z
    yc = (p_c[:, 1] - K[1, 2]) / K[1, 1] * z
    T = stream.T_world_cam(prev)
    land = np.stack([T[r, 0] * xc + T[r, 1] * yc + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)
    sc = harris_np.harris_scores(stream.image(nxt), 9, 0.09)
    kp_next = harris_np.nms_keypoints_fast(sc, cfg["N"], 5)[:, :, 0]
    # re-bind the oracle RANSAC to this frame's correspondences (the estimator object persists)
    rs.model_fn = lambda idx: native.p3p_solve(land[np.asarray(idx).reshape(-1)], n_c[np.asarray(idx).reshape(-1)], K)
    rs.error_fn = lambda m, pop: native.reproj_errors(land, n_c, K, m[0], m[1])
    n0 = len(rs.trace)
    (R, t), inl = rs.find_best_model(np.arange(len(land)))
    Tcw = np.linalg.inv(T)
    tri = dlt_np.linear_triangulation(p_c, n_c, K @ Tcw[:3], K @ np.hstack([R, t[:, None]]))
    return dict(kp_next=kp_next, prev_xy=p_c, next_xy=n_c, landmarks=land, R=R, t=t, inliers=inl, tri=tri,
                draws=len(rs.trace) - n0, iters=rs.iterations_done)


@pytest.mark.parametrize("sampler", ["device", "sequential"])
@pytest.mark.parametrize("H,W,N,hyp", [(240, 320, 300, 256), (480, 640, 500, 1000)])
def test_pipeline_matches_oracle_composition(ctx, H, W, N, hyp, sampler, monkeypatch):
    """sampler = "device": sample indices derived in the solve kernel from raw generator outputs;
    "sequential": the host sampler the pipeline falls back to when a draw may have been rejected
    (VO_SEQ_SAMPLER forces it).  Both must reproduce the reference's sample stream."""
    from vo import _native, synthetic
    if sampler == "sequential":
        monkeypatch.setenv("VO_SEQ_SAMPLER", "1")
    else:
        monkeypatch.delenv("VO_SEQ_SAMPLER", raising=False)
    F = 4
    stream = synthetic.Stream(F, H, W)
    cfg = dict(win=15, lvl=2, N=N)
    pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                            p3p_threshold=1.0, max_iterations=1000)
    for i in range(F):
        pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
    order = stream.order(6)
    pipe.prime(order[0])
    kp = harris_np.nms_keypoints_fast(harris_np.harris_scores(stream.image(order[0]), 9, 0.09), N, 5)[:, :, 0]
    rs = ransac_np.Ransac(4, np.arange(4), None, None, 1.0, 0.9, 0.99, 1000, adaptive=True, p3p=True)
    for a, b in zip(order[:-1], order[1:]):
        ref = oracle_step(stream, a, b, kp, rs, cfg)
        r = pipe.step(a, b)
        got = pipe.fetch(r.n_tracked)
        assert r.n_tracked == len(ref["prev_xy"])
        assert np.array_equal(got["kp_next"], ref["kp_next"])
        assert np.array_equal(got["prev_xy"], ref["prev_xy"]) and np.array_equal(got["next_xy"], ref["next_xy"])
        assert np.allclose(got["landmarks"], ref["landmarks"], rtol=1e-12, atol=1e-12)
        assert r.draws_consumed == ref["draws"] and r.ransac_iterations == ref["iters"]
        R, t = np.array(r.R).reshape(3, 3), np.array(r.t)
        assert np.allclose(R, ref["R"], atol=1e-9) and np.allclose(t, ref["t"], atol=1e-9)
        assert np.array_equal(got["inliers"], ref["inliers"]) and r.n_inliers == ref["inliers"].sum()
        assert np.allclose(got["triangulated"], ref["tri"], rtol=1e-6, atol=1e-6)
        # against analytic ground truth of the stream
        Tcw = np.linalg.inv(stream.T_world_cam(b))
        assert np.abs(R - Tcw[:3, :3]).max() < 5e-3 and np.abs(t - Tcw[:3, 3]).max() < 0.1
        kp = ref["kp_next"]
    pipe.close()


@pytest.mark.parametrize("sampler", ["device", "sequential"])
def test_pipeline_lookahead_matches_step(ctx, sampler, monkeypatch):
    """submit(k+1) before collect(k) (one frame of look-ahead) must give what step() gives frame by
    frame: same counts, same RANSAC bookkeeping, same poses -- including the generator hand-over
    between a collected step and the one already in flight."""
    from vo import _native, synthetic
    if sampler == "sequential":
        monkeypatch.setenv("VO_SEQ_SAMPLER", "1")
    else:
        monkeypatch.delenv("VO_SEQ_SAMPLER", raising=False)
    H, W, N, hyp, F = 240, 320, 300, 256, 5
    stream = synthetic.Stream(F, H, W)
    order = stream.order(12)

    def make():
        pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                                p3p_threshold=1.0, max_iterations=1000)
        for i in range(F):
            pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
        pipe.prime(order[0])
        return pipe

    def fields(r):
        return (r.n_tracked, r.n_inliers, r.best_index, r.hyp_valid, r.ransac_iterations, r.draws_consumed,
                tuple(r.R), tuple(r.t))

    pipe = make()
    ref = [fields(pipe.step(a, b)) for a, b in zip(order[:-1], order[1:])]
    pipe.close()
    pipe = make()
    got = []
    pairs = list(zip(order[:-1], order[1:]))
    pipe.submit(*pairs[0])
    for k in range(len(pairs)):
        if k + 1 < len(pairs):
            pipe.submit(*pairs[k + 1])
        got.append(fields(pipe.collect()))
    last = pipe.fetch(got[-1][0])            # nothing in flight any more: fetch works again
    assert last["prev_xy"].shape == (got[-1][0], 2)
    pipe.close()
    assert got == ref


def test_pipeline_stress_configuration_properties(ctx):
    """BASELINE.json configs[4] (3840x2160, 8000 keypoints, 4-level pyramid, 4000 hypotheses): far
    beyond what the oracle finishes in seconds, so checked through properties that do not depend
    on size -- the greedy NMS rule (keypoints in decreasing score order, no two within r, none
    better left unsuppressed), tracked points near the frame, and the recovered pose against the
    stream's analytic ground truth."""
    from vo import _native, synthetic
    H, W, N, hyp, F, r = 2160, 3840, 8000, 4000, 3, 5
    stream = synthetic.Stream(F, H, W)
    pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=3, hyp=hyp,
                            p3p_threshold=1.0, max_iterations=1000)
    for i in range(F):
        pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
    pipe.prime(0)
    for a, b in ((0, 1), (1, 2)):
        res = pipe.step(a, b)
        got = pipe.fetch(res.n_tracked)
        # pose
        Tcw = np.linalg.inv(stream.T_world_cam(b))
        R, t = np.array(res.R).reshape(3, 3), np.array(res.t)
        assert res.n_tracked > 0.8 * N and res.n_inliers > 0.3 * res.n_tracked
        assert np.abs(R - Tcw[:3, :3]).max() < 5e-3 and np.abs(t - Tcw[:3, 3]).max() < 0.1
        # (OpenCV's rule keeps a track while its window's corner is within one window of the frame)
        assert (got["next_xy"] > -15).all() and (got["next_xy"][:, 0] < W + 15).all() and (got["next_xy"][:, 1] < H + 15).all()
        # NMS of frame b
        kp = got["kp_next"].astype(np.int64)
        sc = ctx.harris_response(stream.image(b), 9, 0.09)
        s = sc[kp[:, 1], kp[:, 0]]
        assert (s > 0).all() and (np.diff(s) <= 0).all(), "keypoints are not in decreasing score order"
        occupied = np.zeros((H, W), bool)
        for (x, y) in kp:                                   # no pick inside an earlier pick's window
            assert not occupied[y, x]
            occupied[max(y - r, 0):y + r + 1, max(x - r, 0):x + r + 1] = True
        # nothing better than the last pick is left outside every window (the greedy rule took the
        # best remaining pixel each time)
        left = np.where(occupied, 0.0, sc)
        assert left.max() <= s[-1]
    pipe.close()


def test_pipeline_records_for_the_shared_map(ctx):
    """vo_pipeline_export_state_post / _join with a step in flight: the record of every collected step
    [T_cw | n | landmarks] equals the refined pose of its result and what fetch() returns for the same step
    run blocking (the per-GPU record that is all-gathered)."""
    from vo import _native, sharding, synthetic
    H, W, N, hyp, F = 240, 320, 300, 256, 5
    stream = synthetic.Stream(F, H, W)
    order = stream.order(8)
    pairs = list(zip(order[:-1], order[1:]))

    def make():
        pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                                p3p_threshold=1.0, max_iterations=1000, refine_iters=10)
        for i in range(F):
            pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
        pipe.prime(order[0])
        return pipe

    pipe = make()
    ref = []
    for a, b in pairs:
        r = pipe.step(a, b)
        ref.append((np.array(r.R_refined).reshape(3, 3), np.array(r.t_refined), pipe.fetch(r.n_tracked)["triangulated"]))
    pipe.close()

    cap = N
    L = sharding.record_length(cap)
    recs = ctx.to_device(np.zeros(len(pairs) * L))
    pipe = make()
    pipe.submit(*pairs[0])
    for k in range(len(pairs)):
        if k + 1 < len(pairs):
            pipe.submit(*pairs[k + 1])
        r = pipe.collect()
        pipe.export_state_post(r, cap, recs + k * L * 8)
    pipe.export_state_join()                 # consumer = the context's stream, which the download uses
    host = ctx.download(recs, (len(pairs) * L,), np.float64)
    pipe.close()
    ctx.free(recs)
    got = sharding.unpack_records(host, len(pairs), cap)
    for (T, lm), (R, t, tri) in zip(got, ref):
        np.testing.assert_array_equal(T[:3, :3], R)
        np.testing.assert_array_equal(T[:3, 3], t)
        np.testing.assert_array_equal(T[3], [0, 0, 0, 1])
        np.testing.assert_array_equal(lm, tri[:cap])
