"""RANSAC oracle and the library's host-side sampler/replay against the goldens
captured from the reference (no GPU needed: these entry points are host logic)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ransac_np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ransac.npz"))


def test_iteration_bound_table():
    for conf, orat, s, k in G["n_iter_table"]:
        assert ransac_np.num_iterations(conf, orat, int(s)) == int(k)


def test_parabola_trace_matches_reference():
    data, max_noise = G["parabola_data"], G["parabola_max_noise"]
    calls = {"n": 0}

    def model_fn(s):
        calls["n"] += 1
        return np.polyfit(s[:, 0], s[:, 1], 2)

    r = ransac_np.Ransac(3, data, model_fn, lambda p, pts: np.abs(np.polyval(p, pts[:, 0]) - pts[:, 1]),
                         float(max_noise[0]) + 1e-5, 1 / 3, 0.99)
    assert r.n_iterations == int(G["parabola_n_iter0"])
    model, inl = r.find_best_model()
    assert np.array_equal(model, G["parabola_model"]) and np.array_equal(inl, G["parabola_inliers"])
    assert r.n_iterations == int(G["parabola_n_iter_final"]) and r.outlier_ratio == float(G["parabola_outlier_ratio_final"])
    assert calls["n"] == int(G["parabola_model_calls"])
    assert np.array_equal(r.rng.integers(0, 2**62, size=4), G["parabola_rng_next"])
    model2, inl2 = r.find_best_model()                              # persistent state across calls
    assert np.array_equal(model2, G["parabola_model2"]) and np.array_equal(inl2, G["parabola_inliers2"])
    assert r.n_iterations == int(G["parabola_n_iter_final2"])
    x = np.linspace(data[:, 0].min(), data[:, 0].max(), 100)        # tests/test_ransac.py:66-72
    assert np.allclose(np.polyval(G["parabola_poly"], x), np.polyval(model, x), atol=2e-3)


@pytest.mark.parametrize("key", [k for k in G.files if k.startswith("choice")])
def test_library_sampler_reproduces_numpy_choice(key):
    from vo import _native
    s, pop = int(key[6:key.index("_")]), int(key.split("pop")[1])
    pcg = _native.Pcg64.from_generator(np.random.default_rng(2023))
    got = _native.rng_choice(pcg, pop, s, G[key].shape[0])
    assert np.array_equal(got, G[key])
    # the advanced state written back into a Generator continues the same stream
    gen = np.random.default_rng(0)
    pcg.to_generator(gen)
    ref = np.random.default_rng(2023)
    for _ in range(G[key].shape[0]):
        ref.choice(np.arange(pop), replace=False, size=s)
    assert np.array_equal(gen.choice(np.arange(pop), replace=False, size=s), ref.choice(np.arange(pop), replace=False, size=s))


def test_library_iteration_bound_and_replay():
    from vo import _native
    for conf, orat, s, k in G["n_iter_table"]:
        assert _native.ransac_num_iterations(conf, orat, int(s)) == int(k)
    # replay: synthetic (valid, count) stream against the oracle loop driven by the same stream
    rng = np.random.default_rng(3)
    N, B = 500, 400
    valid = (rng.random(B) > 0.1).astype(np.uint8)
    counts = rng.integers(0, 400, size=B).astype(np.int32)
    it = iter(range(B))
    cur = {}

    def model_fn(_):
        cur["b"] = next(it)
        return cur["b"] if valid[cur["b"]] else None

    def error_fn(m, pop):
        e = np.ones(N)
        e[: counts[m]] = 0.0
        return e

    r = ransac_np.Ransac(4, np.arange(N), model_fn, error_fn, 0.5, 0.9, 0.99, 1000, p3p=True)
    model, _ = r.find_best_model()
    st = _native.RansacState(0.9, 0.99, 1000, min(1000, ransac_np.num_iterations(0.99, 0.9, 4)), 4, 1)
    n_done, best_c, best_i = C.c_int64(0), C.c_int32(-1), C.c_int32(-1)
    consumed, finished = C.c_int(0), C.c_int(0)
    rc = _native.load().vo_ransac_replay(C.byref(st), valid.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                         B, N, C.byref(n_done), C.byref(best_c), C.byref(best_i), 0, C.byref(consumed),
                                         C.byref(finished))
    assert rc == 0 and finished.value == 1
    assert best_i.value == model and consumed.value == cur["b"] + 1
    assert n_done.value == r.iterations_done and st.n_iterations == r.n_iterations and st.outlier_ratio == r.outlier_ratio
