"""Host-side parts of the drop-in `vo` package (no GPU): containers and bookkeeping pinned
bit-for-bit by goldens captured from the reference, plus the reference's own unit tests
for helpers/features restated against this package."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _HostTriangulator:
    """triangulate_candidates on the host oracle: the bookkeeping scenario needs the
    landmarks it produces, and this test must run without a GPU."""

    def __init__(self, camera1, camera2, **kw):
        self.camera1, self.camera2 = camera1, camera2

    def triangulate_candidates(self, features, current_pose):
        from oracle import dlt_np
        m = features.candidate_mask
        P1, P2 = dlt_np.candidate_projections(self.camera1.intrinsic_matrix, features.poses[m], current_pose)
        return dlt_np.linear_triangulation(features.tracks[m][:, :, 0], features.keypoints[m][:, :, 0], P1, P2).reshape(-1, 3, 1)


def test_bookkeeping_matches_reference_golden():
    import vo.primitives as P
    from vo.sensors import Camera
    from scenarios import bookkeeping_scenario
    ns = types.SimpleNamespace(Features=P.Features, Frame=P.Frame, Matches=P.Matches, State=P.State, Camera=Camera,
                               LandmarksTriangulator=_HostTriangulator)
    got = bookkeeping_scenario(ns)
    ref = np.load(os.path.join(G, "bookkeeping.npz"))
    assert set(got) == set(ref.files)
    for k in ref.files:
        if k.endswith("cand_landmarks") or k.endswith("post_landmarks"):
            assert np.allclose(got[k], ref[k], rtol=1e-9, atol=1e-9, equal_nan=True), k   # through a different SVD call path
        else:
            assert np.array_equal(got[k], ref[k], equal_nan=True), k


def test_helpers_match_reference_golden():
    from vo.helpers import H_matrix_to_twist, twist_to_H_matrix
    g = np.load(os.path.join(G, "helpers.npz"))
    for tw, H, back in zip(g["twists"], g["H"], g["twists_back"]):
        assert np.array_equal(twist_to_H_matrix(tw), H)
        assert np.array_equal(np.real(H_matrix_to_twist(H)), back)


# ---- the reference's tests/test_helpers.py, restated ----
def test_homogeneous_cartesian_skew():
    from vo.helpers import to_cartesian_coordinates, to_homogeneous_coordinates, to_skew_symmetric_matrix
    pts = np.array([[[1.0], [2.0]], [[3.0], [4.0]]])
    hom = to_homogeneous_coordinates(pts)
    assert np.array_equal(hom, np.array([[[1.0], [2.0], [1.0]], [[3.0], [4.0], [1.0]]]))
    assert np.array_equal(to_cartesian_coordinates(hom * 2), pts)
    with pytest.warns(RuntimeWarning):
        to_cartesian_coordinates(np.array([[[1.0], [0.0]]]))
    v = np.array([[1.0], [2.0], [3.0]])
    S = to_skew_symmetric_matrix(v)
    assert np.array_equal(S, np.array([[0, -3, 2], [3, 0, -1], [-2, 1, 0]], dtype=float))
    w = np.array([[0.5], [-1.0], [4.0]])
    assert np.allclose(S @ w, np.cross(v[:, 0], w[:, 0]).reshape(3, 1))
    assert to_skew_symmetric_matrix(np.stack([v, w])).shape == (2, 3, 3)


@pytest.mark.parametrize("D", [2, 3])
def test_normalize_points_invariants(D):
    from vo.helpers import normalize_points, to_cartesian_coordinates, to_homogeneous_coordinates
    rng = np.random.default_rng(2023)
    pts = rng.uniform(-50, 400, size=(200, D, 1))
    n, T = normalize_points(pts)
    assert np.allclose(n.mean(axis=0), 0, atol=1e-12)
    assert np.isclose(np.sqrt(np.mean(np.sum(n ** 2, axis=-2))), np.sqrt(D))
    assert np.allclose(to_cartesian_coordinates(T @ to_homogeneous_coordinates(pts)), n)


# ---- the reference's tests/test_features.py, restated ----
def test_features_defaults():
    from vo.primitives import Features
    f = Features(np.zeros((5, 2, 1)))
    assert f.length == 5 and f.descriptors is None
    assert np.all(np.isnan(f.landmarks)) and f.landmarks.shape == (5, 3, 1)
    assert np.array_equal(f.state, np.zeros(5)) and not f.candidate_mask.any()
    assert np.array_equal(f.poses, np.stack([np.eye(4)] * 5)) and np.array_equal(f.tracks, f.keypoints)
    with pytest.raises(AssertionError):
        Features(np.zeros((5, 2)))
    with pytest.raises(AssertionError):
        f.descriptors = np.zeros((4, 8))
    assert Features(np.zeros((0, 2, 1))).poses.shape == (0, 4, 4)
    f.state = np.array([0, 1, 2, 1, 0])
    f.mask(np.array([True, False, True, True, False]))
    assert f.length == 3 and np.array_equal(f.state, [0, 2, 1])


def test_tracker_rejects_unknown_mode():
    from vo.features import Tracker
    with pytest.raises(Exception, match="Tracker Name not valid"):
        Tracker(None, mode="orb")


def test_generic_ransac_matches_reference_trace():
    """vo.algorithms.RANSAC (callable path) against the reference's parabola trace."""
    from vo.algorithms import RANSAC
    g = np.load(os.path.join(G, "ransac.npz"))
    data = g["parabola_data"]
    r = RANSAC(3, data, lambda s: np.polyfit(s[:, 0], s[:, 1], 2), lambda p, pts: np.abs(np.polyval(p, pts[:, 0]) - pts[:, 1]),
               float(g["parabola_max_noise"][0]) + 1e-5, 1 / 3, 0.99)
    model, inl = r.find_best_model()
    assert np.array_equal(model, g["parabola_model"]) and np.array_equal(inl, g["parabola_inliers"])
    assert r.n_iterations == int(g["parabola_n_iter_final"])
    assert np.array_equal(r.rng.integers(0, 2**62, size=4), g["parabola_rng_next"])   # as the capture script did
    model2, inl2 = r.find_best_model()
    assert np.array_equal(model2, g["parabola_model2"]) and np.array_equal(inl2, g["parabola_inliers2"])


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports everything include/vo_hip.h declares."""
    from vo import _native
    lib = ctypes.CDLL(_native.lib_path())
    header = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", header))
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.vo_version() >= 100
    bound = set(_native._SIGS)
    assert bound <= names, sorted(bound - names)


def test_inlier_limit_on_the_sum_of_squares_is_the_reference_decision():
    """p3p.py:81-108 / ransac.py:104-106: inlier <=> norm(dx, dy)**2 < thr.  The hypothesis kernel compares the sum of
    squares with vo_inlier_sum_sq_limit(thr): the limit is the last double that is an inlier, the next one is not, and
    sums drawn around it decide as the reference does."""
    from vo import _native
    lib = ctypes.CDLL(_native.lib_path())
    lib.vo_inlier_sum_sq_limit.restype = ctypes.c_double
    lib.vo_inlier_sum_sq_limit.argtypes = [ctypes.c_double]
    err = lambda s: np.sqrt(np.float64(s)) ** 2
    rng = np.random.default_rng(11)
    thrs = [1.0, 1.25 ** 2, 1.5625, 2.0, 1e-12, 1e-300, 5e-324, 1e300, 0.1 ** 2, 3.0, 4.0, 9.0, 2.0 ** -1074, 2.0 ** 1023]
    thrs += list(np.exp(rng.uniform(-40, 40, size=300))) + list(rng.uniform(0.5, 4.0, size=300))
    for thr in thrs:
        lim = lib.vo_inlier_sum_sq_limit(float(thr))
        assert lim >= 0.0 and err(lim) < thr, (thr, lim)
        up = np.nextafter(lim, np.inf)
        assert not (err(up) < thr), (thr, lim)
        around = np.float64(lim) * (1.0 + rng.integers(-40, 41, size=64) * 2.0 ** -52)
        assert np.array_equal(err(around) < thr, around <= lim), thr
    for thr in (0.0, -1.0, float("nan")):
        assert lib.vo_inlier_sum_sq_limit(thr) == -1.0


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the context constructor raises; nothing silently runs on the CPU."""
    from vo import _native
    try:
        c = _native.Context(0)
    except _native.VoError as e:
        assert "device" in str(e).lower() or "hip" in str(e).lower()
        return
    c.close()
    pytest.skip("GPU present")


def test_result_record_check_rejects_spliced_records():
    """The GPU writes a step's result record to mapped host memory, where its seven 64-byte lines arrive in no particular
    order: a copy may hold some lines of the slot's previous record.  vo_record_check (the test the pipeline's collect
    applies to every copy, 64 bits of check) must accept whole records only -- here every possible splice of an old and a
    new record at line granularity, deterministic."""
    import ctypes as C
    from vo import _native
    lib = _native.load()
    rng = np.random.default_rng(5)

    def record(seq, seed):
        r = _native.StepResult()
        raw = np.random.default_rng(seed).integers(0, 256, size=C.sizeof(r), dtype=np.uint8)
        C.memmove(C.byref(r), raw.ctypes.data, C.sizeof(r))
        lib.vo_record_seal(C.byref(r), seq)
        return r

    size = C.sizeof(_native.StepResult)
    lines = (size + 63) // 64
    old, new = record(41, 1), record(45, 2)
    assert lib.vo_record_check(C.byref(new), 45) == 1 and lib.vo_record_check(C.byref(old), 41) == 1
    assert lib.vo_record_check(C.byref(new), 41) == 0 and lib.vo_record_check(C.byref(old), 45) == 0
    ob, nb = bytes(old), bytes(new)
    for mask in range(1, 2 ** lines - 1):                      # every mix with at least one old and one new line
        mixed = b"".join((nb if (mask >> k) & 1 else ob)[64 * k:64 * (k + 1)] for k in range(lines))
        buf = C.create_string_buffer(mixed, size)
        assert lib.vo_record_check(buf, 45) == 0 and lib.vo_record_check(buf, 41) == 0, mask
    # records that differ in one field only (the common case: consecutive steps of a quiet stream)
    a = record(7, 3)
    b = _native.StepResult.from_buffer_copy(bytes(a))
    b.n_inliers += 1
    lib.vo_record_seal(C.byref(b), 11)
    ab, bb = bytes(a), bytes(b)
    for mask in range(1, 2 ** lines - 1):
        mixed = b"".join((bb if (mask >> k) & 1 else ab)[64 * k:64 * (k + 1)] for k in range(lines))
        if mixed in (ab, bb):
            continue
        buf = C.create_string_buffer(mixed, size)
        assert lib.vo_record_check(buf, 11) == 0 and lib.vo_record_check(buf, 7) == 0, mask
    # flipped bits anywhere in the data words are caught
    for _ in range(200):
        raw = bytearray(nb)
        k = int(rng.integers(0, size - 8))
        raw[k] ^= 1 << int(rng.integers(0, 8))
        assert lib.vo_record_check(C.create_string_buffer(bytes(raw), size), 45) == 0
