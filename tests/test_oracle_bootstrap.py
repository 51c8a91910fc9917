"""oracle/bootstrap_np.py against the reference's own outputs (tests/golden/bootstrap.npz: 8-point F, E, the four
decompositions, the relative pose with and without RANSAC; tools/make_golden.py golden_bootstrap)."""
import os

import numpy as np

from oracle import bootstrap_np as bo

G = os.path.join(os.path.dirname(__file__), "golden")


def test_bootstrap_oracle_equals_reference_golden():
    g = np.load(os.path.join(G, "bootstrap.npz"))
    K = g["K"]
    pn, T = bo.normalize_points(g["x1"])
    assert np.allclose(pn, g["x1_norm"], rtol=0, atol=1e-13) and np.allclose(T, g["T1"], rtol=1e-13)
    F = bo.find_fundamental_matrix(g["x1"], g["x2"])
    assert np.allclose(F, g["F"], rtol=1e-9, atol=1e-12)
    assert np.allclose(K.T @ F @ K, g["E"], rtol=1e-9, atol=1e-9)
    assert np.allclose(bo.decompose_essential_matrix(g["E"]), g["M4"], atol=1e-12)
    M, X, _, _ = bo.find_relative_pose(g["x1"], g["x2"], K, K, F)
    assert np.allclose(M, g["M"], atol=1e-9) and np.allclose(X, g["X_tri"], rtol=1e-7, atol=1e-7)
    # RANSAC route: the reference's generator, sampler and sequential rule -> the golden's inlier mask
    Fr, inl, _ = bo.find_fundamental_matrix_ransac(g["x1"], g["x2_outliers"], 1e-3, 0.5, 0.99)
    Mr, Xr, mask, _ = bo.find_relative_pose(g["x1"], g["x2_outliers"], K, K, Fr, inl)
    assert np.array_equal(mask, g["inliers_ransac"]) and np.allclose(Mr, g["M_ransac"], atol=1e-9)
    assert np.allclose(Xr, g["X_ransac"], rtol=1e-6, atol=1e-6)
