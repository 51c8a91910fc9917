"""Pose-refinement oracle (oracle/refine_np.py) against what the reference actually calls:
scipy.optimize.least_squares over the twist with one residual per point (p3p.py:188-213,
helpers.py:86-142), restated here with SciPy itself."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.optimize import least_squares

from oracle import refine_np

K = np.array([[700.0, 0.0, 688.0], [0.0, 700.0, 620.0], [0.0, 0.0, 1.0]])


def scene(rng, n, noise):
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-4, 4, n), rng.uniform(4, 40, n)], 1)
    w, t = rng.normal(0, 0.05, 3), rng.normal(0, 0.3, 3)
    R = refine_np.exp_so3(w)
    p = X @ R.T + t
    x = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
    x = x + rng.normal(0, noise, (n, 2))
    R0, t0 = refine_np.exp_so3(w + rng.normal(0, 0.01, 3)), t + rng.normal(0, 0.05, 3)
    return X, x, R, t, R0, t0


def twist_to_H(tw):
    v, w = tw[:3], tw[3:]
    S = np.array([[0, -w[2], w[1], v[0]], [w[2], 0, -w[0], v[1]], [-w[1], w[0], 0, v[2]], [0, 0, 0, 0]])
    return expm(S)


def scipy_refine(X, x, R0, t0, **kw):
    def res(tw):
        H = twist_to_H(tw)
        q = X @ H[:3, :3].T + H[:3, 3]
        u = np.stack([K[0, 0] * q[:, 0] / q[:, 2] + K[0, 2], K[1, 1] * q[:, 1] / q[:, 2] + K[1, 2]], 1)
        return np.linalg.norm(x - u, axis=1)
    H0 = np.eye(4)
    H0[:3, :3], H0[:3, 3] = R0, t0
    S = np.real(logm(H0))
    sol = least_squares(res, np.array([S[0, 3], S[1, 3], S[2, 3], -S[1, 2], S[0, 2], -S[0, 1]]), **kw).x
    H = twist_to_H(sol)
    return H[:3, :3], H[:3, 3], float(np.sum(res(sol) ** 2))


@pytest.mark.parametrize("n,noise", [(1000, 0.5), (200, 0.5), (40, 1.0), (12, 0.3)])
def test_gauss_newton_reaches_what_scipy_is_asked_for(n, noise):
    rng = np.random.default_rng(n)
    X, x, R, t, R0, t0 = scene(rng, n, noise)
    Rg, tg, it, cost = refine_np.refine_pose(X, x, K, R0, t0)
    assert 1 <= it <= 8
    # default tolerances (what the reference runs): within the 1e-4 the north star allows
    Rs, ts, cs = scipy_refine(X, x, R0, t0)
    assert np.abs(Rs - Rg).max() < 1e-4 and np.abs(ts - tg).max() < 1e-4 * (1 + np.abs(tg).max())
    assert cost <= cs * (1 + 1e-12), "Gauss-Newton must not end above SciPy's cost"
    # tightened: the same minimum (SciPy's finite-difference Jacobian limits how close it gets) ...
    Rt, tt, ct = scipy_refine(X, x, R0, t0, ftol=1e-15, xtol=1e-15, gtol=1e-15)
    assert np.abs(Rt - Rg).max() < 5e-5 and np.abs(tt - tg).max() < 5e-4
    assert cost <= ct * (1 + 1e-12)
    # ... and a stationary point of the objective: the gradient J^T e vanishes there
    A, b, _ = refine_np.normal_equations(X, x, K, Rg, tg)
    assert np.abs(np.linalg.solve(A, b)).max() < 1e-10
    assert np.allclose(Rg @ Rg.T, np.eye(3), atol=1e-12)
    # and close to the truth the data came from
    assert np.abs(Rg - R).max() < 5e-3 and np.abs(tg - t).max() < 0.1


def test_degenerate_inputs_leave_the_pose_alone():
    rng = np.random.default_rng(0)
    X, x, R, t, R0, t0 = scene(rng, 2, 0.1)
    Rg, tg, it, cost = refine_np.refine_pose(X, x, K, R0, t0)
    assert it == 0 and np.array_equal(Rg, R0) and np.array_equal(tg, t0)
