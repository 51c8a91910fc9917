"""Pose refinement oracle: the minimiser the reference's `_nonlinear_refinement`
(src/vo/pose_estimation/p3p.py:188-213) asks SciPy's `least_squares` for.

The reference minimises  sum_i || x_i - proj(K, R X_i + t) ||^2  (one residual per point, its
reprojection distance) over the twist of the pose (helpers.py:86-142) with TRF and a numerical
Jacobian, default tolerances 1e-8.  The objective does not depend on the parametrisation, so the
device kernel -- and this restatement -- run Gauss-Newton with the analytic Jacobian on the
left-multiplied increment  T <- [Exp(w) | v] T  until the next step would be below 1e-9 relative
(that step is not taken: the pose returned is the last one evaluated, its cost is exact).  SciPy
stops when the cost changes by < 1e-8 of itself, i.e. up to ~1e-4 away from the minimiser in
pose; tests compare against SciPy at its default tolerances (1e-4) and at tightened ones (1e-8).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import numpy as np


def exp_so3(w):
    th = float(np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]))
    Wx = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + Wx
    return np.eye(3) + (np.sin(th) / th) * Wx + ((1.0 - np.cos(th)) / (th * th)) * (Wx @ Wx)


def normal_equations(X, x, K, R, t):
    """A = sum J^T J (6x6), b = sum J^T e, cost = sum |e|^2 for e = x - proj; J = d proj / d (v, w)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    p = X @ R.T + t
    iz = 1.0 / p[:, 2]
    u = fx * p[:, 0] * iz + cx
    v = fy * p[:, 1] * iz + cy
    e = np.stack([x[:, 0] - u, x[:, 1] - v], axis=1)
    n = len(X)
    J = np.zeros((n, 2, 6))
    J[:, 0, 0] = fx * iz
    J[:, 0, 2] = -fx * p[:, 0] * iz * iz
    J[:, 1, 1] = fy * iz
    J[:, 1, 2] = -fy * p[:, 1] * iz * iz
    # d p / d w = -[p]_x
    for k in range(2):
        a, b, c = J[:, k, 0].copy(), J[:, k, 1].copy(), J[:, k, 2].copy()
        J[:, k, 3] = -b * p[:, 2] + c * p[:, 1]
        J[:, k, 4] = a * p[:, 2] - c * p[:, 0]
        J[:, k, 5] = -a * p[:, 1] + b * p[:, 0]
    A = np.einsum("nki,nkj->ij", J, J)
    b = np.einsum("nki,nk->i", J, e)
    return A, b, float(np.sum(e * e))


def refine_pose(X, x, K, R0, t0, max_iter=20, tol=1e-9):
    """Returns (R, t, iterations, cost).  X (N,3), x (N,2), world -> camera pose."""
    X = np.asarray(X, np.float64).reshape(-1, 3)
    x = np.asarray(x, np.float64).reshape(-1, 2)
    R = np.array(R0, np.float64).reshape(3, 3)
    t = np.array(t0, np.float64).reshape(3)
    if len(X) < 3:
        return R, t, 0, 0.0
    A, b, cost = normal_equations(X, x, K, R, t)
    it = 0
    while it < max_iter:
        try:
            L = np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            break
        d = np.linalg.solve(L.T, np.linalg.solve(L, b))
        E = exp_so3(d[3:])
        Rn, tn = E @ R, E @ t + d[:3]
        if np.sqrt(d @ d) <= tol * (1.0 + np.sqrt(tn @ tn)):   # converged: the step is not worth an evaluation
            break
        An, bn, costn = normal_equations(X, x, K, Rn, tn)
        if not costn <= cost:          # no decrease: keep the previous pose
            break
        R, t, A, b = Rn, tn, An, bn
        it += 1
        done = cost - costn <= 1e-16 * cost
        cost = costn
        if done:
            break
    return R, t, it, cost
