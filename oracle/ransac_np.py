"""NumPy oracle: adaptive RANSAC loop.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates src/vo/algorithms/ransac.py:15-129 (RANSAC) and the way
P3PPoseEstimator configures it (src/vo/pose_estimation/p3p.py:110-121, 167-175).
Pinned by tests/golden/ransac.npz (sampler stream, iteration-bound table, a full
find_best_model trace including the state that persists between calls).
"""
import numpy as np


def num_iterations(confidence, outlier_ratio, s):
    """ransac.py:58-67."""
    return int(np.ceil(np.log(1 - confidence) / np.log(1 - (1 - outlier_ratio) ** s)))


class Ransac:
    def __init__(self, s_points, population, model_fn, error_fn, inlier_threshold, outlier_ratio=0.9,
                 confidence=0.99, max_iterations=np.inf, adaptive=True, p3p=False):
        self.s = s_points
        self.population = np.array(population)
        self.model_fn, self.error_fn = model_fn, error_fn
        self.inlier_threshold = inlier_threshold
        self.outlier_ratio, self.confidence = outlier_ratio, confidence
        self.adaptive, self.p3p = adaptive, p3p
        self.rng = np.random.default_rng(2023)                      # ransac.py:52
        self.max_iterations = max_iterations
        self.n_iterations = min(max_iterations, num_iterations(confidence, outlier_ratio, s_points))
        self.trace = []                                             # (indices, model is None, n_inliers)

    def find_best_model(self, population=None):
        best_n, best_inl, best_model, n = -1, None, None, 0
        if population is not None:
            self.population = np.array(population)
        while n < self.n_iterations:                                # ransac.py:90
            idx = self.rng.choice(np.arange(len(self.population)), replace=False, size=self.s)
            model = self.model_fn(self.population[idx])
            if model is None:                                       # not counted (ransac.py:99-101)
                self.trace.append((idx, True, -1))
                continue
            inl = self.error_fn(model, self.population) < self.inlier_threshold
            cnt = int(inl.sum())
            self.trace.append((idx, False, cnt))
            if cnt > best_n:
                best_n, best_inl, best_model = cnt, inl, model
                if self.adaptive:                                   # ransac.py:114-120
                    self.outlier_ratio = min(max(1 - best_n / len(self.population), 0.01), 0.99)
                    self.n_iterations = int(min(self.max_iterations,
                                                num_iterations(self.confidence, self.outlier_ratio, self.s)))
            n += 1
        if not self.p3p:                                            # ransac.py:125-127
            best_model = self.model_fn(self.population[best_inl])
        self.iterations_done = n
        return best_model, best_inl


def p3p_ransac(X, x, K, inlier_threshold, outlier_ratio=0.9, confidence=0.99, max_iterations=10000):
    """RANSAC as P3PPoseEstimator builds it (p3p.py:110-121) around the C oracle's
    P3P solve and reprojection error.  Returns the Ransac object; call
    ``.find_best_model(pop)`` with ``pop = np.arange(N)``-indexed correspondences."""
    from oracle import native
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 2)

    def model_fn(idx):
        idx = np.asarray(idx).reshape(-1)
        return native.p3p_solve(X[idx], x[idx], K)

    def error_fn(model, pop):
        return native.reproj_errors(X, x, K, model[0], model[1])

    return Ransac(4, np.arange(len(X)), model_fn, error_fn, inlier_threshold, outlier_ratio, confidence,
                  max_iterations, adaptive=True, p3p=True)
