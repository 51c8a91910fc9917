"""Adaptive RANSAC (reference: src/vo/algorithms/ransac.py).

`find_best_model` keeps the reference's generic callable interface (model_fn /
error_fn called once per iteration).  `find_best_model_batched` is the MI355X route:
hypotheses for a whole batch of pre-drawn samples are produced by one call (on the
GPU), and the reference's sequential accept / adaptive-bound rule is replayed over
their (valid, inlier count) so the result, the number of generator draws consumed and
the persistent state (rng, n_iterations, outlier_ratio) are what the sequential loop
would have produced."""
from typing import Callable

import numpy as np


class RANSAC:
    def __init__(self, s_points: int, population, model_fn: Callable, error_fn: Callable, inlier_threshold: float,
                 outlier_ratio: float = 0.9, confidence: float = 0.99, max_iterations: int = np.inf,
                 adaptive: bool = True, p3p: bool = False) -> None:
        self.s = s_points
        self.population = np.array(population)
        self.model_fn = model_fn
        self.error_fn = error_fn
        self.inlier_threshold = inlier_threshold
        self.outlier_ratio = outlier_ratio
        self.confidence = confidence
        self.adaptive = adaptive
        self.p3p = p3p
        self.rng = np.random.default_rng(2023)                       # ransac.py:52
        self.max_iterations = max_iterations
        self.n_iterations = min(max_iterations, self.compute_n_iterations())

    def compute_n_iterations(self) -> int:
        """ransac.py:58-67."""
        k = np.ceil(np.log(1 - self.confidence) / np.log(1 - (1 - self.outlier_ratio) ** self.s))
        return int(k)

    def _accept(self, n_inliers: int) -> None:
        """Adaptive update after a new best model (ransac.py:113-120)."""
        if self.adaptive:
            self.outlier_ratio = min(max(1 - n_inliers / len(self.population), 0.01), 0.99)
            self.n_iterations = int(min(self.max_iterations, self.compute_n_iterations()))

    def find_best_model(self, population=None):
        """Sequential loop, one model_fn / error_fn call per iteration (ransac.py:69-129)."""
        best_n, best_inliers, best_model, n = -1, None, None, 0
        if population is not None:
            self.population = np.array(population)
        assert self.population is not None, "Population must be provided"
        while n < self.n_iterations:
            idxs = self.rng.choice(np.arange(len(self.population)), replace=False, size=self.s)
            model = self.model_fn(self.population[idxs])
            if model is None:                                         # not counted as an iteration
                continue
            inliers = self.error_fn(model, self.population) < self.inlier_threshold
            n_inliers = inliers.sum()
            if n_inliers > best_n:
                best_n, best_inliers, best_model = n_inliers, inliers, model
                self._accept(best_n)
            n += 1
        if not self.p3p:                                              # refit on all inliers
            best_model = self.model_fn(self.population[best_inliers])
        return best_model, best_inliers

    def find_best_model_batched(self, n_population: int, batch_fn: Callable, batch_size: int = 1000):
        """Same result as find_best_model for a population of `n_population` items when
        ``batch_fn(samples (B, s) int32) -> (valid (B,), counts (B,), fetch)`` evaluates B
        hypotheses at once and ``fetch(b) -> (model, inlier mask)`` returns hypothesis b.
        Returns (model, mask, iterations counted)."""
        import ctypes as C
        from vo import _native
        lib = _native.load()
        self.population = np.arange(n_population)
        # the sequential accept / adapt rule over the batch's (valid, count) is vo_ransac_replay (csrc/ransac_host.hip: the
        # loop of ransac.py:90-121 with the reference's own formula for the bound) -- a Python loop over 2000 hypotheses per
        # batch cost the two-view bootstrap more than its kernels
        st = _native.RansacState(float(self.outlier_ratio), float(self.confidence),
                                 -1 if self.max_iterations == np.inf else int(self.max_iterations),
                                 int(min(self.n_iterations, 2 ** 62)), int(self.s), 1 if self.adaptive else 0)
        n_done, best_count, best_idx = C.c_int64(0), C.c_int32(-1), C.c_int32(-1)
        consumed, finished = C.c_int(0), C.c_int(0)
        best, batches = None, 0
        while not finished.value:
            spec = _native.Pcg64.from_generator(self.rng)             # speculative copy of the generator
            samples = _native.rng_choice(spec, n_population, self.s, batch_size)
            valid, counts, fetch = batch_fn(samples)
            valid = np.ascontiguousarray(valid, np.uint8)
            counts = np.ascontiguousarray(counts, np.int32)
            before = best_idx.value
            rc = lib.vo_ransac_replay(C.byref(st), valid.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                      int(batch_size), int(n_population), C.byref(n_done), C.byref(best_count),
                                      C.byref(best_idx), int(batches * batch_size), C.byref(consumed), C.byref(finished))
            if rc != 0:
                raise _native.VoError(rc, "vo_ransac_replay")
            if best_idx.value != before:
                best = fetch(best_idx.value - batches * batch_size)
            real = _native.Pcg64.from_generator(self.rng)             # advance by exactly what was consumed
            _native.rng_choice(real, n_population, self.s, consumed.value)
            real.to_generator(self.rng)
            batches += 1
        self.outlier_ratio, self.n_iterations = float(st.outlier_ratio), int(st.n_iterations)
        return best[0], best[1], int(n_done.value)
