"""Randomised check used by tests/test_gpu_pipeline.py and tools/dev/fuzz_sequences.py: S sequences through one pipeline
(look-ahead, forced faults, detector margins drawn at random) against S one-sequence pipelines stepped one frame at a
time with the detector on every frame.  Returns the number of trials that differed."""
import numpy as np


def run_trials(ctx, seed, trials, verbose=False, big=False):
    from vo import _native, synthetic
    from test_gpu_pipeline import start_state, fields
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        H, W, N, hyp, F = 240, 320, 300, 256, 5
        if big and rng.random() < 0.3:
            H, W, N, hyp = 480, 640, 500, int(rng.choice([64, 512]))
        S = int(rng.integers(2, 5))
        seeds = rng.integers(2023, 12000, size=S)
        fracs = rng.uniform(0.6, 1.0, size=S)
        steps = int(rng.integers(6, 14))
        fault_every = int(rng.choice([0, 0, 3, 5]))
        margin = float(rng.choice([0.02, 0.1, -1.0, 0.005]))
        never = int(rng.random() < 0.25)
        streams = [synthetic.Stream(F, H, W, seed=int(seeds[q]), start=q) for q in range(S)]
        starts = [start_state(streams[q], N, float(fracs[q])) for q in range(S)]
        order = streams[0].order(steps)
        pairs = list(zip(order[:-1], order[1:]))
        single = []
        for q in range(S):
            pipe = _native.Pipeline(ctx, H, W, F, streams[q].K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                                    p3p_threshold=1.0, max_iterations=1000, refine_iters=20, detect_margin=-1.0)
            for i in range(F):
                pipe.set_frame(i, streams[q].image(i))
            pipe.set_state(0, starts[q][0], starts[q][1], starts[q][1])
            try:
                res = [pipe.step(a, b) for a, b in pairs]
            except _native.VoError:
                res = None
            single.append((res, pipe.get_state() if res else None))
            pipe.close()
        if any(r[0] is None for r in single):
            if verbose:
                print("trial", trial, "skipped (a sequence lost track)")
            continue
        pipe = _native.Pipeline(ctx, H, W, F, streams[0].K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                                p3p_threshold=1.0, max_iterations=1000, refine_iters=20, sequences=S,
                                debug_fault_every=fault_every, detect_margin=margin, debug_never_detect=never)
        for q in range(S):
            for i in range(F):
                pipe.set_frame(i, streams[q].image(i), seq=q)
            pipe.set_state(0, starts[q][0], starts[q][1], starts[q][1], seq=q)
        got = []
        pipe.submit(*pairs[0])
        for k in range(len(pairs)):
            if k + 1 < len(pairs):
                pipe.submit(*pairs[k + 1])
            got.append(pipe.collect_all())
        ok = True
        for q in range(S):
            for k in range(len(pairs)):
                if fields(got[k][q]) != fields(single[q][0][k]):
                    ok = False
                    if verbose:
                        names = ("n_features_in redetected n_tracked n_triangulated n_inliers ransac_iterations draws_consumed "
                                 "n_candidates n_dropped n_landmarks R t R_refined t_refined T_wc").split()
                        fa, fb = fields(got[k][q]), fields(single[q][0][k])
                        diff = [(n, a, b) for n, a, b in zip(names, fa, fb) if a != b and not isinstance(a, tuple)]
                        diff += [n for n, a, b in zip(names, fa, fb) if a != b and isinstance(a, tuple)]
                        print("  MISMATCH trial", trial, "seq", q, "step", k, "recovered", got[k][q].recovered, "reason",
                              got[k][q].reserved, "detector_ran", got[k][q].detector_ran, single[q][0][k].detector_ran, diff)
                    break
            st = pipe.get_state(seq=q)
            for key in ("keypoints", "state", "candidate_mask", "landmarks", "tracks", "poses", "curr_pose"):
                if not np.array_equal(st[key], single[q][1][key], equal_nan=True):
                    ok = False
                    if verbose:
                        print("  STATE MISMATCH trial", trial, "seq", q, key)
        if verbose:
            rec = sum(r.recovered for rs in got for r in rs)
            red = sum(r.redetected for rs in got for r in rs)
            print("trial %d %dx%d hyp=%d S=%d steps=%d fault_every=%d margin=%g never=%d: %s (host-path steps %d, re-detects %d)" % (
                trial, H, W, hyp, S, steps, fault_every, margin, never, "ok" if ok else "FAILED", rec, red), flush=True)
        bad += 0 if ok else 1
        pipe.close()
    return bad
