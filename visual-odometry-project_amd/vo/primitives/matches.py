"""Correspondences between two frames (reference: src/vo/primitives/matches.py).

Constructing a Matches object re-orders BOTH frames' features in place into
    [ triangulated | matched | newly matched | unmatched ]
(groups decided by frame 1's state, order inside a group = order of the match list)
and carries landmarks, track starts and track start poses from frame 1 to frame 2."""
import numpy as np

from vo.primitives.frame import Frame


def _nan(shape):
    return np.full(shape, np.nan)


class Matches:
    def __init__(self, frame1: Frame, frame2: Frame, matches: np.ndarray):
        self.frame1 = frame1
        self.frame2 = frame2
        self.newly_matched_idx = None
        self._threshold = 0.1
        f1, f2 = frame1.features, frame2.features

        st = f1.state[matches[:, 0]]
        groups = (st == 2, st == 1, st == 0)                          # triangulated, matched, newly matched
        g1 = [matches[:, 0][g] for g in groups]
        g2 = [matches[:, 1][g] for g in groups]
        rest1 = np.delete(np.arange(0, len(f1.keypoints)), np.concatenate(g1))
        rest2 = np.delete(np.arange(0, len(f2.keypoints)), np.concatenate(g2))
        n_tri, n_mat, n_new = (len(g) for g in g1)
        order1 = g1 + [rest1]
        order2 = g2 + [rest2]

        def regroup(arr, order):
            return np.concatenate([arr[i] for i in order], axis=0)

        # ---- frame 1 (matches.py:39-110) ----
        kp1 = regroup(f1.keypoints, order1)
        desc1 = None if f1.descriptors is None else regroup(f1.descriptors, order1)
        land1 = regroup(f1.landmarks, order1)
        state1 = np.concatenate((2 * np.ones_like(g1[0]), np.ones_like(g1[1]), np.ones_like(g1[2]),
                                 0 * np.ones_like(rest1)))
        assert not np.any(np.isnan(f1.tracks[g1[1]])), "NaN in matched tracks"
        tracks1 = np.concatenate((_nan((n_tri, 2, 1)), f1.tracks[g1[1]], f1.keypoints[g1[2]],
                                  _nan((len(rest1), 2, 1))))
        poses1 = np.concatenate((_nan((n_tri, 4, 4)), f1.poses[g1[1]], f1.poses[g1[2]],
                                 _nan((len(rest1), 4, 4))))
        assert not np.any(np.isnan(land1[state1 == 2])), "NaN in triangulated landmarks"
        f1.keypoints, f1.state, f1.descriptors = kp1, state1, desc1
        f1.landmarks, f1.tracks, f1.poses = land1, tracks1, poses1

        # ---- frame 2 (matches.py:112-212); track data comes positionally from re-ordered frame 1 ----
        kp2 = regroup(f2.keypoints, order2)
        desc2 = None if f1.descriptors is None else regroup(f2.descriptors, order2)
        land2 = np.concatenate([f1.landmarks[:n_tri], f2.landmarks[g2[1]], f2.landmarks[g2[2]],
                                f2.landmarks[rest2]], axis=0)
        state2 = np.concatenate((2 * np.ones_like(g2[0]), np.ones_like(g2[1]), np.ones_like(g2[2]),
                                 np.zeros_like(rest2)))
        a, b = n_tri, n_tri + n_mat
        tracks2 = np.concatenate((_nan((n_tri, 2, 1)), f1.tracks[a:b], f1.tracks[b:b + n_new],
                                  kp2[b + n_new:]))                    # unmatched: a new track starts here
        poses2 = np.concatenate((_nan((n_tri, 4, 4)), f1.poses[a:b], f1.poses[b:b + n_new],
                                 _nan((len(rest2), 4, 4))))
        assert len(tracks2) == len(kp2), "Length of tracks and keypoints do not match"
        assert not np.any(np.isnan(land2[state2 == 2])), "NaN in triangulated landmarks"
        f2.keypoints, f2.state, f2.descriptors = kp2, state2, desc2
        f2.landmarks, f2.tracks, f2.poses = land2, tracks2, poses2
