"""Deterministic KITTI-shaped synthetic stream.

A textured corridor (ground y=+1.65 m, ceiling y=-4 m, walls x=+-6 m, end wall
z=+400 m) seen by a pinhole camera that drives forward 0.8 m per frame with a small
yaw oscillation.  Every pixel is the exact ray/plane intersection, so depth maps and
camera poses are analytic ground truth.  Texture = hash of the integer cell the hit
point falls in (cell size 0.25 m, doubled with distance so cells stay several
pixels wide) + per-frame Gaussian noise, quantised to uint8.  No files, no network.
"""
import numpy as np

GROUND_Y, CEIL_Y, WALL_X, END_Z = 1.65, -4.0, 6.0, 400.0
STEP_Z = 0.8


def intrinsics(H, W):
    """fx = fy for an 80 degree horizontal field of view (KITTI's is ~82)."""
    f = 0.5 * W / np.tan(np.deg2rad(40.0))
    return np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]])


def pose_world_cam(k):
    """Camera-to-world transform of frame k (4x4)."""
    yaw = 0.01 * np.sin(0.1 * k)
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    T[:3, 3] = [0.0, 0.0, STEP_Z * k]
    return T


def _hash_u8(i, j, surf, seed):
    h = (i.astype(np.int64) * 73856093) ^ (j.astype(np.int64) * 19349663) ^ (np.int64(surf) * 83492791) ^ np.int64(seed)
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(33)
    h *= np.uint64(0xFF51AFD7ED558CCD)
    h ^= h >> np.uint64(33)
    h *= np.uint64(0xC4CEB9FE1A85EC53)
    h ^= h >> np.uint64(33)
    return (h & np.uint64(0xFF)).astype(np.float32)


def render(k, H, W, seed=2023, noise_sigma=2.0):
    """Frame k: (image uint8 (H, W), depth float32 (H, W), T_world_cam (4, 4), K (3, 3))."""
    K = intrinsics(H, W)
    T = pose_world_cam(k)
    R, C = T[:3, :3], T[:3, 3]
    u = (np.arange(W, dtype=np.float64) - K[0, 2]) / K[0, 0]
    v = (np.arange(H, dtype=np.float64) - K[1, 2]) / K[1, 1]
    dx_c, dy_c = np.meshgrid(u, v)
    dz_c = np.ones_like(dx_c)
    # ray directions in the world (camera z component is 1, so the ray parameter is the depth)
    dx = R[0, 0] * dx_c + R[0, 1] * dy_c + R[0, 2] * dz_c
    dy = R[1, 0] * dx_c + R[1, 1] * dy_c + R[1, 2] * dz_c
    dz = R[2, 0] * dx_c + R[2, 1] * dy_c + R[2, 2] * dz_c
    big = np.float64(1e30)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_planes = [
            np.where(dy > 0, (GROUND_Y - C[1]) / dy, big),
            np.where(dy < 0, (CEIL_Y - C[1]) / dy, big),
            np.where(dx > 0, (WALL_X - C[0]) / dx, big),
            np.where(dx < 0, (-WALL_X - C[0]) / dx, big),
            np.where(dz > 0, (END_Z - C[2]) / dz, big),
        ]
    t = np.stack(t_planes)
    surf = np.argmin(t, axis=0)
    depth = np.min(t, axis=0)
    px, py, pz = C[0] + depth * dx, C[1] + depth * dy, C[2] + depth * dz
    # surface coordinates (a, b) per plane
    a = np.select([surf <= 1, surf <= 3], [px, py], default=px)
    b = np.select([surf <= 1, surf <= 3], [pz, pz], default=py)
    # level of detail: cell size doubles every time the depth doubles past 20 m
    lod = np.floor(np.log2(np.maximum(depth / 20.0, 1.0))).astype(np.int64)
    cell = 0.25 * np.exp2(lod.astype(np.float64))
    ci = np.floor(a / cell).astype(np.int64)
    cj = np.floor(b / cell).astype(np.int64)
    tex = _hash_u8(ci, cj, surf * 16 + lod, seed)
    rng = np.random.default_rng(1000 + k + 7919 * (seed - 2023))
    img = tex + rng.normal(0.0, noise_sigma, size=tex.shape).astype(np.float32)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return img, depth.astype(np.float32), T, K


class Stream:
    """n frames rendered up front; ``order(steps)`` walks them back and forth so that
    consecutive frames are always neighbours (any number of steps from n frames)."""

    def __init__(self, n_frames, H, W, seed=2023, start=0):
        self.H, self.W, self.n = H, W, n_frames
        self.frames = [render(start + k, H, W, seed) for k in range(n_frames)]
        self.K = self.frames[0][3]

    def image(self, i):
        return self.frames[i][0]

    def depth(self, i):
        return self.frames[i][1]

    def T_world_cam(self, i):
        return self.frames[i][2]

    def order(self, steps):
        idx, d, out = 0, 1, [0]
        for _ in range(steps):
            if idx + d < 0 or idx + d >= self.n:
                d = -d
            idx += d
            out.append(idx)
        return out
