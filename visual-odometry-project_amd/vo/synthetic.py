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


def render(k, H, W, seed=2023, noise_sigma=2.0, want_depth=True, rows=32):
    """Frame k: (image uint8 (H, W), depth float32 (H, W) or None, T_world_cam (4, 4), K (3, 3)).
    Evaluated in blocks of `rows` image rows (cache-sized; every pixel's arithmetic and the order of the
    generator's draws are those of the whole-image evaluation, so the frames do not depend on `rows`)."""
    K = intrinsics(H, W)
    T = pose_world_cam(k)
    R, C = T[:3, :3], T[:3, 3]
    u = (np.arange(W, dtype=np.float64) - K[0, 2]) / K[0, 0]
    v = (np.arange(H, dtype=np.float64) - K[1, 2]) / K[1, 1]
    rng = np.random.default_rng(1000 + k + 7919 * (seed - 2023))
    img = np.empty((H, W), np.uint8)
    depth_out = np.empty((H, W), np.float32) if want_depth else None
    big = np.float64(1e30)
    for r0 in range(0, H, rows):
        r1 = min(H, r0 + rows)
        dx_c = np.broadcast_to(u, (r1 - r0, W))
        dy_c = np.broadcast_to(v[r0:r1, None], (r1 - r0, W))
        # ray directions in the world (camera z component is 1, so the ray parameter is the depth)
        dx = R[0, 0] * dx_c + R[0, 1] * dy_c + R[0, 2] * 1.0
        dy = R[1, 0] * dx_c + R[1, 1] * dy_c + R[1, 2] * 1.0
        dz = R[2, 0] * dx_c + R[2, 1] * dy_c + R[2, 2] * 1.0
        with np.errstate(divide="ignore", invalid="ignore"):
            planes = (
                np.where(dy > 0, (GROUND_Y - C[1]) / dy, big),
                np.where(dy < 0, (CEIL_Y - C[1]) / dy, big),
                np.where(dx > 0, (WALL_X - C[0]) / dx, big),
                np.where(dx < 0, (-WALL_X - C[0]) / dx, big),
                np.where(dz > 0, (END_Z - C[2]) / dz, big),
            )
        # nearest plane, the first one on ties (argmin over the planes)
        depth = planes[0].copy()
        surf = np.zeros(depth.shape, np.int64)
        for i in range(1, 5):
            m = planes[i] < depth
            depth[m] = planes[i][m]
            surf[m] = i
        px, py, pz = C[0] + depth * dx, C[1] + depth * dy, C[2] + depth * dz
        # surface coordinates (a, b) per plane
        a = np.where(surf <= 1, px, np.where(surf <= 3, py, px))
        b = np.where(surf <= 3, pz, py)
        # level of detail: cell size doubles every time the depth doubles past 20 m
        lod = np.floor(np.log2(np.maximum(depth / 20.0, 1.0))).astype(np.int64)
        cell = 0.25 * np.exp2(lod.astype(np.float64))
        ci = np.floor(a / cell).astype(np.int64)
        cj = np.floor(b / cell).astype(np.int64)
        tex = _hash_u8(ci, cj, surf * 16 + lod, seed)
        im = tex + rng.normal(0.0, noise_sigma, size=tex.shape).astype(np.float32)
        img[r0:r1] = np.clip(np.rint(im), 0, 255).astype(np.uint8)
        if want_depth:
            depth_out[r0:r1] = depth.astype(np.float32)
    return img, depth_out, T, K


def _render_image(args):
    k, H, W, seed = args
    return render(k, H, W, seed, want_depth=False)[0]


def _cache_file(job):
    import os
    d = os.environ.get("VO_SYNTH_CACHE")
    return os.path.join(d, "synth_%dx%d_seed%d_frame%d.npy" % (job[1], job[2], job[3], job[0])) if d else None


def render_images(jobs, workers=0):
    """Images of many frames, jobs = [(k, H, W, seed), ...], rendered by `workers` processes (spawned: safe to call
    from a process that holds a GPU context; 0 = in this process).  VO_SYNTH_CACHE=<dir>: frames found there are read
    instead of rendered, rendered ones are left there (tools/run_profiles.sh renders once for its dozen runs -- and keeps
    worker processes out of the profiler)."""
    import os
    out = [None] * len(jobs)
    todo = []
    for i, j in enumerate(jobs):
        f = _cache_file(j)
        if f and os.path.exists(f):
            out[i] = np.load(f)
        else:
            todo.append(i)
    if todo:
        sub = [jobs[i] for i in todo]
        if workers <= 1 or len(sub) < 4:
            imgs = [_render_image(j) for j in sub]
        else:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(min(workers, len(sub))) as pool:
                imgs = pool.map(_render_image, sub, chunksize=max(1, len(sub) // (4 * workers)))
        for i, im in zip(todo, imgs):
            out[i] = im
            f = _cache_file(jobs[i])
            if f:
                os.makedirs(os.path.dirname(f), exist_ok=True)
                tmp = f + ".tmp%d.npy" % os.getpid()
                np.save(tmp, im)
                os.replace(tmp, f)
    return out


class Stream:
    """n frames of one scene (`seed`), forward from frame `start`.  Frames are rendered when first asked for
    (`prefetch` renders many at once, in worker processes); depth maps only on request."""

    def __init__(self, n_frames, H, W, seed=2023, start=0):
        self.H, self.W, self.n, self.seed, self.start = H, W, n_frames, seed, start
        self.K = intrinsics(H, W)
        self._img, self._depth = {}, {}

    def prefetch(self, frames=None, workers=0):
        todo = [i for i in (range(self.n) if frames is None else frames) if i not in self._img]
        for i, im in zip(todo, render_images([(self.start + i, self.H, self.W, self.seed) for i in todo], workers)):
            self._img[i] = im
        return self

    def image(self, i):
        if i not in self._img:
            if not 0 <= i < self.n:
                raise IndexError(i)
            self._img[i] = _render_image((self.start + i, self.H, self.W, self.seed))
        return self._img[i]

    def depth(self, i):
        if i not in self._depth:
            if not 0 <= i < self.n:
                raise IndexError(i)
            self._depth[i] = render(self.start + i, self.H, self.W, self.seed)[1]
        return self._depth[i]

    def T_world_cam(self, i):
        return pose_world_cam(self.start + i)

    def order(self, steps):
        idx, d, out = 0, 1, [0]
        for _ in range(steps):
            if idx + d < 0 or idx + d >= self.n:
                d = -d
            idx += d
            out.append(idx)
        return out
