"""Synthetic inputs for tests and bench.  No reference image is used.

A frame is an analytic texture A(x,y) = 128 + sum_k a_k sin(2*pi*f_k*(x cos t_k + y sin t_k) + p_k)
+ Gaussian blobs, so a second frame B(x,y) = A(T^-1(x,y)) for a known similarity T has exact
ground-truth flow p -> T(p).  Seed 0x3D52.
"""
import numpy as np

SEED = 0x3D52


class Texture:
    def __init__(self, w, h, seed=SEED, n_waves=48, n_blobs=150, channels=1):
        rng = np.random.default_rng(seed)
        self.w, self.h, self.channels = w, h, channels
        self.waves = []
        self.blobs = []
        for _ in range(channels):
            f = np.exp(rng.uniform(np.log(1 / 96.0), np.log(1 / 6.0), n_waves))
            th = rng.uniform(0, np.pi, n_waves)
            ph = rng.uniform(0, 2 * np.pi, n_waves)
            a = 1.0 / f
            a *= 40.0 / np.sqrt(0.5 * np.sum(a * a))         # sigma of the sum ~ 40
            self.waves.append((f, th, ph, a))
            bx = rng.uniform(0, w, n_blobs); by = rng.uniform(0, h, n_blobs)
            bs = rng.uniform(1.5, 6.0, n_blobs); ba = rng.uniform(-60, 60, n_blobs)
            self.blobs.append((bx, by, bs, ba))

    def _eval(self, c, X, Y):
        f, th, ph, a = self.waves[c]
        v = np.full(X.shape, 128.0)
        for k in range(len(f)):
            v += a[k] * np.sin(2 * np.pi * f[k] * (X * np.cos(th[k]) + Y * np.sin(th[k])) + ph[k])
        bx, by, bs, ba = self.blobs[c]
        for k in range(len(bx)):
            r = 4.5 * bs[k]
            m = (np.abs(X - bx[k]) < r) & (np.abs(Y - by[k]) < r)
            if m.any():
                v[m] += ba[k] * np.exp(-((X[m] - bx[k]) ** 2 + (Y[m] - by[k]) ** 2) / (2 * bs[k] ** 2))
        return v

    def render(self, T=None, w=None, h=None):
        """u8 image (h,w) or (h,w,C) of the texture moved by the 3x3 similarity/affine T
        (B(x) = A(T^-1 x)); T=None renders A itself."""
        w = self.w if w is None else w
        h = self.h if h is None else h
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
        if T is not None:
            Ti = np.linalg.inv(np.asarray(T, np.float64))
            Wd = Ti[2, 0] * xs + Ti[2, 1] * ys + Ti[2, 2]          # 1 for the similarities; a plane-induced homography divides
            X = (Ti[0, 0] * xs + Ti[0, 1] * ys + Ti[0, 2]) / Wd
            Y = (Ti[1, 0] * xs + Ti[1, 1] * ys + Ti[1, 2]) / Wd
        else:
            X, Y = xs, ys
        out = [np.clip(np.rint(self._eval(c, X, Y)), 0, 255).astype(np.uint8) for c in range(self.channels)]
        return out[0] if self.channels == 1 else np.stack(out, -1)


def similarity(w, h, tx, ty, rot_deg, zoom):
    """translation (tx,ty) + rotation about the image centre + zoom about the centre"""
    cx, cy = 0.5 * (w - 1), 0.5 * (h - 1)
    c, s = np.cos(np.deg2rad(rot_deg)) * zoom, np.sin(np.deg2rad(rot_deg)) * zoom
    return np.array([[c, -s, cx - c * cx + s * cy + tx],
                     [s, c, cy - s * cx - c * cy + ty],
                     [0, 0, 1.0]])


def apply_T(T, xy):
    xy = np.asarray(xy, np.float64)
    return np.stack([T[0, 0] * xy[:, 0] + T[0, 1] * xy[:, 1] + T[0, 2],
                     T[1, 0] * xy[:, 0] + T[1, 1] * xy[:, 1] + T[1, 2]], -1)


KITTI_W, KITTI_H = 1241, 376
KITTI_MOTION = dict(tx=3.25, ty=-1.50, rot_deg=0.3, zoom=1.01)      # cfg 1/2/4
HD_MOTION = dict(tx=7.5, ty=2.25, rot_deg=0.0, zoom=1.02)            # cfg 5


def frame_pair(w=KITTI_W, h=KITTI_H, motion=None, seed=SEED):
    """(A, B, T): two u8 frames and the ground-truth transform with B(x) = A(T^-1 x)."""
    motion = KITTI_MOTION if motion is None else motion
    tex = Texture(w, h, seed)
    T = similarity(w, h, **motion)
    return tex.render(), tex.render(T), T


def frame_stream(n_frames, w=KITTI_W, h=KITTI_H, motion=None, seed=SEED):
    """n_frames u8 frames with cumulative motion T^i, and the per-step transform T."""
    motion = KITTI_MOTION if motion is None else motion
    tex = Texture(w, h, seed)
    T = similarity(w, h, **motion)
    frames, Ti = [], np.eye(3)
    for _ in range(n_frames):
        frames.append(tex.render(None if np.allclose(Ti, np.eye(3)) else Ti))
        Ti = T @ Ti
    return frames, T


def grid_points(w, h, cell, n=None, border=8, seed=SEED):
    """deterministic jittered grid (one point per cell, mirrors src/features.cpp:15-16 with
    cell_size from src/config.cpp:10), >= border px from the image edge, optionally thinned to n."""
    rng = np.random.default_rng(seed + 1)
    cols, rows = -(-w // cell), -(-h // cell)
    gy, gx = np.mgrid[0:rows, 0:cols]
    x = (gx + rng.uniform(0.15, 0.85, gx.shape)) * cell
    y = (gy + rng.uniform(0.15, 0.85, gy.shape)) * cell
    pts = np.stack([x.ravel(), y.ravel()], -1)
    pts[:, 0] = np.clip(pts[:, 0], border, w - 1 - border)
    pts[:, 1] = np.clip(pts[:, 1], border, h - 1 - border)
    if n is not None and n < len(pts):
        keep = np.sort(rng.choice(len(pts), n, replace=False))
        pts = pts[keep]
    return pts.astype(np.float32)


def panorama_tile(w, h, seed=SEED):
    """3-channel u8 tile (decorrelated channels) for the warp workload"""
    return Texture(w, h, seed, channels=3).render()


def tile_homography(i, w, h, seed=SEED):
    """H_i = T(offset) * R(+-2 deg) * perspective(1e-5): forward map tile -> canvas (f32 3x3)"""
    rng = np.random.default_rng(seed + 100 + i)
    ang = np.deg2rad(rng.uniform(-2, 2))
    c, s = np.cos(ang), np.sin(ang)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    P = np.array([[1, 0, 0], [0, 1, 0], [rng.uniform(-1e-5, 1e-5), rng.uniform(-1e-5, 1e-5), 1.0]])
    Tr = np.array([[1, 0, rng.uniform(20, 120)], [0, 1, rng.uniform(20, 120)], [0, 0, 1.0]])
    return (Tr @ R @ P).astype(np.float32)


# ---- sparse image alignment: a textured plane seen from two camera poses ---------------------------------
def rodrigues(w):
    """rotation matrix of the rotation vector w (f64)"""
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def plane_scene(w, h, K, R, t, n=(0.05, -0.1, 1.0), d=12.0, seed=SEED):
    """Two views of a textured plane n . X = d (reference camera frame): returns (ref image, cur image, H) where the current
    camera sees X_cur = R X_ref + t.  A plane induces the homography H = K (R + t n^T / d) K^-1 between the views, so the second
    image is the first one warped by H -- rendered analytically from the texture, no resampling of pixels."""
    n = np.asarray(n, np.float64)
    H = K @ (R + np.outer(t, n) / d) @ np.linalg.inv(K)
    tex = Texture(w, h, seed)
    return tex.render(), tex.render(H / H[2, 2]), H


def plane_points(uv, K, n=(0.05, -0.1, 1.0), d=12.0):
    """3-D points (reference camera frame, f32) of the pixels uv on the plane n . X = d"""
    n = np.asarray(n, np.float64)
    rays = (np.linalg.inv(K) @ np.c_[uv, np.ones(len(uv))].T).T
    depth = d / (rays @ n)
    return (rays * depth[:, None]).astype(np.float32)
