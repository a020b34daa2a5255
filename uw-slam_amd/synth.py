"""Deterministic synthetic frame pairs: band-limited noise textures re-rendered under a small
random SE(3) for a fronto-parallel plane at depth z.  Host-side test/bench input only.
"""
import numpy as np
from scipy import ndimage


def texture(w, h, seed, sigma=3.0):
    """Standard-normal field -> Gaussian blur (sigma px) -> min-max to [0,255] u8."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((h, w)).astype(np.float32)
    f = ndimage.gaussian_filter(f, sigma, mode="wrap")
    lo, hi = float(f.min()), float(f.max())
    return np.clip(np.rint((f - lo) * (255.0 / (hi - lo))), 0, 255).astype(np.uint8)


def rodrigues(rvec):
    rvec = np.asarray(rvec, np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def random_motion(rng, max_t=0.01, max_deg=0.5):
    """‖t‖ ~ U(0, max_t) m, rotation angle ~ U(0, max_deg)."""
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, max_deg))
    tdir = rng.standard_normal(3)
    tdir /= np.linalg.norm(tdir)
    return rodrigues(axis * ang), tdir * rng.uniform(0, max_t)


def render_pair(w, h, fx, fy, cx, cy, seed, z=1.0, max_t=0.01, max_deg=0.5, with_depth=False):
    """Returns (ref u8, tgt u8, depth u16 or None, R, t).  tgt(u') = ref(H^-1 u'), H = K (R + t n^T / z) K^-1."""
    rng = np.random.default_rng(seed + 7919)
    ref = texture(w, h, seed)
    R, t = random_motion(rng, max_t, max_deg)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    H = K @ (R + np.outer(t, [0, 0, 1.0]) / z) @ np.linalg.inv(K)
    Hi = np.linalg.inv(H)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    u = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
    v = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    tgt = ndimage.map_coordinates(ref.astype(np.float32), [v, u], order=1, mode="reflect")
    tgt = np.clip(np.rint(tgt), 0, 255).astype(np.uint8)
    depth = None
    if with_depth:
        # TUM-style u16 depth at the reference's 0.0002 m/unit scale (Tracker.cpp:1261), with a few invalid zeros
        depth = np.full((h, w), int(round(z / 0.0002)), np.uint16)
        holes = rng.random((h, w)) < 0.01
        depth[holes] = 0
    return ref, tgt, depth, R, t


def shifted_pair(w, h, seed, dx=2, dy=0):
    """Reference + the same texture rolled by integer pixels (known-answer cases)."""
    ref = texture(w, h, seed)
    return ref, np.roll(ref, (dy, dx), axis=(0, 1))
