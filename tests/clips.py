"""Seeded input clips that exercise every branch of the per-pixel step (test helper)."""
import numpy as np


def make_clip(kind, frames, H, W, C, seed):
    rng = np.random.default_rng(seed)
    shape = (frames, H, W, C)
    if kind == "noise":
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == "static":
        return np.broadcast_to(rng.integers(0, 256, (1, H, W, C), dtype=np.uint8), shape).copy()
    if kind == "dark":  # lots of zeros and tiny values: d = 128 paths
        return rng.choice(np.array([0, 0, 0, 1, 2, 3, 255], np.uint8), shape)
    if kind == "jitter":  # slow drift within / across c_thresh
        base = rng.integers(0, 256, (1, H, W, C)).astype(np.int64)
        walk = np.cumsum(rng.integers(-2, 3, shape), axis=0)
        return np.clip(base + walk, 0, 255).astype(np.uint8)
    if kind == "runs":  # piecewise-constant runs of random length per pixel
        out = np.zeros(shape, np.uint8)
        cur = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        for k in range(frames):
            change = rng.random((H, W, C)) < 0.08
            newv = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
            cur = np.where(change, newv, cur)
            out[k] = cur
        return out
    if kind == "steps":  # long constant runs with rare big steps, incl. to/from zero
        out = np.zeros(shape, np.uint8)
        cur = rng.choice(np.array([0, 1, 3, 17, 100, 128, 200, 255], np.uint8), (H, W, C))
        for k in range(frames):
            change = rng.random((H, W, C)) < 0.015
            newv = rng.choice(np.array([0, 1, 2, 5, 64, 127, 129, 254, 255], np.uint8), (H, W, C))
            cur = np.where(change, newv, cur)
            out[k] = cur
        return out
    raise ValueError(kind)
