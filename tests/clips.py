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
    if kind == "corners":  # bright / dark rectangles drifting over a textured background: FAST corners that move
        out = np.zeros(shape, np.uint8)
        bg = rng.integers(90, 110, (H, W, C)).astype(np.int64)
        nrect = max(2, (H * W) // 400)
        rx = rng.integers(0, W, nrect).astype(np.float64)
        ry = rng.integers(0, H, nrect).astype(np.float64)
        rw = rng.integers(5, max(6, W // 3), nrect)
        rh = rng.integers(5, max(6, H // 3), nrect)
        vx = rng.uniform(-0.7, 0.7, nrect)
        vy = rng.uniform(-0.7, 0.7, nrect)
        val = rng.choice(np.array([5, 20, 200, 250]), nrect)
        for k in range(frames):
            img = bg + rng.integers(-2, 3, (H, W, C))
            for r in range(nrect):
                x0, y0 = int(rx[r] + vx[r] * k) % W, int(ry[r] + vy[r] * k) % H
                img[y0:min(H, y0 + rh[r]), x0:min(W, x0 + rw[r])] = val[r]
            out[k] = np.clip(img, 0, 255).astype(np.uint8)
        return out
    raise ValueError(kind)


def quiet_group_clip(frames, H, W, rng, jitter, C=1):
    """Content for the quiet-GROUP paths of the blocked kernels (16 frames of a quiet unit decided at once): static rows
    of every kind -- black, dark (tiny sums: several firings per group), mid, bright -- with jitter inside the contrast
    band, and ONE disturbance per 64-frame block that walks through every position of a 16-frame group.
    Returns (clip [frames][H][W][C] u8, the frames at which a row changes)."""
    base = rng.integers(0, 256, (1, H, W, C))
    base[0, 0] = 0
    base[0, 1, : W // 2] = 1
    base[0, 1, W // 2:] = 3
    base[0, 2] = rng.integers(2, 9, (W, C))
    clip = np.repeat(base, frames, axis=0).astype(np.int64)
    if jitter:
        clip[:, 3:] += rng.integers(-jitter, jitter + 1, (frames, H - 3, W, C))
        clip[:, 1] += rng.integers(0, 2, (frames, W, C))          # 0 / 1 flicker on near-black pixels: black roots that wake up
    clip = np.clip(clip, 0, 255)
    breaks = []
    for blk in range(frames // 64):
        pos = 64 * blk + 16 * (blk % 4) + (blk * 5 + 3) % 16      # every position of a group over 16 blocks
        row = 3 + blk % (H - 3)
        clip[pos:, row] = 255 - clip[pos:, row]                   # a flush: the row restarts (and pops again later)
        breaks.append(pos)
    return clip.astype(np.uint8), breaks
