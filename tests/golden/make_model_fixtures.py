#!/usr/bin/env python3
"""Generates tests/golden/model_fixtures.npz: randomised known answers for the mode combinations NO reference test or
fixture pins: Collapse with delta_t_max = ref_time (the headline), Collapse + AbsoluteT with
delta_t_max 7650 (the reference's defaults), crf 3 / 6 / 9 numbers, RGB, Normal + AbsoluteT.

The expected events come from the model below -- a SECOND, independent restatement of one pixel-step, written from the
behavioural spec of SURVEY.md Appendix A alone (per-pixel Python objects, node lists, numpy float32 scalars), not from
oracle/adder_oracle.c and not from the device header.  It is the only available substitute for reference vectors in these
modes: tests/test_oracle_golden.py holds the C oracle to it, tests/test_gpu_parity.py the HIP path.  On a disagreement
re-read event_pixel_tree.rs:213-287,317-413 -- neither side is to be trusted blindly.

usage: python tests/golden/make_model_fixtures.py        (deterministic: seeds are fixed; ~1 minute)
"""
import os
import sys

import numpy as np

F = np.float32
D_MAX, D_ZERO, D_EMPTY = 127, 128, 255
EPS = F(1.1920929e-7)


def trunc_u32(x):
    """Rust `f32 as u32`: truncate toward zero, saturate, NaN -> 0."""
    x = float(x)
    if not x > 0.0:
        return 0
    if x >= 4294967296.0:
        return 0xFFFFFFFF
    return int(x)


def flog2(x):
    """floor(log2(trunc(x))) clamped to 127; 128 if x < 1."""
    t = int(float(x))
    if t < 1:
        return D_ZERO
    return min(t.bit_length() - 1, D_MAX)


def pow2(d):
    return F(0.0) if d == D_ZERO else F(2.0) ** F(d)


class Node:
    __slots__ = ("d", "integ", "dt", "best")

    def __init__(self, intensity):
        self.d = flog2(intensity)
        self.integ = F(0.0)
        self.dt = F(0.0)
        self.best = None  # (d, delta_t)


class Pixel:
    """state per pixel-channel of Appendix A"""

    def __init__(self, cfg):
        self.cfg = cfg
        self.nodes = [Node(F(1.0))]  # PixelArena::new(1.0): d = 0
        self.base_val = 0
        self.c_thresh = cfg["c_start"]
        self.c_ctr = cfg["ctr_start"]
        self.popped = False
        self.need_pop = False
        self.last_fired = F(0.0)
        self.running_t = F(0.0)
        self.out = []

    # ---- emit (R6) ----
    def emit_value(self, ev):
        d, delta = ev
        if self.cfg["abs_t"]:
            delta = F(delta + self.last_fired)
            self.last_fired = delta
            ref = self.cfg["ref"]
            lf = trunc_u32(self.last_fired)
            if lf % ref != 0:
                lf = (lf // ref + 1) * ref
            self.last_fired = F(lf)
        return (d, trunc_u32(delta))

    # ---- step ----
    def step(self, v, T):
        self.out = []
        inten = F(v)
        if self.need_pop:
            self.pop_top(inten)
        lo = max(self.base_val - self.c_thresh, 0)
        hi = min(self.base_val + self.c_thresh, 255)
        if v < lo or v > hi:
            self.flush(inten)
            self.base_val = v
        self.integrate(inten, T)
        if self.need_pop:
            self.pop_top(inten)
        return self.out

    def integrate(self, inten, T):
        cfg = self.cfg
        tail = self.nodes[-1]
        if tail.dt == 0 and tail.integ == 0:
            tail.d = flog2(inten)
        self.running_t = F(self.running_t + T)
        idx = 0
        while True:
            n = self.nodes[idx]
            s = F(n.integ + inten)
            if s >= pow2(n.d):
                nd = flog2(s)
                if nd == D_ZERO or n.d == D_ZERO or inten < EPS:
                    prop = F(1.0)
                else:
                    prop = F(F(pow2(nd) - n.integ) / inten)
                n.d = nd
                n.best = (nd, F(n.dt + F(T * prop)))
                if nd < D_MAX:
                    n.integ = s
                    n.dt = F(n.dt + T)
                    k = nd + 1
                    while k < 128 and not (2 ** k > int(float(n.integ))):
                        k += 1
                    n.d = k if k < 128 else D_ZERO
                del self.nodes[idx + 1:]
                self.nodes.append(Node(inten))
                break
            n.integ = s
            n.dt = F(n.dt + T)
            if self.popped and cfg["collapse"]:
                break
            if idx + 1 >= len(self.nodes):
                break
            idx += 1
        root = self.nodes[0]
        self.need_pop = root.d == D_MAX or (root.dt >= F(cfg["dtm"]) and not self.popped)
        if self.c_thresh < cfg["c_max"]:
            if self.c_ctr >= cfg["vel"] - 1:
                self.c_thresh = min(self.c_thresh + 1, 255)
                self.c_ctr = 0
            else:
                self.c_ctr = min(self.c_ctr + (trunc_u32(T) // cfg["ref"]) % 256, 255)

    def flush(self, inten):
        local = []
        for n in self.nodes:
            if n.best is not None:
                local.append(self.emit_value(n.best))
            elif n.dt > 0 and n.integ == 0:
                local.append(self.emit_value((D_ZERO, n.dt)))
                n.dt = F(0.0)
        if self.popped and self.cfg["collapse"] and local:
            self.out.append(local[0])
            self.last_fired = self.running_t
            self.out.append((D_EMPTY, trunc_u32(self.running_t)))
            self.nodes = [Node(inten)]
        else:
            self.out.extend(local)
            last = self.nodes[-1]
            self.nodes = [last]
        self.need_pop = False
        self.popped = False

    def pop_top(self, inten):
        self.need_pop = False
        root = self.nodes[0]
        ev = None
        if root.best is None:
            if root.integ == 0 and root.dt > 0:
                ev = (D_ZERO, root.dt)
                root.dt = F(0.0)
                root.d = flog2(inten)
            else:
                root.best = (D_ZERO if root.integ < 1 else flog2(root.integ), root.dt)
                self.nodes = [root, Node(inten)]
        if root.best is not None:
            ev = root.best
            self.nodes.pop(0)
        self.popped = True
        self.out.append(self.emit_value(ev))


def run_case(cfg, frames):
    """frames [T][H][W][C] u8 -> list over frames of arrays (x, y, c, d, t) in raster order."""
    T, H, W, C = frames.shape
    px = [[[Pixel(cfg) for _ in range(C)] for _ in range(W)] for _ in range(H)]
    per_frame = []
    span = F(cfg["ref"])
    for k in range(T):
        ev = []
        for y in range(H):
            for x in range(W):
                for c in range(C):
                    for d, t in px[y][x][c].step(int(frames[k, y, x, c]), span):
                        ev.append((x, y, 0xFF if C == 1 else c, d, t))
        per_frame.append(ev)
    return per_frame


# ---- content ----
def make_clip(rng, kind, T, H, W, C):
    if kind == "noise":
        return rng.integers(0, 256, (T, H, W, C), dtype=np.uint8)
    if kind == "jitter":
        base = rng.integers(0, 256, (1, H, W, C))
        amp = rng.choice([1, 2, 4, 9])
        clip = np.clip(base + rng.integers(-amp, amp + 1, (T, H, W, C)), 0, 255)
        return clip.astype(np.uint8)
    if kind == "runs":
        clip = np.zeros((T, H, W, C), np.uint8)
        for y in range(H):
            for x in range(W):
                for c in range(C):
                    k = 0
                    while k < T:
                        n = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 40, 70]))
                        clip[k:k + n, y, x, c] = rng.choice([0, 1, 2, 3, 7, 16, 100, 128, 200, 254, 255])
                        k += n
        return clip
    if kind == "dark":
        return rng.choice(np.array([0, 0, 0, 0, 1, 2, 3], np.uint8), (T, H, W, C))
    if kind == "static":
        clip = np.repeat(rng.integers(0, 256, (1, H, W, C), dtype=np.uint8), T, axis=0)
        clip[T - T // 5:] = 255 - clip[T - T // 5:]
        return clip
    raise ValueError(kind)


CRF = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}


def main():
    rng = np.random.default_rng(0xADDE5EED)
    cases = []
    # (collapse, abs_t, dtm / ref, crf, channels): the unpinned combinations, weighted towards the two that matter most
    combos = ([(1, 0, 1, 0, 1)] * 5 + [(1, 1, 30, 0, 1)] * 5 + [(1, 1, 30, 3, 1)] * 4 + [(1, 0, 30, 3, 1)] * 2 +
              [(1, 1, 1, 0, 1)] * 2 + [(1, 1, 30, 6, 1), (1, 1, 30, 9, 1), (1, 0, 1, 6, 1), (1, 0, 1, 9, 3)] +
              [(1, 1, 30, 3, 3)] * 2 + [(1, 0, 1, 0, 3)] + [(0, 1, 4, 0, 1), (0, 1, 30, 3, 3), (1, 1, 4, 3, 1)])
    kinds = ["runs", "jitter", "noise", "dark", "static"]
    i = 0
    while len(cases) < 216:
        collapse, abs_t, ratio, crf, C = combos[i % len(combos)]
        kind = kinds[(i // len(combos) + i) % len(kinds)]
        ref = int(rng.choice([255, 255, 255, 1000, 20]))
        W, H = int(rng.integers(1, 6)), int(rng.integers(1, 5))
        T = int(rng.integers(36, 90))
        base, cmax, vel = CRF[crf]
        default_pixels = crf == 3 and i % 7 == 0  # construction defaults: c_thresh 10, counter 1 (SURVEY 8(a) note 6)
        cfg = {"collapse": bool(collapse), "abs_t": bool(abs_t), "ref": ref, "dtm": ref * ratio, "c_max": cmax, "vel": vel,
               "c_start": 10 if default_pixels else base, "ctr_start": 1 if default_pixels else 0}
        clip = make_clip(rng, kind, T, H, W, C)
        cases.append((cfg, kind, clip, run_case(cfg, clip)))
        i += 1
    # pack
    params = np.zeros(len(cases), dtype=[("collapse", "u1"), ("abs_t", "u1"), ("ref", "<u4"), ("dtm", "<u4"), ("c_max", "u1"),
                                         ("vel", "u1"), ("c_start", "u1"), ("ctr_start", "u1"), ("W", "<u2"), ("H", "<u2"),
                                         ("C", "u1"), ("T", "<u2"), ("frame_pos", "<u8"), ("count_pos", "<u8"), ("event_pos", "<u8")])
    frames, counts, events = [], [], []
    fpos = cpos = epos = 0
    for k, (cfg, kind, clip, per_frame) in enumerate(cases):
        T, H, W, C = clip.shape
        params[k] = (cfg["collapse"], cfg["abs_t"], cfg["ref"], cfg["dtm"], cfg["c_max"], cfg["vel"], cfg["c_start"],
                     cfg["ctr_start"], W, H, C, T, fpos, cpos, epos)
        frames.append(clip.reshape(-1))
        fpos += clip.size
        for ev in per_frame:
            counts.append(len(ev))
            events.extend(ev)
            epos += len(ev)
        cpos += T
    ev = np.array(events, dtype=np.int64).reshape(-1, 5)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_fixtures.npz")
    np.savez_compressed(out, params=params, frames=np.concatenate(frames), counts=np.array(counts, np.uint32),
                        ev_x=ev[:, 0].astype(np.uint16), ev_y=ev[:, 1].astype(np.uint16), ev_c=ev[:, 2].astype(np.uint8),
                        ev_d=ev[:, 3].astype(np.uint8), ev_t=ev[:, 4].astype(np.uint32))
    print(f"{len(cases)} cases, {len(ev)} events, {fpos} input bytes -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    sys.exit(main())
