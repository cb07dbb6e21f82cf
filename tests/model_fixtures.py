"""Loader of tests/golden/model_fixtures.npz (tests/golden/make_model_fixtures.py: known answers of an independent
second restatement for the mode combinations no reference artefact pins)."""
import os

import numpy as np

EVENT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("d", "u1"), ("pad", "<u2"), ("t", "<u4")])


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "model_fixtures.npz"))
    ev = np.zeros(len(z["ev_x"]), EVENT_DTYPE)
    for k in ("x", "y", "c", "d", "t"):
        ev[k] = z["ev_" + k]
    cases = []
    for p in z["params"]:
        W, H, C, T = int(p["W"]), int(p["H"]), int(p["C"]), int(p["T"])
        fp, cp, ep = int(p["frame_pos"]), int(p["count_pos"]), int(p["event_pos"])
        counts = z["counts"][cp:cp + T].astype(np.int64)
        cases.append({
            "collapse": bool(p["collapse"]), "abs_t": bool(p["abs_t"]), "ref": int(p["ref"]), "dtm": int(p["dtm"]),
            "c_max": int(p["c_max"]), "vel": int(p["vel"]), "c_start": int(p["c_start"]), "ctr_start": int(p["ctr_start"]),
            "frames": z["frames"][fp:fp + T * H * W * C].reshape(T, H, W, C), "counts": counts,
            "events": ev[ep:ep + int(counts.sum())]})
    return cases
