"""PCIe-inclusive throughput of the host-buffer entry point (not a test): host frames in,
host events out, through adder_hip_integrate_batch (synchronous, unpipelined)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 64
d = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
L = A.load()
import ctypes as C
fp = L.adder_hip_alloc_pinned(T * W * H)
frames = np.frombuffer((C.c_uint8 * (T * W * H)).from_address(fp), dtype=np.uint8).reshape(T, H, W, 1)
frames[...] = d.cpu().numpy().reshape(T, H, W, 1)
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
best = 1e9
for it in range(4):
    hv.reset()
    t0 = time.perf_counter()
    ev, offs = hv.integrate_batch(frames, out_cap=int(W * H * T * 0.5))
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"frames": T, "events": int(len(ev)), "seconds": round(best, 4),
                  "Mpx_per_s_pcie_inclusive": round(W * H * T / best / 1e6, 1)}))
# breakdown: the C call alone (no numpy copy of the result)
import ctypes as C
n = C.c_size_t(0); offs = np.zeros(T + 1, np.uint64)
cap = int(W * H * T * 0.5); out = hv._host_out(cap)
for it in range(3):
    hv.reset(); t0 = time.perf_counter()
    rc = hv.L.adder_hip_integrate_batch(hv.h, frames.ctypes.data, T, W * H, W, 255.0, out.ctypes.data, cap, C.byref(n), offs.ctypes.data)
    dt = time.perf_counter() - t0
    print("C call only:", rc, round(dt, 4), "s ->", round(W * H * T / dt / 1e6, 1), "Mpx/s; kernel ms", hv.last_batch_ms())

# ---- device-side raw sink (9-byte records), unpipelined: one C call
nb, ne = C.c_size_t(0), C.c_size_t(0)
cap_b = cap * 9
for it in range(3):
    hv.reset(); t0 = time.perf_counter()
    rc = hv.L.adder_hip_integrate_batch_raw(hv.h, frames.ctypes.data, T, W * H, W, 255.0, out.ctypes.data, cap_b,
                                            C.byref(nb), C.byref(ne), offs.ctypes.data)
    dt = time.perf_counter() - t0
    print("batch_raw C call:", rc, round(dt, 4), "s ->", round(W * H * T / dt / 1e6, 1), "Mpx/s;", nb.value, "bytes")

# ---- pipelined submit/collect, 4 clips of T frames back to back (state carries over)
K = 6
p, po = C.c_void_p(), C.c_void_p()
for it in range(3):
    hv.reset(); t0 = time.perf_counter(); tot = 0
    hv.L.adder_hip_stream_submit(hv.h, frames.ctypes.data, T, W * H, W, 255.0, cap)
    for k in range(K):
        if k + 1 < K:
            rc = hv.L.adder_hip_stream_submit(hv.h, frames.ctypes.data, T, W * H, W, 255.0, cap)
            assert rc == 0, rc
        rc = hv.L.adder_hip_stream_collect(hv.h, C.byref(p), C.byref(nb), C.byref(ne), C.byref(po))
        assert rc == 0, rc
        tot += nb.value
    dt = time.perf_counter() - t0
    print("pipelined stream:", round(dt, 4), "s for", K * T, "frames ->", round(W * H * T * K / dt / 1e6, 1), "Mpx/s;",
          tot, "bytes,", round(tot / dt / 1e9, 1), "GB/s down")
