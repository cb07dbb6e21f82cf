"""Host-to-device copy rate of ONE 1080p frame (2 MB, page-locked) on this box, back to back on one stream -- what bounds the
per-frame ring when everything else overlaps."""
import torch, time
for nbytes in (1920 * 1080, 4 * 1920 * 1080, 64 << 20):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(20):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    n = 200 if nbytes < (32 << 20) else 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        d.copy_(h, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"H2D {nbytes} bytes: {us:.1f} us per copy = {nbytes / us / 1e3:.1f} GB/s")
    h2 = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    e0.record()
    for _ in range(n):
        h2.copy_(d, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"D2H {nbytes} bytes: {us:.1f} us per copy = {nbytes / us / 1e3:.1f} GB/s")

# the same 2 MB copy when it is NOT queued behind another one (the ring's case: one upload per submit call, a period apart):
# start -> end on the device's clock, with the host pacing the submissions
nbytes = 1920 * 1080
h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for gap_us in (0, 30, 60, 100):
    durs, spacing = [], []
    evs = []
    with torch.cuda.stream(s):
        for i in range(120):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            d.copy_(h, non_blocking=True)
            b.record(s)
            evs.append((a, b))
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < gap_us:
                pass
    torch.cuda.synchronize()
    for i in range(20, 120):
        durs.append(evs[i][0].elapsed_time(evs[i][1]) * 1e3)
        spacing.append(evs[i - 1][0].elapsed_time(evs[i][0]) * 1e3)
    durs.sort(); spacing.sort()
    print(f"host gap {gap_us:3d} us: copy start->end median {durs[len(durs)//2]:.1f} us, start spacing median {spacing[len(spacing)//2]:.1f} us")
