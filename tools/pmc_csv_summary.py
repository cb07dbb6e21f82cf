#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes written as csv (one *counter_collection.csv per pass).

usage: pmc_csv_summary.py <dir holding pmc_*/ sub-directories> [--traffic out.json] [--frames-per-launch N]

--frames-per-launch: the frames every frame-kernel launch of the profiled command covered (run it with a frame count
that is a multiple of the chunk, so that all launches are alike); written into the traffic JSON, bench.py scales by it.

--traffic writes the HBM bytes per launch of the frame kernel (FETCH_SIZE and WRITE_SIZE are KiB on
gfx950; FETCH_SIZE counts wide coalesced reads at 1/2 and is doubled, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: [0, 0.0])
    meta = {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                if "adder" not in name:
                    continue
                kernel = name.split("(")[0].replace("void ", "").replace(",", ";")
                cname = row.get("Counter_Name") or row.get("counter_name")
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                a = acc[(kernel, cname)]
                a[0] += 1
                a[1] += val
                meta[kernel] = row.get("Grid_Size") or row.get("grid_size")
    print("kernel,counter,dispatches,avg_per_dispatch")
    for (kernel, cname), (n, s) in sorted(acc.items()):
        print(f"{kernel},{cname},{n},{s / n:.1f}")
    if "--traffic" in sys.argv:
        out = sys.argv[sys.argv.index("--traffic") + 1]
        res = {"source": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of an eager bench.py run "
                         "(tools/profile_round.sh), averaged per kernel launch",
               "correction": "FETCH_SIZE doubled (gfx950 counts wide coalesced reads at 1/2, MI355X_MICROARCH.md HBM "
                             "section); WRITE_SIZE as reported; both are KiB",
               "kernels": {}}
        if "--frames-per-launch" in sys.argv:
            res["frames_per_launch"] = float(sys.argv[sys.argv.index("--frames-per-launch") + 1])
        for (k, c) in sorted(acc):
            if c != "FETCH_SIZE":
                continue
            fetch = acc[(k, "FETCH_SIZE")][1] / acc[(k, "FETCH_SIZE")][0]
            w = acc.get((k, "WRITE_SIZE"), [1, 0.0])
            write = w[1] / max(w[0], 1)
            res["kernels"][k] = {"launches": acc[(k, "FETCH_SIZE")][0],
                                 "fetch_size_kib_per_launch": round(fetch, 1),
                                 "write_size_kib_per_launch": round(write, 1),
                                 "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
