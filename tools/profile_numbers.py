"""The numbers DESIGN.md / profiles/README.md quote, straight from the committed profiles/r04_* files (kernel durations
per launch, VALU / SALU per wave-frame, HBM bytes per launch): python tools/profile_numbers.py"""
import csv, json, os
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def stats(tag):
    d = {}
    for r in list(csv.reader(open(os.path.join(P, f"r04_{tag}_kernel_stats.csv"))))[1:]:
        if not any(x in r[0] for x in ("synth", "rocclr", "native")):
            d[r[0].split("(")[0].replace("void adder::", "").replace("adder::", "")] = round(float(r[3]) / 1000, 1)
    return d


def pmc(tag, frames):
    d = {}
    for r in csv.reader(open(os.path.join(P, f"r04_pmc_{tag}.csv"))):
        if len(r) >= 4 and r[0] != "kernel":
            d.setdefault(r[0].replace("adder::", ""), {})[r[1]] = float(r[3])
    out = {}
    for k, v in d.items():
        if any(x in k for x in ("lr_kernel", "rr_kernel", "cr_kernel", "cb_kernel", "frame_kernel", "lean_kernel", "lean1", "expand")):
            w = v["SQ_WAVES"]
            per = frames if "expand" not in k else 1.0
            out[k] = {"VALU": round(v["SQ_INSTS_VALU"] / w / per, 1), "SALU": round(v["SQ_INSTS_SALU"] / w / per, 1),
                      "fetch_MB_x2": round(2 * v["FETCH_SIZE"] * 1024 / 1e6, 1), "write_MB": round(v["WRITE_SIZE"] * 1024 / 1e6, 1)}
    return out


for tag in ("bench_eager_serial", "events_output_eager", "default_mode_eager", "default_mode_delta_eager", "cr_dtm7650_abs_eager",
            "cb_dtm7650_abs_eager", "normal_dtm255_delta_eager", "normal_dtm7650_abs_eager", "one_frame_per_launch_eager"):
    print(tag, stats(tag))
for tag, fr in (("default", 160 / 3), ("events_output", 160 / 3), ("lean_step_no_runs", 160 / 3), ("default_mode_dtm7650_delta", 64),
                ("default_mode_dtm7650_abs", 64), ("cr_dtm7650_delta", 64), ("cr_dtm7650_abs", 64), ("cb_dtm7650_delta", 64),
                ("cb_dtm7650_abs", 64), ("normal_dtm255_delta", 64), ("one_frame_per_launch", 1)):
    print(tag, json.dumps(pmc(tag, fr)))
