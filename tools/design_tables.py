"""Markdown tables for DESIGN.md section 5 from a bench.py JSON line: python tools/design_tables.py gpurun_out/<run>/bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"headline: {d['value']} Mpixels/s, {d['ms_per_step']} ms per step, e = {d['events_per_pixel_frame']}, records = {d['records_per_unit_frame']}")
print(f"roofline: chunk {r['chunk_us']} us = frame kernel {r['frame_kernel_launch_us']} (per {r['frames_per_launch']} frames) + post {r['scan_offsets_expand_us']}; "
      f"achieved {r['achieved']} GB/s, frac {r['frac']}, at record bytes {r['frac_at_record_bytes']}, traffic {r['traffic']} ({r['traffic_over_algorithmic']}x, "
      f"{r['traffic_over_record_bytes']}x at record bytes); one frame per launch: {d['roofline_one_frame_per_launch']['launch_avg_us']} us, frac {d['roofline_one_frame_per_launch']['frac']}")
print("output_check:", d.get("output_check"))
print()
print("| leg | K1 | µs per frame | Mpixels/s | e | frac (wall) | frame kernel µs / frame | scan + offsets + expansion µs / frame | kernels frac | frame kernel frac | expansion frac |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for l in d.get("secondary", []):
    if "error" in l:
        print(f"| {l['workload']} | error: {l['error'][:60]} |")
        continue
    print(f"| {l['workload']} | {l.get('frame_kernel', '').replace('adder_', '').replace('_kernel', '')} | {l['us_per_frame']} | {l['value']:,.0f} | {l['events_per_unit_frame']} | {l['frac']} | {l.get('frame_kernel_us_per_frame')} | "
          f"{l.get('scan_offsets_expand_us_per_frame')} | {l.get('kernels_frac')} | {l.get('frame_kernel_frac')} | {l.get('expansion_frac')} |")
print()
for k, v in d.get("end_to_end", {}).items():
    print(k, {q: v.get(q) for q in ("value", "unit", "us_per_frame_sustained", "mpixels_per_s", "GBs", "error") if v.get(q) is not None})
cb = d.get("cpu_baseline", {})
print("cpu_baseline:", {q: cb.get(q) for q in ("value", "unit", "cores", "kind", "sample", "cpus_usable", "gpu_events_match_bit_exact")})
