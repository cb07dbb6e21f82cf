"""The default-quality per-frame ring leg as the FIRST context of a process (what a transcoder process sees), twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import torch
import adder_amd as A
import bench_legs as B
print(os.environ.get("GPU_MAX_HW_QUEUES"), "first", B.end_to_end_default_quality(torch, A, 1920, 1080)["us_per_frame_sustained"],
      "second", B.end_to_end_default_quality(torch, A, 1920, 1080)["us_per_frame_sustained"])
