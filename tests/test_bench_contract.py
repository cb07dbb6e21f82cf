"""bench.py's contract as the driver uses it, on the GPU box: the one JSON line, its required keys, and the N > 1 path
(two ranks sharing the box's one GPU over gloo -- RCCL refuses two ranks per device -- with the chunk-pipelined gather
inside the timed step; bench.py itself asserts merged_total == total_events on rank 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line_has_roofline_secondary_and_cpu_baseline():
    d = _run(["--frames", "130", "--steps", "3", "--warmup", "1", "--cpu-seconds", "4", "--secondary-ms", "15"],
             {"ADDER_BENCH_PLAN_STEPS": "3"})
    for k in REQUIRED + ["cpu_baseline", "secondary"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["workload"].startswith("1920x1080")
    assert 0.0 < d["roofline"]["frac"] < 1.0 and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["gpu_events_match_bit_exact"] is True
    assert len(d["secondary"]) >= 8 and all("error" not in leg for leg in d["secondary"]), d["secondary"]


@pytest.mark.parametrize("gather", ["torch", "layout", "records"])
def test_bench_two_ranks_share_the_device_and_gather_chunk_by_chunk(gather):
    d = _run(["--gpus", "2", "--frames", "150", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end",
              "--no-secondary", "--skip-roofline", "--gather", gather],
             {"ADDER_BENCH_SHARE_DEVICE": "1", "ADDER_BENCH_PLAN_STEPS": "2", "ADDER_BENCH_GATHER_CHUNK": "64"})
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["events_per_pixel_frame"] > 0.2
    if gather == "records":  # root (rank 0) takes more rows than its peer: sharding.gather_peer_share
        assert 540 < d["config"]["rows_per_gpu"] < 1080 and d["config"]["row_bands"][1][1] == 1080
    else:
        assert d["config"]["rows_per_gpu"] == 540
    if gather == "records":  # the peers' records + tables are well under half of their events' bytes
        w = d["records_over_the_wire"]
        assert 0 < w["bytes_per_step_all_peers"] < 0.5 * w["events_bytes_per_step_all_peers"]
