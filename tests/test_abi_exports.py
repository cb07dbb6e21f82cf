"""CPU-side checks of the boundary: the shared library loads, exports every symbol that
include/*.h declare, refuses to run without a GPU (no silent fallback), and its
raw sink reproduces the reference's container known answers."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in ("adder_hip.h", "adder_framer.h", "adder_compressed.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names |= set(re.findall(r"\b(adder_(?:hip|raw|framer|compressed)_\w+)\s*\(", hdr))
    return sorted(names)


def test_gather_library_exports_every_declared_symbol():
    """include/adder_gather.h <-> libadder_rccl.so (the multi-GPU gather a Rust host binds)."""
    import ctypes
    import adder_amd
    from adder_amd import gather
    hdr = open(os.path.join(ROOT, "include", "adder_gather.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(adder_(?:gather|host_image)_\w+)\s*\(", hdr))
    assert len(names) >= 7 and names == set(gather.SYMBOLS)
    adder_amd.load()
    L = ctypes.CDLL(gather.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_library_exports_every_declared_symbol():
    import adder_amd
    L = adder_amd.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(adder_amd._native.SYMBOLS), "binding table out of sync with the header"


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import adder_amd
    with pytest.raises(adder_amd.AdderHipError) as ei:
        adder_amd.HipVideo(8, 8, 1)
    assert ei.value.code == -3


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "adder-codec-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f in ("adder_pixel.hpp",), (dirpath, f)


def test_records_wire_layout_is_consistent():
    """adder_hip_records_wire_bytes / _sections (host arithmetic only): six sections at 256-byte multiples, in order, each
    large enough for what adder_hip_records_to_wire copies into it."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
    from adder_amd import _native as N
    L = N.load()
    for nf, nseg, rb, nrec in ((1, 16, 8, 0), (64, 2032, 8, 123457), (37, 16208, 12, 5_000_001), (64, 48, 12, 1)):
        sec = (C.c_size_t * 6)()
        L.adder_hip_records_wire_sections(nf, nseg, rb, sec)
        sec = [int(x) for x in sec]
        total = int(L.adder_hip_records_wire_bytes(nf, nseg, rb, nrec))
        need = [(nf + 1) * 8, nf * 8, nf * nseg * 4, nf * nseg * 4, nf * nseg * 4, nrec * rb]
        ends = sec[1:] + [total]
        assert sec[0] == 0 and all(s % 256 == 0 for s in sec) and total % 256 == 0
        assert all(e - s >= n for s, e, n in zip(sec, ends, need)), (sec, total, need)
