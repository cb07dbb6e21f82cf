"""Container known answers of the reference for the raw sink (R8/R9 of SURVEY 8(a)),
checked for the product's C-ABI sink (adder_raw_*), the oracle's C sink and the
independent numpy codec:
  encoder.rs:401-448 `raw3`: 1x1x3 plane, one event -> 59 bytes = 37 + 11 + 11
  decoder.rs:414-452: header sizes 25 / 29 / 33 for codec v0 / v1 / v2
  tests/integration_tests.rs:200-245: file lengths 36 (v0 closed) / 40 (v1 closed) / 33 (v2 unclosed)
and every sample `.adder` file must survive read -> re-write unchanged.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
import adder_stream_np as S
import adder_amd as A

SINKS = {
    "product": (A.raw_header, A.raw_events, A.raw_eof),
    "oracle": (O.raw_header, O.raw_events, O.raw_eof),
}


@pytest.mark.parametrize("sink", sorted(SINKS))
def test_raw3_is_59_bytes(sink):
    hdr, evs, eof = SINKS[sink]
    e = np.zeros(1, A.EVENT_DTYPE)
    e["x"], e["y"], e["c"], e["d"], e["t"] = 0, 0, 0, 0, 0  # encoder.rs:420-428
    blob = hdr(3, 1, 1, 3, 1, 1, 1, 0, 1, 0) + evs(e, 3) + eof()
    assert len(blob) == 59
    assert blob[37:48] == bytes([0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    assert blob[48:] == bytes([0xFF, 0xFF, 0xFF, 0xFF, 1, 0, 0, 0, 0, 0, 0])


@pytest.mark.parametrize("sink", sorted(SINKS))
def test_header_sizes_and_file_lengths(sink):
    hdr, evs, eof = SINKS[sink]
    sizes = [len(hdr(v, 50, 100, 1, 53000, 4000, 50000, 0, 1, 0)) for v in range(4)]
    assert sizes == [25, 29, 33, 37]
    assert len(hdr(0, 50, 100, 1, 53000, 4000, 50000) + eof()) == 36
    assert len(hdr(1, 50, 100, 1, 53000, 4000, 50000) + eof()) == 40
    assert len(hdr(2, 50, 100, 1, 53000, 4000, 50000)) == 33
    # EOF is the 11-byte form even for 1-channel streams (raw/stream.rs:79-92)
    assert len(eof()) == 11
    h = hdr(3, 200, 50, 1, 6113, 255, 6120, 0, 0, 0)
    assert h[:7] == b"adder\x03b" and h[23] == 9 and h[24] == 1
    assert S.parse_header(h)["tps"] == 6113


@pytest.mark.parametrize("name", ["sample_3_ordered.adder", "sample_3_unordered.adder", "bunny_v2_dt.adder",
                                  "bunny_v2_t.adder", "nyc_v1_1px.adder", "nyc_source_v2_2_1px.adder",
                                  "adder_info_test_sample.adder"])
def test_sample_files_round_trip(golden_dir, name):
    raw = open(os.path.join(golden_dir, name), "rb").read()
    meta, ev, closed = S.read_adder(raw)
    assert len(ev) > 0
    ch = meta["channels"]
    for sink in SINKS.values():
        hdr, evs, eof = sink
        blob = hdr(meta["version"], meta["width"], meta["height"], ch, meta["tps"], meta["ref_interval"],
                   meta["delta_t_max"], meta["source_camera"], meta["time_mode"], meta["adu_interval"])
        blob += evs(ev, ch)
        if closed:
            blob += eof()
        assert blob == raw[: len(blob)]
        assert len(raw) - len(blob) < meta["event_size"] + 11  # nothing but a truncated tail may remain
    assert S.write_adder(meta, ev, close=closed) == raw[: len(S.write_adder(meta, ev, close=closed))]


def test_event_wire_forms():
    e = np.zeros(2, A.EVENT_DTYPE)
    e["x"], e["y"], e["c"], e["d"], e["t"] = [0x0102, 7], [0x0304, 9], [0xFF, 0xFF], [5, 255], [0x0A0B0C0D, 1]
    assert A.raw_events(e[:1], 1) == bytes([1, 2, 3, 4, 5, 0x0A, 0x0B, 0x0C, 0x0D])
    e["c"] = [2, 0]
    assert A.raw_events(e[:1], 3) == bytes([1, 2, 3, 4, 1, 2, 5, 0x0A, 0x0B, 0x0C, 0x0D])
    assert A.raw_events(e, 3) == O.raw_events(e, 3) == S.write_adder(
        dict(version=0, width=1, height=1, tps=1, ref_interval=1, delta_t_max=1, channels=3), e, close=False)[25:]
