"""Independent numpy reader/writer of the raw `.adder` container (test helper).

Follows adder-codec-core/src/codec/header.rs:14-25, encoder.rs:170-229 (header +
V1/V2/V3 extensions), raw/stream.rs:101-120 (9/11-byte big-endian events) and
raw/stream.rs:79-92 (11-byte EOF).  Written separately from oracle/adder_oracle.c
so the two can be diffed against each other and against the golden files.
"""
import struct

import numpy as np

EVENT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("d", "u1"), ("pad", "<u2"), ("t", "<u4")]
)
WIRE9 = np.dtype([("x", ">u2"), ("y", ">u2"), ("d", "u1"), ("t", ">u4")])
WIRE11 = np.dtype([("x", ">u2"), ("y", ">u2"), ("some", "u1"), ("c", "u1"), ("d", "u1"), ("t", ">u4")])
EOF = bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x01, 0, 0, 0, 0, 0, 0])


def parse_header(buf):
    magic, version, endian, w, h, tps, ref, dtm, esize, ch = struct.unpack(">5sBBHHIIIBB", buf[:25])
    assert magic == b"adder" and endian == ord("b")
    meta = dict(version=version, width=w, height=h, tps=tps, ref_interval=ref, delta_t_max=dtm,
                event_size=esize, channels=ch, source_camera=0, time_mode=0, adu_interval=0)
    off = 25
    if version >= 1:
        (meta["source_camera"],) = struct.unpack(">I", buf[off:off + 4]); off += 4
    if version >= 2:
        (meta["time_mode"],) = struct.unpack(">I", buf[off:off + 4]); off += 4
    if version >= 3:
        (meta["adu_interval"],) = struct.unpack(">I", buf[off:off + 4]); off += 4
    meta["header_size"] = off
    return meta


def build_header(meta):
    b = struct.pack(">5sBBHHIIIBB", b"adder", meta["version"], ord("b"), meta["width"], meta["height"],
                    meta["tps"], meta["ref_interval"], meta["delta_t_max"],
                    9 if meta["channels"] == 1 else 11, meta["channels"])
    if meta["version"] >= 1:
        b += struct.pack(">I", meta["source_camera"])
    if meta["version"] >= 2:
        b += struct.pack(">I", meta["time_mode"])
    if meta["version"] >= 3:
        b += struct.pack(">I", meta["adu_interval"])
    return b


def read_adder(buf):
    """-> (meta, events[EVENT_DTYPE], closed: bool).  Stops at the EOF event."""
    meta = parse_header(buf)
    body = buf[meta["header_size"]:]
    es = meta["event_size"]
    closed = False
    if meta["channels"] == 1:
        # EOF is 11 bytes even here (raw/stream.rs:79-92); it starts with ff ff ff ff
        n = len(body) // es
        arr = np.frombuffer(body[: n * es], dtype=WIRE9)
        eof = np.nonzero((arr["x"] == 0xFFFF) & (arr["y"] == 0xFFFF))[0]
        if len(eof):
            n = int(eof[0])
            closed = body[n * es: n * es + 11] == EOF
        arr = arr[:n]
        ev = np.zeros(n, EVENT_DTYPE)
        ev["x"], ev["y"], ev["c"], ev["d"], ev["t"] = arr["x"], arr["y"], 0xFF, arr["d"], arr["t"]
    else:
        n = len(body) // es
        arr = np.frombuffer(body[: n * es], dtype=WIRE11)
        eof = np.nonzero((arr["x"] == 0xFFFF) & (arr["y"] == 0xFFFF))[0]
        if len(eof):
            n = int(eof[0])
            closed = True
        arr = arr[:n]
        assert np.all(arr["some"] == 1)
        ev = np.zeros(n, EVENT_DTYPE)
        ev["x"], ev["y"], ev["c"], ev["d"], ev["t"] = arr["x"], arr["y"], arr["c"], arr["d"], arr["t"]
    return meta, ev, closed


def write_adder(meta, events, close=True):
    out = build_header(meta)
    if meta["channels"] == 1:
        w = np.zeros(len(events), WIRE9)
        w["x"], w["y"], w["d"], w["t"] = events["x"], events["y"], events["d"], events["t"]
    else:
        w = np.zeros(len(events), WIRE11)
        w["x"], w["y"], w["some"], w["c"], w["d"], w["t"] = (
            events["x"], events["y"], 1, events["c"], events["d"], events["t"])
    out += w.tobytes()
    if close:
        out += EOF
    return out
