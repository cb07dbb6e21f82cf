"""Pins oracle/compressed_oracle.py (the restatement of the reference's compressed sink) by
re-running the reference's OWN tests on it: the round-trip tests of
adder-codec-core/src/codec/compressed/stream.rs:455-947, the Weights unit tests of fenwick/mod.rs:116-155,
and the integration test over adder-codec-core/tests/samples/virat_small_gray.adder
(tests/integration_tests.rs:44-85).  The reference holds no compressed golden bytes, so these properties --
not a byte comparison with the Rust original -- are what pins it."""
import os
import struct

import numpy as np
import pytest

from oracle import compressed_oracle as CO


def _out(w, h, num_intervals, dt_ref=255):
    return CO.CompressedOutput(w, h, 1, tps=7650, ref_interval=dt_ref, delta_t_max=dt_ref * num_intervals,
                               adu_interval=num_intervals, write_header=False)


def _decode(data, w, h, num_intervals, dt_ref=255):
    return CO.decode(data, width=w, height=h, channels=1, ref_interval=dt_ref, adu_interval=num_intervals,
                     header_size=0)


def test_weights_unit_tests():  # fenwick/mod.rs:116-155
    w = CO.Weights(3)
    assert w.total == 4
    assert [w.range(None), w.range(0), w.range(1), w.range(2)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert [w.symbol(0), w.symbol(1), w.symbol(2), w.symbol(3)] == [None, 0, 1, 2]


def test_compress_empty():  # stream.rs:455-508
    start_t, dt_ref, n = 0, 255, 10
    co = _out(16, 32, n)
    counter, done = 0, False
    for y in range(30):
        for x in range(16):
            co.ingest_event(x, y, 0xFF, 7, min(280 + counter, start_t + dt_ref * n))
            if 280 + counter > start_t + dt_ref * n:
                break
            counter += 1
    assert len(co.close()) > 0


def test_compress_decompress_barely_full():  # stream.rs:510-609
    start_t, dt_ref, n = 0, 255, 10
    co = _out(16, 32, n)
    cand, inp = (7, 12), []
    counter = 0
    for y in range(30):
        for x in range(16):
            t = min(280 + counter, start_t + dt_ref * n)
            if (y, x) == cand:
                inp.append((x, y, 0xFF, 7, t))
            co.ingest_event(x, y, 0xFF, 7, t)
            if 280 + counter > start_t + dt_ref * n:
                break
            counter += 1
    co.ingest_event(0, 0, 0xFF, 7, start_t + dt_ref * n + 1)
    counter += 1
    output = co.close()
    assert 0 < len(output) < counter * 9
    events = _decode(output, 16, 30, n)
    got = [e for e in events[: counter - 1] if (e[1], e[0]) == cand]
    assert got == inp


def _check_candidate(inp, out, check_d=True):
    assert len(inp) >= len(out)
    for a, b in zip(inp, out):
        assert a[4] - 5 <= b[4] < a[4] + 5
        if check_d:
            assert a[3] == b[3]


def test_compress_decompress_several():  # stream.rs:611-701
    dt_ref, n = 255, 5
    co = _out(16, 32, n)
    cand, inp, counter = (7, 12), [], 0
    for _ in range(10):
        for y in range(30):
            for x in range(16):
                ev = (x, y, 0xFF, 7, 280 + counter)
                if (y, x) == cand:
                    inp.append(ev)
                co.ingest_event(*ev)
                counter += 1
    output = co.close()
    assert 0 < len(output) < counter * 9
    events = _decode(output, 16, 30, n)
    _check_candidate(inp, [e for e in events[: counter - 1] if (e[1], e[0]) == cand])


def test_compress_decompress_several_single():  # stream.rs:703-819
    dt_ref, n = 255, 5
    co = _out(32, 16, n)
    cand, inp, counter = (7, 12), [], 0
    for i in range(60):
        ev = (12, 7, 0xFF, 7, 280 + i * 100 + counter)
        inp.append(ev)
        co.ingest_event(*ev)
        counter += 1
    co.ingest_event(19, 14, 0xFF, 7, 280)  # a late event
    for i in range(60, 70):
        ev = (12, 7, 0xFF, 7, 280 + i * 100 + counter)
        inp.append(ev)
        co.ingest_event(*ev)
        counter += 1
    output = co.close()
    assert len(output) > 0
    events = _decode(output, 32, 16, n)
    _check_candidate(inp, [e for e in events[: counter + 1] if (e[1], e[0]) == cand], check_d=False)


def test_compress_decompress_several_with_skip():  # stream.rs:821-946
    dt_ref, n = 255, 10
    co = _out(30, 30, n)
    cand, inp, counter = (7, 12), [], 0

    def sweep():
        nonlocal counter
        for i in range(10):
            for y in range(30):
                for x in range(30):
                    if not (y == 14 and x == 14 or i % 3 == 0 and y >= 16 and x < 16):
                        ev = (x, y, 0xFF, 7, 280 + counter)
                        if (y, x) == cand:
                            inp.append(ev)
                        co.ingest_event(*ev)
                        counter += 1

    sweep()
    co.ingest_event(14, 14, 0xFF, 7, 280)  # a late event into a pixel that was skipped
    sweep()
    output = co.close()
    assert 0 < len(output) < counter * 9
    events = _decode(output, 30, 30, n)
    _check_candidate(inp, [e for e in events if (e[1], e[0]) == cand])


def test_create_compressed_stream_grows_with_adus():  # compressed/mod.rs:20-79
    def run(ts):
        co = CO.CompressedOutput(100, 100, 1, tps=2550, ref_interval=100, delta_t_max=100, adu_interval=1, time_mode=1)
        for t in ts:
            co.ingest_event(0, 0, 0xFF, 5, t)
        return co.close(), co.header_size
    one, hs = run([100])
    assert len(one) > hs
    three, _ = run([100, 200, 300])
    assert len(three) > len(one)


def _read_raw(path):
    data = open(path, "rb").read()
    ver = data[5]
    hs = 25 + 4 * min(ver, 3)
    w, h = struct.unpack(">HH", data[7:11])
    tps, ref, dtm = struct.unpack(">III", data[11:23])
    es, ch = data[23], data[24]
    body = np.frombuffer(data[hs: hs + (len(data) - hs) // es * es], np.uint8).reshape(-1, es)
    x = body[:, 0].astype(np.uint32) << 8 | body[:, 1]
    y = body[:, 2].astype(np.uint32) << 8 | body[:, 3]
    assert es == 9
    d = body[:, 4]
    t = (body[:, 5].astype(np.uint64) << 24 | body[:, 6].astype(np.uint64) << 16 | body[:, 7].astype(np.uint64) << 8
         | body[:, 8]).astype(np.uint32)
    keep = ~((x == 0xFFFF) & (y == 0xFFFF))  # the EOF event
    return dict(w=w, h=h, tps=tps, ref=ref, dtm=dtm, ver=ver), x[keep], y[keep], d[keep], t[keep]


def test_virat_sample_build_many_frames(golden_dir):  # adder-codec-core/tests/integration_tests.rs:44-85
    meta, x, y, d, t = _read_raw(os.path.join(golden_dir, "virat_small_gray.adder"))
    assert meta["w"] == 192                                   # test_read_adder_raw
    adu_interval = meta["dtm"] // meta["ref"]                 # "a fix since we're reading a v2-encoded file"
    n = min(len(x), 25_000)                                   # pure-Python coder: a prefix of the file
    co = CO.CompressedOutput(meta["w"], meta["h"], 1, tps=meta["tps"], ref_interval=meta["ref"],
                             delta_t_max=meta["dtm"], adu_interval=adu_interval)
    for i in range(n):
        co.ingest_event(int(x[i]), int(y[i]), 0xFF, int(d[i]), int(t[i]))
    out = co.close()
    assert len(out) < n * 9                                   # the reference's only assertion
    # beyond the reference: the stream decodes, every pixel's d sequence survives, t within the lossy tolerance
    ev = CO.decode(out, width=meta["w"], height=meta["h"], channels=1, ref_interval=meta["ref"],
                   adu_interval=adu_interval, header_size=co.header_size)
    assert 0 < len(ev) <= n
    by_px = {}
    for i in range(n):
        by_px.setdefault((int(x[i]), int(y[i])), []).append((int(d[i]), int(t[i])))
    got_px = {}
    for (ex, ey, ec, ed, et) in ev:
        got_px.setdefault((ex, ey), []).append((ed, et))
    assert set(got_px) <= set(by_px)
    exact = sum(1 for k, v in got_px.items() if [a for a, _ in v] == [a for a, _ in by_px[k]][: len(v)])
    assert exact > 0.9 * len(got_px)
