"""The product's CPU compressed sink / source (include/adder_compressed.h, csrc/adder_compressed.cpp) against the
oracle restatement (oracle/compressed_oracle.py): the STREAM must be byte-identical, and so must what both decode
from it.  No GPU needed: this stage runs on the host."""
import os

import numpy as np
import pytest

import adder_amd as A
from oracle import compressed_oracle as CO
from test_compressed_oracle import _read_raw


def _oracle_stream(events, w, h, c, *, tps, ref, dtm, adu, tm=1, cmax=7, header=True):
    co = CO.CompressedOutput(w, h, c, tps=tps, ref_interval=ref, delta_t_max=dtm, adu_interval=adu, time_mode=tm,
                             c_thresh_max=cmax, write_header=header)
    for e in events:
        co.ingest_event(int(e["x"]), int(e["y"]), int(e["c"]), int(e["d"]), int(e["t"]))
    return co.close(), co.header_size


def _product_stream(events, w, h, c, *, tps, ref, dtm, adu, tm=1, cmax=7, header=True, threads=0, pieces=1):
    enc = A.CompressedEncoder(w, h, c, tps=tps, ref_interval=ref, delta_t_max=dtm, adu_interval=adu, time_mode=tm,
                              c_thresh_max=cmax, write_header=header, threads=threads)
    for part in np.array_split(events, pieces):
        enc.ingest(part)
    out = enc.close()
    enc.destroy()
    return out


def _events(rows):
    ev = np.zeros(len(rows), A.EVENT_DTYPE)
    for i, (x, y, c, d, t) in enumerate(rows):
        ev[i] = (x, y, c, d, 0, t)
    return ev


def _oracle_decode(data, w, h, c, ref, adu, hs):
    return _events(CO.decode(data, width=w, height=h, channels=c, ref_interval=ref, adu_interval=adu, header_size=hs))


def _both(events, w, h, c, **kw):
    want, hs = _oracle_stream(events, w, h, c, **kw)
    for threads, pieces in ((1, 1), (4, 7)):
        got = _product_stream(events, w, h, c, threads=threads, pieces=pieces, **kw)
        assert got == want, (threads, pieces, len(got), len(want))
    dec, p = A.compressed_decode(want, has_header=kw.get("header", True), width=w, height=h, channels=c,
                                 ref_interval=kw["ref"], adu_interval=kw["adu"])
    assert np.array_equal(dec, _oracle_decode(want, w, h, c, kw["ref"], kw["adu"], hs))
    return want, dec


def test_reference_round_trip_scenarios_byte_identical():
    # the streams of stream.rs:510-946 (bare CompressedOutput: no header)
    rows, counter = [], 0
    for _ in range(10):
        for y in range(30):
            for x in range(16):
                rows.append((x, y, 0xFF, 7, 280 + counter))
                counter += 1
    _both(_events(rows), 16, 32, 1, tps=7650, ref=255, dtm=255 * 5, adu=5, header=False)
    rows, counter = [], 0
    for i in range(60):
        rows.append((12, 7, 0xFF, 7, 280 + i * 100 + counter))
        counter += 1
    rows.append((19, 14, 0xFF, 7, 280))
    for i in range(60, 70):
        rows.append((12, 7, 0xFF, 7, 280 + i * 100 + counter))
        counter += 1
    _both(_events(rows), 32, 16, 1, tps=7650, ref=255, dtm=255 * 5, adu=5, header=False)
    rows, counter = [], 0
    for rep in range(2):
        for i in range(10):
            for y in range(30):
                for x in range(30):
                    if not (y == 14 and x == 14 or i % 3 == 0 and y >= 16 and x < 16):
                        rows.append((x, y, 0xFF, 7, 280 + counter))
                        counter += 1
        if rep == 0:
            rows.append((14, 14, 0xFF, 7, 280))
    _both(_events(rows), 30, 30, 1, tps=7650, ref=255, dtm=2550, adu=10, header=False)


@pytest.mark.parametrize("channels,cmax", [(1, 7), (3, 7), (1, 0), (3, 25)])
def test_random_streams_byte_identical(channels, cmax):
    rng = np.random.default_rng(channels * 100 + cmax)
    w, h, n = 70, 45, 4000
    ev = np.zeros(n, A.EVENT_DTYPE)
    ev["x"] = rng.integers(0, w, n)
    ev["y"] = rng.integers(0, h, n)
    ev["c"] = 0xFF if channels == 1 else rng.integers(0, channels, n)
    ev["d"] = rng.choice(np.array([0, 1, 3, 5, 7, 8, 9, 12, 128, 255], np.uint8), n)
    # mostly increasing time with jitter: late events, repeats (the drop rule), jumps over several ADUs,
    # large inter-event gaps (bit-shifted and full-width residuals)
    t = np.cumsum(rng.integers(0, 40, n)) + rng.integers(-300, 300, n)
    t[n // 2:] += 40_000
    ev["t"] = np.clip(t, 0, None)
    ev["t"][rng.integers(0, n, 50)] = rng.integers(0, 2 ** 31, 50)  # wild values
    _both(ev, w, h, channels, tps=7650, ref=255, dtm=7650, adu=30, cmax=cmax)
    _both(ev[: n // 3], w, h, channels, tps=5000, ref=100, dtm=100, adu=1, cmax=cmax)


def test_transcoder_stream_round_trip_within_tolerance():
    """A real transcoder stream (the CPU oracle's events for a lossy AbsoluteT clip): compress with the product,
    decode, and compare per pixel: every d sequence survives and t stays within the lossy tolerance."""
    from oracle import oracle as O
    import clips
    W, H, T = 48, 40, 90
    clip = clips.make_clip("jitter", T, H, W, 1, seed=12)
    ov = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    ov.set_crf_parameters(7, 7)
    ov.reset_c_thresh(2)
    ev = np.concatenate([ov.integrate_matrix(f) for f in clip])
    want, dec = _both(ev, W, H, 1, tps=7650, ref=255, dtm=7650, adu=30)
    assert len(want) < len(ev) * 9
    key_in = ev["y"].astype(np.int64) * W + ev["x"]
    key_out = dec["y"].astype(np.int64) * W + dec["x"]
    for px in np.unique(key_in)[:400]:
        a, b = ev[key_in == px], dec[key_out == px]
        a = a[np.argsort(a["t"], kind="stable")]
        assert len(b) <= len(a)
        assert np.array_equal(np.sort(b["d"]), np.sort(a["d"][: len(b)])) or len(b) < len(a)


def test_virat_sample_full_file_product(golden_dir):
    """adder-codec-core/tests/integration_tests.rs:44-85 (test_build_many_frames) with the product, on the
    WHOLE file; the first 15 000 events also byte-for-byte against the oracle."""
    meta, x, y, d, t = _read_raw(os.path.join(golden_dir, "virat_small_gray.adder"))
    ev = np.zeros(len(x), A.EVENT_DTYPE)
    ev["x"], ev["y"], ev["c"], ev["d"], ev["t"] = x, y, 0xFF, d, t
    adu = meta["dtm"] // meta["ref"]
    kw = dict(tps=meta["tps"], ref=meta["ref"], dtm=meta["dtm"], adu=adu)
    out = _product_stream(ev, meta["w"], meta["h"], 1, **kw)
    assert len(out) < len(ev) * 9
    dec, p = A.compressed_decode(out)
    assert (p.width, p.height, p.adu_interval) == (meta["w"], meta["h"], adu) and 0 < len(dec) <= len(ev)
    _both(ev[:15_000], meta["w"], meta["h"], 1, **kw)


def test_errors():
    with pytest.raises(A.AdderHipError):
        A.CompressedEncoder(0, 10, 1, tps=1, ref_interval=1, delta_t_max=1, adu_interval=1)
    enc = A.CompressedEncoder(8, 8, 1, tps=1, ref_interval=255, delta_t_max=255, adu_interval=1)
    with pytest.raises(A.AdderHipError):
        enc.ingest(_events([(9, 0, 0xFF, 1, 5)]))
    enc.destroy()
    with pytest.raises(A.AdderHipError):
        A.compressed_decode(b"adder" + bytes(40))


def test_decode_of_corrupt_or_truncated_adus_terminates():
    """A truncated or corrupt ADU must end in an error (or a shorter, valid event list) in bounded time and memory: the
    reference's decoder returns an error at the end of its input (arithmetic-coding-adder-dep/src/decoder.rs), it
    never keeps producing events.  Reads past the coded data and the EOF symbol are errors here."""
    import struct
    import time
    rng = np.random.default_rng(3)
    n = 3000
    ev = np.zeros(n, A.EVENT_DTYPE)
    ev["x"], ev["y"], ev["c"] = rng.integers(0, 33, n), rng.integers(0, 20, n), 0xFF
    ev["d"] = rng.choice(np.array([0, 3, 7, 8, 128, 255], np.uint8), n)
    ev["t"] = np.cumsum(rng.integers(0, 3, n))  # everything inside the first ADU's span
    good = _product_stream(ev, 33, 20, 1, tps=7650, ref=255, dtm=255 * 40, adu=40, header=False)
    n_adu = struct.unpack(">I", good[:4])[0]
    assert 8 < n_adu <= len(good) - 4
    cases = []
    for keep in (0, 1, 2, 5, n_adu // 4, n_adu // 2, n_adu - 2):  # the first ADU cut short, its length prefix fixed up
        cases.append(struct.pack(">I", keep) + good[4:4 + keep])
    rng = np.random.default_rng(5)
    for k in range(24):  # byte noise inside the first ADU
        b = bytearray(good[:4 + n_adu])
        for pos in rng.integers(4, 4 + n_adu, 1 + k % 5):
            b[pos] = rng.integers(0, 256)
        cases.append(bytes(b))
    cases.append(struct.pack(">I", 64) + bytes(64))        # all zeros
    cases.append(struct.pack(">I", 64) + bytes([255]) * 64)  # all ones
    t0 = time.time()
    errors = 0
    for data in cases:
        try:
            dec, _ = A.compressed_decode(data, has_header=False, width=33, height=20, channels=1, ref_interval=255,
                                         adu_interval=40)
            assert len(dec) < 1 << 23
        except A.AdderHipError:
            errors += 1
    assert errors >= 5          # the truncated ones at the very least
    assert time.time() - t0 < 60.0
    with pytest.raises(A.AdderHipError):  # a header (or caller) that asks for an absurd plane
        A.compressed_decode(good, has_header=False, width=65535, height=65535, channels=3, ref_interval=255,
                            adu_interval=40)
