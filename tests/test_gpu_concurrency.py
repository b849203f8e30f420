"""The C-ABI is re-entrant like the reference's (SURVEY 8b "Threading"): calls from several host threads at once, mixed
encode/decode, different pipelines and shapes, more threads than pooled contexts -- every result must be the oracle's."""
import threading

import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def test_many_threads_mixed_calls(sqy, oracle):
    jobs = [("bitswap1->lz4", synth.stack((24, 128, 160), np.uint16)),
            ("diff3x3x1->bitswap1->lz4", synth.stack((20, 96, 128), np.uint16)),
            ("lz4", synth.stack((16, 200, 300), np.uint8)),
            ("quantiser->bitswap1->lz4", synth.stack((12, 128, 128), np.uint16)),
            ("raster_reorder->lz4", synth.stack((16, 64, 96), np.uint16)),
            ("frame_shuffle->lz4", synth.stack((18, 64, 128), np.uint8))]
    want = [oracle.pipeline_encode(p, v) for p, v in jobs]
    back = [oracle.pipeline_decode(b) for b in want]
    errors = []

    def worker(t):
        try:
            for it in range(8):
                k = (t + it) % len(jobs)
                p, v = jobs[k]
                extra = 16 * v.shape[0] + 512 if "frame_shuffle" in p else 0
                rc, blob = sqy.encode(p, v, nthreads=0, extra_capacity=extra)
                assert rc == 0 and blob == want[k], ("encode", t, it, p)
                rc, dec = sqy.decode(blob)
                assert rc == 0 and np.array_equal(dec, back[k]), ("decode", t, it, p)
        except Exception as e:   # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(12)]     # more than the 8 pooled contexts
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors[:3]


def test_concurrent_full_width_diff_decodes(sqy, oracle):
    """six host threads decode a volume with a diff3x3x1 stage at once (round 2: a one-launch kernel whose strips waited for each
    other and could end up half resident side by side; now a chain of one ordinary launch per frame: nothing waits for a workgroup)"""
    import time
    vol = synth.stack((48, 1024, 256), np.uint16)
    blob = oracle.pipeline_encode("diff3x3x1->bitswap1->lz4", vol)
    errors = []

    def worker(t):
        try:
            for _ in range(4):
                rc, dec = sqy.decode(blob)
                assert rc == 0 and np.array_equal(dec, vol), ("decode", t)
        except Exception as e:   # pragma: no cover
            errors.append(repr(e))

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors[:3]
    assert time.perf_counter() - t0 < 60, "decodes stalled (strip kernels waiting for strips that could not start?)"


def test_diff_decode_next_to_encodes_in_flight(sqy, oracle):
    """the diff3x3x1 decode (a chain of 47 frame launches + a copy on a side stream) while encodes of a large stack keep the chip's LDS
    full of LZ4 chunk waves on other streams: the decode may be late, it must not stall or go wrong"""
    import time
    vol = synth.stack((48, 1024, 256), np.uint16)
    blob = oracle.pipeline_encode("diff3x3x1->bitswap1->lz4", vol)
    big = synth.stack((64, 1024, 1024), np.uint16)                     # 128 MiB: 512 chunks per encode
    want_big = oracle.pipeline_encode("bitswap1->lz4", big)
    errors, stop = [], threading.Event()

    def encoder(t):
        try:
            while not stop.is_set():
                rc, b = sqy.encode("bitswap1->lz4", big, nthreads=0)
                assert rc == 0 and b == want_big, ("encode", t)
        except Exception as e:   # pragma: no cover
            errors.append(repr(e))

    enc = [threading.Thread(target=encoder, args=(t,)) for t in range(2)]
    for th in enc:
        th.start()
    t0 = time.perf_counter()
    try:
        for _ in range(6):
            rc, dec = sqy.decode(blob)
            assert rc == 0 and np.array_equal(dec, vol)
    finally:
        stop.set()
        for th in enc:
            th.join(timeout=600)
    assert not errors, errors[:3]
    assert time.perf_counter() - t0 < 120


def test_serial_layout_calls_in_flight(sqy, oracle):
    """the block-parallel paths of the serial layout (nthreads = 1: table guesses + verify + second parse on the way in, symbolic decode
    + tail scan on the way out) from ten host threads at once, each context with its own tables and references"""
    rng = np.random.default_rng(4)
    n = 30 * (256 << 10) + 999
    a = np.zeros(n, np.uint8); idx = rng.integers(0, n, n // 40); a[idx] = rng.integers(1, 256, idx.size)
    b = rng.integers(0, 256, n, dtype=np.uint8)
    for i in range(256 << 10, n - 400, 256 << 10):               # guesses that fail: a second parse runs as well
        b[i:i + 300] = b[i - 1000:i - 700]
    jobs = [("lz4", a.reshape(1, 1, -1)), ("lz4", b.reshape(1, 1, -1)), ("bitswap1->lz4", synth.stack((40, 256, 256), np.uint16)),
            ("lz4(blocksize_kb=64)", np.repeat(rng.integers(0, 4, n // 64 + 1, dtype=np.uint8), 64)[:n].reshape(1, 1, -1))]
    want = [oracle.pipeline_encode(p, v, nthreads=1) for p, v in jobs]
    errors = []

    def worker(t):
        try:
            for it in range(5):
                k = (t + it) % len(jobs)
                p, v = jobs[k]
                rc, blob = sqy.encode(p, v, nthreads=1)
                assert rc == 0 and blob == want[k], ("encode", t, it, p)
                rc, dec = sqy.decode(blob)
                assert rc == 0 and np.array_equal(dec, v), ("decode", t, it, p)
        except Exception as e:   # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(10)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors[:3]
