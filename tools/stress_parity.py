"""Randomised parity stress (GPU box): random pipelines x shapes x data kinds x LZ4 layouts, encode compared byte for byte with the
oracle, decode compared with the oracle's decode, until the time budget is used.  Prints the first mismatch and exits non-zero.
    python tools/stress_parity.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sqeazy_amd
from oracle import sqy_oracle as o

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
rng = np.random.default_rng(seed)
PIPES16 = ["bitswap1->lz4", "lz4", "diff3x3x1->bitswap1->lz4", "diff3x3x1->lz4", "frame_shuffle->lz4", "raster_reorder->lz4",
           "quantiser->bitswap1->lz4", "quantiser->lz4", "zcurve_reorder->lz4", "bitshuffle->lz4", "tile_shuffle(tile_size=4)->lz4",
           "lz4(blocksize_kb=64)", "bitswap1->lz4(blocksize_kb=64,framestep_kb=256)", "bitswap1->lz4(n_chunks_of_input=3)",
           "diff3x3x1->bitshuffle->lz4", "pass_through->bitswap1->lz4", "bitswap1",
           # round 5: the reorder / shuffle stages behind the sink (tail filters on char)
           "quantiser->raster_reorder->lz4", "quantiser->zcurve_reorder(tile_size=4)->bitswap1->lz4", "quantiser->tile_shuffle(tile_size=8)->lz4",
           "quantiser->frame_shuffle->lz4", "quantiser->diff3x3x1->lz4"]
PIPES8 = ["bitswap1->lz4", "lz4", "frame_shuffle->lz4", "raster_reorder->lz4", "zcurve_reorder->lz4", "bitshuffle->lz4",
          "lz4(blocksize_kb=64)", "diff3x3x1->lz4"]


def data(shape, dtype, kind):
    n = int(np.prod(shape)); hi = 65536 if dtype == np.uint16 else 256
    if kind == 0: a = rng.integers(0, hi, n)
    elif kind == 1:
        a = rng.integers(90, 140, n); m = rng.random(n) < 0.01; a[m] = rng.integers(0, hi, int(m.sum()))
    elif kind == 2: a = np.repeat(rng.integers(0, hi, n // 97 + 1), 97)[:n]
    elif kind == 3:
        p = int(rng.integers(1, 80)); a = np.tile(rng.integers(0, hi, p), n // p + 1)[:n]
    elif kind == 4:
        a = np.zeros(n, np.int64); idx = rng.integers(0, n, max(1, n // 50)); a[idx] = rng.integers(0, hi, idx.size)
    elif kind == 5: a = (np.arange(n) // 7 + rng.integers(0, 4, n)) % hi
    elif kind == 6:                                            # medium runs: matches of 4..300 bytes at small offsets
        a = np.empty(n + 400, np.int64); pos = 0
        while pos < n:
            lit = int(rng.integers(0, 12)); a[pos:pos + lit] = rng.integers(0, hi, lit); pos += lit
            p = int(rng.integers(1, 40)); run = int(rng.integers(4, 200))
            a[pos:pos + run] = np.tile(rng.integers(0, hi, p), run // p + 1)[:run]; pos += run
        a = a[:n]
    elif kind == 7:                                            # low-entropy bytes: many short matches
        a = rng.integers(0, 4, n) * (hi // 4)
    elif kind == 8:                                            # noise with all-zero stretches (holes inside stored chunks)
        a = rng.integers(0, hi, n); step = int(rng.integers(3, 400)) * 8192
        for s0 in range(int(rng.integers(0, 8192)), n, step): a[s0:s0 + int(rng.integers(1, 3)) * 8192] = 0
    else:                                                      # small values in a few blobs, zero elsewhere (zero pieces inside sparse chunks, empty top planes)
        a = np.zeros(n, np.int64)
        for _ in range(int(rng.integers(1, 6))):
            s0 = int(rng.integers(0, n)); ln = int(rng.integers(1, max(2, n // 4)))
            a[s0:s0 + ln] = rng.integers(0, 1 << int(rng.integers(1, 13)), min(ln, n - s0))
    return a.astype(dtype).reshape(shape)


t0 = time.time(); cases = 0; skipped = 0; last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60:
        print("  ... %d cases equal so far" % cases, flush=True); last_note = time.time()
    dtype = np.uint16 if rng.random() < 0.7 else np.uint8
    pipe = str(rng.choice(PIPES16 if dtype == np.uint16 else PIPES8))
    big = rng.random() < float(os.environ.get("STRESS_BIG", "0.15"))
    z, y, x = (int(rng.integers(2, 70)), int(rng.integers(3, 300)), int(rng.integers(3, 400))) if not big else \
              (int(rng.integers(8, 40)), int(rng.integers(200, 700)), int(rng.integers(256, 1100)))
    if pipe.startswith(("tile_shuffle", "raster_reorder", "zcurve")) and rng.random() < 0.7:
        z, y, x = [max(8, v - v % 8) for v in (z, y, x)]
    if "diff3x3x1" in pipe and dtype == np.uint8:
        z, y, x = min(z, 100), min(y, 100), min(x, 100)
    if "diff3x3x1" in pipe and rng.random() < 0.5:
        x = max(8, x - x % 8)                                   # the strip kernel's geometry
    nth = 1 if rng.random() < float(os.environ.get("STRESS_SERIAL", "0.3")) else 0     # (the serial layout: block-parallel encode and decode since round 4)
    if dtype == np.uint16 and rng.random() < 0.35:
        # frames in place (and the holes in them): a 16-bit bitswap1 in front of a chunked lz4, whole tiles of 8192 voxels
        pipe = str(rng.choice(["bitswap1->lz4", "diff3x3x1->bitswap1->lz4", "bitswap1->lz4(blocksize_kb=64,framestep_kb=64)"]))
        y, x = max(64, y - y % 64), max(128, x - x % 128)
        nth = 0 if rng.random() < 0.7 else nth
    vol = data((z, y, x), dtype, int(rng.integers(0, 10)))
    try:
        want = o.pipeline_encode(pipe, vol, nthreads=nth)
    except (ValueError, NotImplementedError):
        skipped += 1
        continue
    extra = 16 * vol.size // 8 + 16 * z + 1024 if ("shuffle" in pipe) else None
    rc, blob = sqeazy_amd.encode(pipe, vol, nthreads=nth, extra_capacity=extra)
    if rc != 0 or blob != want:
        print("ENCODE MISMATCH", pipe, (z, y, x), np.dtype(dtype).name, "nthreads", nth, "rc", rc, "seed", seed, "case", cases)
        sys.exit(1)
    rc, back = sqeazy_amd.decode(want)
    ref = o.pipeline_decode(want)
    # (frames with equal metrics share a source frame (std::find): the stage is not invertible then; the frames the map does not name
    # come out as zeros on both sides -- compared like everything else)
    if rc != 0 or not np.array_equal(back, ref):
        print("DECODE MISMATCH", pipe, (z, y, x), np.dtype(dtype).name, "nthreads", nth, "rc", rc, "seed", seed, "case", cases)
        sys.exit(1)
    cases += 1
print("stress_parity: %d cases equal (%d shapes the reference does not define skipped), seed %d, %.0f s" % (cases, skipped, seed, time.time() - t0))
