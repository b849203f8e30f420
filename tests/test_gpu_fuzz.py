"""Seeded random sweep over shapes x data x pipelines: blob bytes against the oracle and SQY_Decode back to the input.
Small volumes (the oracle finishes each in well under a second); the structured cases live in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PIPES_U16 = ["bitswap1->lz4", "lz4", "diff3x3x1->bitswap1->lz4", "diff3x3x1->lz4", "frame_shuffle->lz4", "frame_shuffle->bitswap1->lz4",
             "raster_reorder->lz4", "raster_reorder->bitswap1->lz4", "quantiser->bitswap1->lz4", "quantiser->lz4", "bitswap1",
             "lz4(blocksize_kb=64,framestep_kb=64)", "bitswap1->lz4(n_chunks_of_input=3)"]
PIPES_U8 = ["bitswap1->lz4", "lz4", "frame_shuffle->lz4", "raster_reorder->lz4", "diff3x3x1->lz4", "lz4(blocksize_kb=64,framestep_kb=64)"]


def _data(rng, shape, dtype, kind):
    n = int(np.prod(shape))
    hi = 65536 if dtype == np.uint16 else 256
    if kind == 0:
        a = rng.integers(0, hi, n)
    elif kind == 1:                                            # narrow band + rare outliers (microscopy-like)
        a = rng.integers(90, 140, n)
        m = rng.random(n) < 0.01
        a[m] = rng.integers(0, hi, int(m.sum()))
    elif kind == 2:                                            # long runs
        a = np.repeat(rng.integers(0, hi, n // 97 + 1), 97)[:n]
    elif kind == 3:                                            # short period
        p = int(rng.integers(1, 40))
        a = np.tile(rng.integers(0, hi, p), n // p + 1)[:n]
    elif kind == 4:                                            # sparse
        a = np.zeros(n, np.int64)
        idx = rng.integers(0, n, max(1, n // 50))
        a[idx] = rng.integers(0, hi, idx.size)
    else:                                                      # smooth ramp + noise
        a = (np.arange(n) // 7 + rng.integers(0, 4, n)) % hi
    return a.astype(dtype).reshape(shape)


def _cases():
    rng = np.random.default_rng(20240917)
    out = []
    for i in range(72):
        dtype = np.uint16 if i % 3 else np.uint8
        pipes = PIPES_U16 if dtype == np.uint16 else PIPES_U8
        pipe = pipes[int(rng.integers(0, len(pipes)))]
        z, y, x = int(rng.integers(3, 40)), int(rng.integers(4, 120)), int(rng.integers(4, 200))
        if pipe.startswith("raster_reorder"):                  # defined geometries only: full tiles, or a remainder everywhere
            ts = 16 // np.dtype(dtype).itemsize
            if i % 2:
                z, y, x = [max(ts, v - v % ts) for v in (z, y, x)]
            else:
                z, y, x = [v + 1 if v % ts == 0 else v for v in (z + ts, y + ts, x + ts)]
        if "diff3x3x1" in pipe and dtype == np.uint8:
            z, y, x = min(z, 100), min(y, 100), min(x, 100)
        out.append((i, pipe, (z, y, x), dtype, int(rng.integers(0, 6)), int(rng.integers(0, 2 ** 31))))
    return out


@pytest.mark.parametrize("i,pipeline,shape,dtype,kind,seed", _cases(), ids=lambda v: str(v) if not isinstance(v, type) else v.__name__)
def test_random_case(sqy, oracle, i, pipeline, shape, dtype, kind, seed):
    vol = _data(np.random.default_rng(seed), shape, dtype, kind)
    try:
        want = oracle.pipeline_encode(pipeline, vol)
    except (ValueError, NotImplementedError):
        pytest.skip("shape outside what the reference defines for this pipeline")
    extra = 16 * shape[0] + 512 if "frame_shuffle" in pipeline else 0
    rc, blob = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=extra)
    assert rc == 0
    assert blob == want, (pipeline, shape, np.dtype(dtype).name, kind)
    rc, back = sqy.decode(blob)
    assert rc == 0
    if pipeline.startswith("quantiser"):
        assert np.array_equal(back, oracle.pipeline_decode(want))
    elif "frame_shuffle" in pipeline:
        # frames with equal metrics map to the same source frame (std::find in the reference): the stage is then not
        # invertible and the oracle's own decode does not restore the input either
        if not np.array_equal(back, vol):
            assert not np.array_equal(oracle.pipeline_decode(want), vol)
    else:
        assert np.array_equal(back, vol)


# ---- round 2: the new stages and both LZ4 layouts in the sweep; corrupted blobs never crash the decoder ----
PIPES2_U16 = ["zcurve_reorder->lz4", "zcurve_reorder(tile_size=4)->bitswap1->lz4", "bitshuffle->lz4", "bitshuffle(block_size=128)->lz4",
              "diff3x3x1->bitshuffle->lz4", "quantiser->bitshuffle->lz4", "pass_through->bitswap1->lz4", "bitswap1->lz4", "diff3x3x1->bitswap1->lz4",
              "lz4(blocksize_kb=64)", "bitswap1->lz4(blocksize_kb=64,n_chunks_of_input=2)", "tile_shuffle(tile_size=4)->lz4"]
PIPES2_U8 = ["zcurve_reorder->lz4", "bitshuffle->lz4", "bitswap1->lz4", "lz4(blocksize_kb=64)", "pass_through->lz4", "tile_shuffle(tile_size=2)->lz4"]


def _cases2():
    rng = np.random.default_rng(7771)
    out = []
    for i in range(60):
        dtype = np.uint16 if i % 3 else np.uint8
        pipes = PIPES2_U16 if dtype == np.uint16 else PIPES2_U8
        pipe = pipes[int(rng.integers(0, len(pipes)))]
        z, y, x = int(rng.integers(3, 48)), int(rng.integers(4, 160)), int(rng.integers(4, 260))
        if pipe.startswith("tile_shuffle"):
            ts = 4 if "4" in pipe else 2
            z, y, x = [max(ts, v - v % ts) for v in (z, y, x)]
        out.append((i, pipe, (z, y, x), dtype, int(rng.integers(0, 6)), int(rng.integers(0, 2 ** 31)), 1 if i % 2 else 2))
    return out


@pytest.mark.parametrize("i,pipeline,shape,dtype,kind,seed,nthreads", _cases2(), ids=lambda v: str(v) if not isinstance(v, type) else v.__name__)
def test_random_case_round2(sqy, oracle, i, pipeline, shape, dtype, kind, seed, nthreads):
    vol = _data(np.random.default_rng(seed), shape, dtype, kind)
    try:
        want = oracle.pipeline_encode(pipeline, vol, nthreads=nthreads)
    except (ValueError, NotImplementedError):
        pytest.skip("shape outside what the reference defines for this pipeline")
    extra = 16 * vol.size // 8 + 512 if "tile_shuffle" in pipeline else None
    rc, blob = sqy.encode(pipeline, vol, nthreads=nthreads, extra_capacity=extra)
    assert rc == 0
    assert blob == want, (pipeline, shape, np.dtype(dtype).name, kind, nthreads)
    rc, back = sqy.decode(blob)
    assert rc == 0
    assert np.array_equal(back, oracle.pipeline_decode(want))
    if not pipeline.startswith(("quantiser", "tile_shuffle")):
        assert np.array_equal(back, vol)


def test_corrupted_blobs_never_crash(sqy, oracle):
    """random damage to header fields and payload bytes of blobs of every stage: the decoder answers (any code, any data) and
    the next, undamaged decode still works -- no fault, no hang, no exception across the ABI"""
    rng = np.random.default_rng(99)
    vol = _data(rng, (16, 40, 64), np.uint16, 1)                # (extents the tiled stages are defined for)
    good = {}
    for pipe in ("bitswap1->lz4", "zcurve_reorder->lz4", "tile_shuffle(tile_size=4)->lz4", "bitshuffle->lz4", "frame_shuffle->lz4",
                 "quantiser->bitswap1->lz4", "diff3x3x1->bitswap1->lz4", "raster_reorder->lz4"):
        good[pipe] = (oracle.pipeline_encode(pipe, vol), oracle.pipeline_encode(pipe, vol, nthreads=1))
    for pipe, blobs in good.items():
        for blob in blobs:
            h = oracle.header_unpack(blob)
            for trial in range(12):
                b = bytearray(blob)
                if trial % 3 == 0:                             # somewhere in the header text (numbers, names, base64 maps)
                    p = int(rng.integers(0, h["size"]))
                    b[p] = int(rng.integers(32, 127))
                elif trial % 3 == 1:                           # somewhere in the payload
                    for _ in range(int(rng.integers(1, 6))):
                        b[int(rng.integers(h["size"], len(b)))] = int(rng.integers(0, 256))
                else:                                          # cut short
                    b = b[:int(rng.integers(h["size"] // 2, len(b)))]
                rc, back = sqy.decode(bytes(b))
                assert rc in (0, 1, 11, 101), (pipe, trial, rc)      # the documented set: ok, tail filter / header, sink (+10), head filter (+100)
            rc, back = sqy.decode(blob)
            assert rc == 0, pipe
