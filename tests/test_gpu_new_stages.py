"""GPU parity of the stages added in round 2 -- zcurve_reorder, tile_shuffle, bitshuffle -- through the C-ABI against the
oracle (tile_shuffle / zcurve_reorder pinned by the reference's tile / label vectors, bitshuffle parity-unpinned: its
library is not in the reference tree)."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _rt(sqy, oracle, pipeline, vol, lossless=True, extra=None):
    rc, blob = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=extra)
    assert rc == 0, pipeline
    want = oracle.pipeline_encode(pipeline, vol)
    assert blob == want, "%s on %r: %d vs %d bytes" % (pipeline, vol.shape, len(blob), len(want))
    rc, back = sqy.decode(blob)
    assert rc == 0, pipeline
    assert np.array_equal(back, oracle.pipeline_decode(blob)), pipeline
    if lossless:
        assert np.array_equal(back, vol), pipeline


@pytest.mark.parametrize("shape,ts", [((8, 8, 8), 2), ((8, 8, 8), 4), ((8, 8, 8), 8), ((8, 16, 8), 2), ((7, 16, 7), 2), ((5, 3, 9), 2),
                                      ((64, 128, 256), 16), ((33, 70, 129), 4), ((128, 128, 128), 128), ((12, 12, 12), 8)])
def test_zcurve_reorder(sqy, oracle, shape, ts):
    vol = synth.stack(shape) if min(shape) >= 8 else np.random.default_rng(1).integers(0, 65536, shape, dtype=np.uint16)
    _rt(sqy, oracle, "zcurve_reorder(tile_size=%d)->lz4" % ts, vol)
    _rt(sqy, oracle, "zcurve_reorder(tile_size=%d)" % ts, vol)
    v8 = synth.stack(shape, np.uint8) if min(shape) >= 8 else (vol & 0xff).astype(np.uint8)
    _rt(sqy, oracle, "zcurve_reorder(tile_size=%d)->bitswap1->lz4" % ts, v8)


def test_zcurve_reorder_default_tile_and_refusals(sqy, oracle):
    _rt(sqy, oracle, "zcurve_reorder->bitswap1->lz4", synth.stack((16, 32, 48)))
    for shape, ts in (((8, 8, 8), 3), ((8, 8, 8), 16), ((16, 16, 8), 16), ((8, 8, 8), 256)):
        assert sqy.encode("zcurve_reorder(tile_size=%d)->lz4" % ts, np.zeros(shape, np.uint16), nthreads=2)[0] == 1


@pytest.mark.parametrize("shape,ts", [((8, 8, 8), 4), ((8, 8, 8), 2), ((32, 64, 96), 32), ((64, 64, 64), 16), ((16, 48, 32), 8)])
def test_tile_shuffle(sqy, oracle, shape, ts):
    # tiles with distinct metrics (so that the stage is invertible): a ramp over the tiles plus noise that keeps the order
    rng = np.random.default_rng(5)
    n = [d // ts for d in shape]
    ntiles = n[0] * n[1] * n[2]
    z, y, x = np.indices(shape)
    t = (z // ts) * n[1] * n[2] + (y // ts) * n[2] + x // ts
    perm = rng.permutation(ntiles)
    vol = (perm[t] * 40 + rng.integers(0, 8, shape)).astype(np.uint16)
    _rt(sqy, oracle, "tile_shuffle(tile_size=%d)->lz4" % ts, vol, extra=16 * ntiles + 256)
    _rt(sqy, oracle, "tile_shuffle(tile_size=%d)->bitswap1->lz4" % ts, vol, extra=16 * ntiles + 256)
    # real data: equal metrics map several slots to the first such tile -- bytes must still match the oracle, decode too
    _rt(sqy, oracle, "tile_shuffle(tile_size=%d)->lz4" % ts, synth.stack(shape), lossless=False, extra=16 * ntiles + 256)
    _rt(sqy, oracle, "tile_shuffle(tile_size=%d)->lz4" % ts, synth.stack(shape, np.uint8), lossless=False, extra=16 * ntiles + 256)


def test_tile_shuffle_sums_beyond_2_pow_24_and_refusals(sqy, oracle):
    """32^3 tiles of large values: the sequential binary32 sum rounds on the way (order of the additions matters)"""
    rng = np.random.default_rng(6)
    vol = rng.integers(30000, 65536, (32, 64, 64), dtype=np.uint16)
    _rt(sqy, oracle, "tile_shuffle->lz4", vol, lossless=False, extra=4096)
    assert sqy.encode("tile_shuffle(tile_size=3)->lz4", np.zeros((8, 8, 8), np.uint16), nthreads=2)[0] == 1
    assert sqy.encode("tile_shuffle->lz4", np.zeros((16, 16, 16), np.uint16), nthreads=2)[0] == 1


@pytest.mark.parametrize("n", [4096, 3 * 4096 + 5, 8192 * 4 + 4104, 100, 9, 8, 7, 1, 1 << 20])
def test_bitshuffle(sqy, oracle, n):
    rng = np.random.default_rng(n)
    for dt in (np.uint16, np.uint8):
        vol = rng.integers(0, np.iinfo(dt).max + 1, n, dtype=dt).reshape(1, 1, n)
        _rt(sqy, oracle, "bitshuffle->lz4", vol)
        _rt(sqy, oracle, "bitshuffle", vol)
    vol = rng.integers(0, 65536, n, dtype=np.uint16).reshape(1, 1, n)
    _rt(sqy, oracle, "bitshuffle(block_size=64)->lz4", vol)
    _rt(sqy, oracle, "bitshuffle(block_size=8192)->lz4", vol)


def test_bitshuffle_pipelines(sqy, oracle):
    vol = synth.stack((24, 64, 96))
    _rt(sqy, oracle, "bitshuffle->lz4", vol)                           # the reference's full-pipeline benchmark (bench/benchmark_full_pipeline_impl.cpp:11)
    _rt(sqy, oracle, "diff3x3x1->bitshuffle->lz4", vol)
    _rt(sqy, oracle, "quantiser->bitshuffle->lz4", vol, lossless=False)
    _rt(sqy, oracle, "bitshuffle->lz4", synth.stack((24, 64, 96), np.uint8))
    rc, blob = sqy.encode("bitshuffle->lz4", vol, nthreads=1)          # and the serial LZ4 layout behind it
    assert rc == 0 and blob == oracle.pipeline_encode("bitshuffle->lz4", vol, nthreads=1)


def test_pass_through_sink(sqy, oracle):
    """pass_through_scheme_impl.hpp:66-95: the sink that only re-types the voxels to bytes; tail filters then work on bytes"""
    for vol in (synth.stack((12, 40, 56)), synth.stack((12, 40, 56), np.uint8)):
        for pipe in ("pass_through", "pass_through->lz4", "pass_through->bitswap1->lz4", "bitswap1->pass_through->lz4",
                     "diff3x3x1->pass_through->bitshuffle->lz4"):
            if vol.dtype == np.uint8 and pipe.startswith("diff"):
                continue
            _rt(sqy, oracle, pipe, vol)


# ---- frame_shuffle(frame_chunk_size=N): N frames per sort unit ----
@pytest.mark.parametrize("fcs", [2, 4, 8, 32])
def test_frame_shuffle_chunks_of_frames(sqy, oracle, fcs):
    rng = np.random.default_rng(fcs)
    vols = [synth.stack((32, 64, 128), np.uint16), synth.stack((64, 128, 128), np.uint8),
            (rng.integers(0, 4000, (32, 1, 1), dtype=np.uint16) * np.ones((32, 48, 40), np.uint16)),           # units ordered by their level
            rng.integers(0, 65536, (32, 20, 52), dtype=np.uint16)]
    for vol in vols:
        vol = np.ascontiguousarray(vol)
        for pipe in ("frame_shuffle(frame_chunk_size=%d)->lz4" % fcs, "frame_shuffle(frame_chunk_size=%d)->bitswap1->lz4" % fcs,
                     "frame_shuffle(frame_chunk_size=%d)" % fcs):
            if vol.dtype == np.uint8 and "bitswap1" in pipe:
                continue
            for nthreads in (2, 1):
                # (extra room: the reorder_map grows the header past SQY_Pipeline_Max_Compressed_Length on small stacks)
                rc, blob = sqy.encode(pipe, vol, nthreads=nthreads, extra_capacity=1 << 16)
                assert rc == 0, (pipe, vol.shape)
                want = oracle.pipeline_encode(pipe, vol, nthreads=nthreads)
                assert blob == want, (pipe, vol.shape, vol.dtype, nthreads)
                rc, back = sqy.decode(blob)
                assert rc == 0
                assert np.array_equal(back, oracle.pipeline_decode(blob))


def test_frame_shuffle_chunk_that_does_not_divide_is_refused(sqy):
    vol = synth.stack((30, 32, 32), np.uint16)
    assert sqy.pipeline_possible("frame_shuffle(frame_chunk_size=4)->lz4", np.uint16)      # (depends on the shape: decided at encode time)
    rc, _ = sqy.encode("frame_shuffle(frame_chunk_size=4)->lz4", vol)
    assert rc == 1
    assert not sqy.pipeline_possible("frame_shuffle(frame_chunk_size=0)->lz4", np.uint16)
