"""GPU parity for block-linked LZ4 frames: the serial layout the reference's own callers ask for (nthreads = 1:
tests/test_pipeline_interface.cpp:218-224, SqeazyLibraryTests.java:46-53,208-215, src/sqy.cpp:190) and chunks that span
several LZ4 blocks (framestep_kb > blocksize_kb, n_chunks_of_input), through the C-ABI against the oracle, whose
restatement of liblz4's linked-block mode is pinned to liblz4 1.9.3 (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _streams(n, seed=5):
    """byte streams that exercise matches into the previous block, long literal runs, raw-stored blocks"""
    rng = np.random.default_rng(seed)
    yield "zeros", np.zeros(n, np.uint8)
    yield "random", rng.integers(0, 256, n, dtype=np.uint8)
    yield "noise1", rng.integers(0, 2, n, dtype=np.uint8)
    yield "noise3", rng.integers(0, 8, n, dtype=np.uint8)
    yield "ramp", (np.arange(n) % 251).astype(np.uint8)
    a = np.zeros(n, np.uint8)
    idx = rng.integers(0, n, n // 50)
    a[idx] = rng.integers(1, 256, idx.size)
    yield "sparse", a
    b = np.tile(rng.integers(0, 256, 7000, dtype=np.uint8), n // 7000 + 1)[:n].copy()
    b[::1531] ^= 1
    yield "periodic", b
    c = rng.integers(0, 256, n, dtype=np.uint8)
    for i in range(66000, n - 300, 66000):                # repeats just inside / outside the 64 KiB window
        c[i:i + 300] = c[i - 65000:i - 65000 + 300]
    yield "farrep", c
    yield "runs", np.repeat(rng.integers(0, 4, n // 64 + 1, dtype=np.uint8), 64)[:n]
    d = rng.integers(0, 256, n, dtype=np.uint8)           # half incompressible (stored raw), half zeros, per block
    for i in range(0, n, 2 * (256 << 10)):
        d[i:i + (256 << 10)] = 0
    yield "rawmix", d


SIZES = (300_000, 2 * (1 << 20) + 12345)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("name", [s[0] for s in _streams(16)])
def test_serial_layout_bytes(sqy, oracle, name, n):
    data = dict(_streams(n))[name]
    vol = data.reshape(1, 1, -1)
    rc, blob = sqy.encode("lz4", vol, nthreads=1)
    assert rc == 0
    want = oracle.pipeline_encode("lz4", vol, nthreads=1)
    assert len(blob) == len(want)
    assert blob == want
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), data)


@pytest.mark.parametrize("cfg", ["blocksize_kb=64", "blocksize_kb=64,framestep_kb=256", "framestep_kb=1024",
                                 "n_chunks_of_input=3", "blocksize_kb=64,n_chunks_of_input=5", "n_chunks_of_input=1"])
@pytest.mark.parametrize("nthreads", [1, 2])
def test_multi_block_frames(sqy, oracle, cfg, nthreads):
    n = 2 * (1 << 20) + 4321
    for name in ("sparse", "periodic", "farrep", "noise3", "rawmix"):
        data = dict(_streams(n, seed=8))[name]
        vol = data.reshape(1, 1, -1)
        pipe = "lz4(%s)" % cfg
        rc, blob = sqy.encode(pipe, vol, nthreads=nthreads)
        assert rc == 0, (cfg, name)
        want = oracle.pipeline_encode(pipe, vol, nthreads=nthreads)
        assert blob == want, (cfg, name, nthreads, len(blob), len(want))
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back.reshape(-1), data), (cfg, name)


@pytest.mark.parametrize("W", [64, 128, 352])
def test_java_binding_shape_nthreads_1(sqy, oracle, W):
    """SqeazyLibraryTests.java:46-53,208-215: 256 x 128 x W uint16, "bitswap1->lz4", nthreads = 1"""
    vol = synth.stack((W, 128, 256))
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=1)
    assert rc == 0
    assert blob == oracle.pipeline_encode("bitswap1->lz4", vol, nthreads=1)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)


def test_pipeline_interface_roundtrip_nthreads_1(sqy, oracle):
    """tests/test_pipeline_interface.cpp:210-238,388-416: default pipelines, nthreads = 1, encode then decode"""
    rng = np.random.default_rng(3)
    vol16 = synth.stack((40, 96, 200))
    vol8 = synth.stack((40, 96, 200), np.uint8)
    for pipe, vol in (("bitswap1->lz4", vol16), ("diff3x3x1->bitswap1->lz4", vol16), ("quantiser->lz4", vol16),
                      ("lz4", vol8), ("frame_shuffle->lz4", vol8), ("bitswap1->lz4", vol8),
                      ("lz4", rng.integers(0, 65536, (9, 100, 333), dtype=np.uint16))):
        rc, blob = sqy.encode(pipe, vol, nthreads=1)
        assert rc == 0, pipe
        assert blob == oracle.pipeline_encode(pipe, vol, nthreads=1), pipe
        rc, back = sqy.decode(blob)
        assert rc == 0, pipe
        if not pipe.startswith("quantiser"):
            assert np.array_equal(back, vol), pipe


def test_serial_layout_u16_planes_8mib(sqy, oracle):
    """32 linked blocks of real bit planes: every block boundary inside long zero runs and inside noise"""
    vol = synth.stack((16, 512, 512))
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=1)
    assert rc == 0
    assert blob == oracle.pipeline_encode("bitswap1->lz4", vol, nthreads=1)
