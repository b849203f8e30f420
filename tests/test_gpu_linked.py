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


# ---- block-linked frames parsed block-parallel (round 4) -------------------------------------------------------------------
# Frames of three blocks and more are not walked by one wavefront any more: every block is parsed from a table rebuilt by a warm-up
# over the >= 64 KiB in front of it, the tables are verified against what the block in front really left and the blocks that fail
# are parsed again in order (sqy_kernels.h: Lz4SpecArgs).  The result has to be liblz4's whatever the guesses were worth: the knob
# SQY_BLOCK_PARALLEL_WARMUP makes them worthless (0: every block starts from an empty table and is parsed again) or leaves them at
# the default; `histmatch` data breaks the default guess for chosen blocks (a block of noise that begins with a match into the block
# in front of it: liblz4's skip counter restarts there, the warm-up, which cannot see that far back, probes other positions from then on).
def _histmatch(n, every, seed=11):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, n, dtype=np.uint8)
    B = 256 << 10
    for k, i in enumerate(range(B, n - 400, B)):
        if k % every == 0:
            d[i:i + 300] = d[i - 1000:i - 700]
    return d


def _profile_names(sqy):
    return set(sqy.profile_get().keys())


@pytest.mark.parametrize("warmup", [None, "0", "1", "200000", "1000000"])
@pytest.mark.parametrize("name", ["zeros", "random", "noise3", "sparse", "periodic", "farrep", "runs", "rawmix", "hist1", "hist3", "planes"])
def test_block_parallel_serial_layout(sqy, oracle, options, name, warmup):
    n = 24 * (256 << 10) + 54321
    if name == "hist1":
        data = _histmatch(n, 1)
    elif name == "hist3":
        data = _histmatch(n, 3)
    elif name == "planes":
        data = np.ascontiguousarray(oracle.bitswap1_encode_planes(synth.stack((12, 512, 512)).reshape(-1))).view(np.uint8).reshape(-1)
    else:
        data = dict(_streams(n, seed=21))[name]
    vol = data.reshape(1, 1, -1)
    want = oracle.pipeline_encode("lz4", vol, nthreads=1)
    if warmup is not None:
        options("block_parallel_warmup", int(warmup))
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, blob = sqy.encode("lz4", vol, nthreads=1)
    sqy.profile_enable(False)
    assert rc == 0
    assert blob == want, (name, warmup, len(blob), len(want))
    names = _profile_names(sqy)
    assert "lz4_linked_blocks" in names and "lz4_linked_verify" in names, names         # the block-parallel path ran
    if warmup == "0" or (name in ("hist1", "hist3") and warmup in (None, "1")):
        assert "lz4_linked_redo" in names, "the guess was expected to fail somewhere"  # .. and so did the second parse
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), data)
    # the frame walk of rounds 2-3 gives the same bytes
    options("block_parallel", 0)
    rc, blob2 = sqy.encode("lz4", vol, nthreads=1)
    assert rc == 0 and blob2 == want


@pytest.mark.parametrize("cfg", ["blocksize_kb=64", "blocksize_kb=64,framestep_kb=448", "framestep_kb=1024", "blocksize_kb=1024,framestep_kb=3072",
                                 "blocksize_kb=4096", "n_chunks_of_input=3", "n_chunks_of_input=1", "accel=-3"])
@pytest.mark.parametrize("nthreads", [1, 2])
def test_block_parallel_other_block_layouts(sqy, oracle, cfg, nthreads):
    """block sizes of 64 KiB (the warm-up is exactly liblz4's reach), update sizes that are no multiple of the block size (short blocks in
    the middle of a frame: the warm-up spans several), frames of many blocks in the chunked layout, liblz4's acceleration"""
    n = 9 * (1 << 20) + 7777
    for name in ("sparse", "periodic", "farrep", "noise3", "rawmix", "hist"):
        data = _histmatch(n, 2, seed=5) if name == "hist" else dict(_streams(n, seed=9))[name]
        vol = data.reshape(1, 1, -1)
        pipe = "lz4(%s)" % cfg
        rc, blob = sqy.encode(pipe, vol, nthreads=nthreads)
        assert rc == 0, (cfg, name)
        want = oracle.pipeline_encode(pipe, vol, nthreads=nthreads)
        assert blob == want, (cfg, name, nthreads, len(blob), len(want))
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back.reshape(-1), data), (cfg, name)


@pytest.mark.parametrize("cfg", ["", "blocksize_kb=64"])
def test_given_up_blocks_in_linked_frames(sqy, oracle, options, capfd, cfg):
    """Regression for round 5's GPU memory fault at address 0 (gpurun_out/r05_sc1.log, r05_t1.log; DESIGN.md 3 "The fault of round 5"):
    a wavefront that GUESSES a block's table (block-parallel parse, mode 1) and meets a stream of short sequences gives the block up --
    `redo_dense` -- and the first build of that rule still appended such a block to the chunked layout's dense list, which block-linked
    launches do not have (null).  Here: one frame of linked blocks in which blocks that are given up (a match every few bytes: > 2048
    short matches per block) are followed by good ones and by more of their kind, so that the verify / parse-again rounds see a given-up
    block in front of a good one, runs of several blocks, and a second round whose run list is shorter than the first's."""
    rng = np.random.default_rng(77)
    B = (64 << 10) if "64" in cfg else (256 << 10)
    kinds = "gdgddggdgdddgg"                                            # g = good, d = dense (given up while guessing)
    parts = []
    word = rng.integers(0, 256, 3000, dtype=np.uint8)
    for i, k in enumerate(kinds):
        if k == "d":
            parts.append(rng.integers(0, 2, B, dtype=np.uint8))        # 0 / 1 bytes: a short match every few bytes
        else:
            p = np.tile(word, B // word.size + 1)[:B].copy()
            p[::997] ^= (i + 1)
            parts.append(p)
    data = np.concatenate(parts + [rng.integers(0, 2, 12345, dtype=np.uint8)])
    vol = data.reshape(1, 1, -1)
    pipe = "lz4(%s)" % cfg if cfg else "lz4"
    want = oracle.pipeline_encode(pipe, vol, nthreads=1)
    options("block_parallel_stats", 1)
    capfd.readouterr()
    for _ in range(2):                                                  # (the second call reuses the workspace: stale tables and lists in it)
        sqy.profile_reset(); sqy.profile_enable(True)
        rc, blob = sqy.encode(pipe, vol, nthreads=1)
        sqy.profile_enable(False)
        assert rc == 0
        assert blob == want
        assert "lz4_linked_redo" in _profile_names(sqy), "no block was parsed again: the given-up path did not run"
    err = capfd.readouterr().err
    rounds = [ln for ln in err.splitlines() if "lz4 block-parallel: round" in ln]
    assert rounds, err[-500:]
    nruns = [int(ln.split(" runs")[0].split()[-1]) for ln in rounds]
    first = nruns[:len(nruns) // 2]                                     # the first call's rounds (both calls print the same)
    # blocks were given up and parsed again (round 0 lists them), every later round's list is no longer than the one before, the last is empty
    assert first[0] >= 1 and first[-1] == 0 and all(a >= b for a, b in zip(first, first[1:])), first
    nbad = [int(ln.split(" of ")[0].split()[-1]) for ln in rounds][:len(first)]
    assert nbad[0] >= kinds.count("d"), (nbad, "every dense block is given up while its table is guessed")
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), data)
    # the frame walk gives the same bytes
    options("block_parallel", 0)
    rc, blob2 = sqy.encode(pipe, vol, nthreads=1)
    assert rc == 0 and blob2 == want
