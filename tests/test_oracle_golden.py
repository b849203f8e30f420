"""CPU: the oracle against the committed golden vectors (liblz4 1.9.3 / the reference's SSE bit-plane gather),
and -- when oracle/_ref is present -- against the real thing directly."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle.gen_golden import gen_bytes
from sqeazy_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_meta_pins_liblz4_193():
    assert G["_meta"]["liblz4"].startswith("1.9.3")
    assert G["_meta"]["LZ4F_compressBound_256k"] == 262152      # tests/test_lz4_sandbox.cpp:387-430
    assert G["_meta"]["LZ4F_HEADER_SIZE_MAX"] == 19


@pytest.mark.parametrize("case", G["lz4_block"], ids=lambda c: "%s_%d" % (c["kind"], c["n"]))
def test_lz4_block_vs_liblz4_golden(oracle, case):
    d = gen_bytes(case["kind"], case["n"], case["seed"])
    c = oracle.lz4_block_compress(d)
    if case["csize"] == 0:
        assert c is None
    else:
        assert len(c) == case["csize"] and sha(c) == case["sha256"]
        assert oracle.lz4_block_decompress(c, case["n"]) == d.tobytes()


@pytest.mark.parametrize("case", G["lz4_frames"], ids=lambda c: "%s_%d_%s" % (c["kind"], c["n"], c.get("config", "")))
def test_lz4_frames_vs_liblz4_golden(oracle, case):
    d = gen_bytes(case["kind"], case["n"], case["seed"])
    cfg = oracle.Lz4Config(case.get("config", ""))
    f = oracle.lz4_encode_chunked(d, cfg)
    assert f.size == case["bytes"] and sha(f.tobytes()) == case["sha256"]
    assert np.array_equal(oracle.lz4_decode_frames(f, case["n"]), d)


@pytest.mark.parametrize("case", G["lz4_linked"], ids=lambda c: "%s_%d_%s" % (c["kind"], c["n"], c["config"]))
def test_lz4_linked_frames_vs_liblz4_golden(oracle, case):
    """block-linked frames: the serial layout (nthreads == 1) and chunks of several LZ4 blocks, bytes from liblz4 1.9.3"""
    d = gen_bytes(case["kind"], case["n"], case["seed"])
    cfg = oracle.Lz4Config(case["config"])
    f = oracle.lz4_encode_serial(d, cfg)
    assert f.size == case["serial_bytes"] and sha(f.tobytes()) == case["serial_sha256"]
    assert np.array_equal(oracle.lz4_decode_frames(f, case["n"]), d)
    f = oracle.lz4_encode_chunked(d, cfg)
    assert f.size == case["chunked_bytes"] and sha(f.tobytes()) == case["chunked_sha256"]
    assert np.array_equal(oracle.lz4_decode_frames(f, case["n"]), d)


@pytest.mark.parametrize("case", G["bitswap1_u16"], ids=lambda c: c["name"])
def test_bitswap1_vs_reference_sse_golden(oracle, case):
    arr = {"ramp128": lambda: np.arange(128, dtype=np.uint16),
           "random_64k": lambda: np.random.default_rng(5).integers(0, 65536, 1 << 16, dtype=np.uint16),
           "synth_16x64x64": lambda: synth.stack((16, 64, 64)).reshape(-1),
           "lowbits_4096": lambda: np.random.default_rng(6).integers(0, 16, 4096, dtype=np.uint16)}[case["name"]]()
    out = oracle.bitswap1_encode(arr)
    assert sha(out.tobytes()) == case["sha256"]
    assert np.array_equal(oracle.bitswap1_encode_planes(arr, 3), out)
    assert np.array_equal(oracle.bitswap1_decode(out), arr)


VOLS = {"synth_u16_32x64x64": lambda: synth.stack((32, 64, 64)), "synth_u16_24x100x52": lambda: synth.stack((24, 100, 52)),
        "synth_u8_48x64x96": lambda: synth.stack((48, 64, 96), np.uint8)}


@pytest.mark.parametrize("case", G["pipelines"], ids=lambda c: c["volume"] + ":" + c["pipeline"])
def test_pipeline_blobs_golden(oracle, case):
    vol = VOLS[case["volume"]]()
    blob = oracle.pipeline_encode(case["pipeline"], vol)
    assert len(blob) == case["bytes"] and sha(blob) == case["sha256"]
    h = oracle.header_unpack(blob)
    if "payload_sha256" in case:
        assert sha(blob[h["size"]:]) == case["payload_sha256"]
    back = oracle.pipeline_decode(blob)
    if "quantiser" not in case["pipeline"]:
        assert np.array_equal(back, vol)
    # the layout the reference's default callers get (nthreads = 1: one block-linked frame)
    blob1 = oracle.pipeline_encode(case["pipeline"], vol, nthreads=1)
    assert len(blob1) == case["nthreads1_bytes"] and sha(blob1) == case["nthreads1_sha256"]
    if "nthreads1_payload_sha256" in case:
        assert sha(blob1[oracle.header_unpack(blob1)["size"]:]) == case["nthreads1_payload_sha256"]


def test_raw_fixtures(oracle):
    d = np.fromfile(os.path.join(GOLD, "sparse_10000.in.bin"), np.uint8)
    want = np.fromfile(os.path.join(GOLD, "sparse_10000.lz4frames.bin"), np.uint8)
    assert np.array_equal(oracle.lz4_encode_chunked(d), want)
    v = np.fromfile(os.path.join(GOLD, "synth_u16_4x16x32.in.bin"), np.uint16)
    assert np.array_equal(v, synth.stack((4, 16, 32)).reshape(-1))
    bs = np.fromfile(os.path.join(GOLD, "synth_u16_4x16x32.bitswap1.bin"), np.uint16)
    assert np.array_equal(oracle.bitswap1_encode(v), bs)
    pay = np.fromfile(os.path.join(GOLD, "synth_u16_4x16x32.bitswap1_lz4_payload.bin"), np.uint8)
    blob = oracle.pipeline_encode("bitswap1->lz4", v.reshape(4, 16, 32))
    assert blob[oracle.header_unpack(blob)["size"]:] == pay.tobytes()


# ---- directly against the real pieces when they are available (build container; prebuilt .so on the GPU box) ----
def _ref():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libsqy_ref.so not available here")
    return ref


def test_live_liblz4_and_reference_sse(oracle):
    ref = _ref()
    assert ref.lz4_version() == 10903
    rng = np.random.default_rng(123)
    for n in (777, 70001, 300000):
        for maker in (lambda: rng.integers(0, 4, n, dtype=np.uint8), lambda: rng.integers(0, 256, n, dtype=np.uint8),
                      lambda: np.repeat(rng.integers(0, 256, n // 37 + 1, dtype=np.uint8), 37)[:n]):
            d = maker()
            assert oracle.lz4_block_compress(d) == ref.lz4_block(d) or n > 262144
            assert np.array_equal(oracle.lz4_encode_chunked(d), ref.lz4_encode_parallel(d, nthreads=3))
    x = rng.integers(0, 65536, 128 * 50, dtype=np.uint16)
    assert np.array_equal(oracle.bitswap1_encode(x), ref.bitswap1_encode_u16(x, 2))
    # liblz4's decoder accepts our frames
    d = synth.stack((8, 64, 64)).reshape(-1).view(np.uint8)
    assert np.array_equal(ref.lz4_decode_frames(oracle.lz4_encode_chunked(d), d.size), d)


def test_live_liblz4_linked_blocks(oracle):
    """the oracle's model of liblz4's linked-block mode (external-dictionary / prefix modes, LZ4F's tmp buffer) against
    liblz4 itself on update sizes that are not multiples of the block size"""
    ref = _ref()
    rng = np.random.default_rng(321)
    n = 1_000_000
    streams = [gen_bytes(k, n, 9) for k in ("sparse", "farrep", "8level", "rawmix", "periodic")]
    for d in streams:
        for kb, bid in ((256, 5), (64, 4)):
            cfg = oracle.Lz4Config("blocksize_kb=%d" % kb)
            for step in (100000, 300001, 65536, 262145, 999999, 5_000_000, int(rng.integers(1000, 700000))):
                want = ref.lz4_encode_serial(d, framestep=step, block_id=bid)
                assert np.array_equal(oracle.lz4_encode_serial(d, cfg, framestep=step), want), (kb, step)


# ---- liblz4 acceleration above 1: sqeazy's lz4(accel=-k) ----
A = json.load(open(os.path.join(GOLD, "accel.json")))


@pytest.mark.parametrize("case", A["cases"], ids=lambda c: "%s_%d_accel%d" % (c["kind"], c["n"], c["accel"]))
def test_lz4_acceleration_vs_liblz4_golden(oracle, case):
    d = gen_bytes(case["kind"], case["n"], case["seed"])
    assert oracle.lz4_acceleration(case["accel"]) == case["acceleration"]
    c = oracle.lz4_block_compress(d, acceleration=case["acceleration"])
    if case["block_csize"] == 0:
        assert c is None
    else:
        assert len(c) == case["block_csize"] and sha(c) == case["block_sha256"]
    cfg = oracle.Lz4Config("accel=%d" % case["accel"])
    f = oracle.lz4_encode_chunked(d, cfg)
    assert f.size == case["chunked_bytes"] and sha(f.tobytes()) == case["chunked_sha256"]
    assert np.array_equal(oracle.lz4_decode_frames(f, case["n"]), d)
    f = oracle.lz4_encode_serial(d, cfg)
    assert f.size == case["serial_bytes"] and sha(f.tobytes()) == case["serial_sha256"]
    assert np.array_equal(oracle.lz4_decode_frames(f, case["n"]), d)
