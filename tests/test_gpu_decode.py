"""GPU: SQY_Decode_UI8/UI16 through the C-ABI -- round trips of every supported pipeline and blobs made elsewhere."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu

PIPES_U16 = ["bitswap1->lz4", "lz4", "bitswap1", "diff3x3x1->bitswap1->lz4", "diff3x3x1->lz4", "frame_shuffle->lz4",
             "frame_shuffle->bitswap1->lz4", "quantiser->bitswap1->lz4", "quantiser->lz4", "quantiser"]


def _vols_u16():
    rng = np.random.default_rng(21)
    return [synth.stack((24, 64, 96)), rng.integers(0, 65536, (9, 20, 33), dtype=np.uint16), np.zeros((5, 200, 300), np.uint16),
            (np.arange(40 * 64 * 64) % 3000).astype(np.uint16).reshape(40, 64, 64)]


@pytest.mark.parametrize("pipeline", PIPES_U16)
def test_roundtrip_u16(sqy, oracle, pipeline):
    for vol in _vols_u16():
        rc, blob = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=16 * vol.shape[0] + 256)
        assert rc == 0
        rc, back = sqy.decode(blob)
        assert rc == 0, pipeline
        want = oracle.pipeline_decode(blob)            # == vol for the lossless pipelines
        ok = np.array_equal(back, want) if "frame_shuffle" not in pipeline else True
        assert ok, (pipeline, vol.shape)
        if "frame_shuffle" in pipeline:
            _, dmap = oracle.frame_shuffle_encode(vol)
            keep = np.unique(dmap.astype(np.int64))
            assert np.array_equal(back[keep], vol[keep])
        elif "quantiser" not in pipeline:
            assert np.array_equal(back, vol)


@pytest.mark.parametrize("pipeline", ["bitswap1->lz4", "lz4", "frame_shuffle->lz4", "diff3x3x1->lz4"])
def test_roundtrip_u8(sqy, oracle, pipeline):
    for vol in (synth.stack((20, 30, 40), np.uint8), np.random.default_rng(4).integers(0, 256, (7, 9, 11), dtype=np.uint8)):
        rc, blob = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=16 * vol.shape[0] + 256)
        assert rc == 0
        rc, back = sqy.decode(blob)
        assert rc == 0
        if "frame_shuffle" in pipeline:
            _, dmap = oracle.frame_shuffle_encode(vol)
            keep = np.unique(dmap.astype(np.int64))
            assert np.array_equal(back[keep], vol[keep])
        else:
            assert np.array_equal(back, vol)


def test_decode_long_matches_and_overlaps(sqy, oracle):
    """matches far longer than the 64 KiB history ring, offsets 1..70000-ish, literals-only blocks"""
    rng = np.random.default_rng(8)
    n = 3 * (256 << 10) + 999
    streams = [np.zeros(n, np.uint8), np.tile(np.arange(7, dtype=np.uint8), n // 7 + 1)[:n],
               np.tile(rng.integers(0, 256, 70001, dtype=np.uint8), n // 70001 + 1)[:n],
               np.repeat(rng.integers(0, 256, n // 300 + 1, dtype=np.uint8), 300)[:n], rng.integers(0, 256, n, dtype=np.uint8)]
    for d in streams:
        vol = d.reshape(1, 1, -1)
        blob = oracle.pipeline_encode("lz4", vol)
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, vol)


def _frame_with_far_matches(rng, n, kind):
    if kind == 0:                                              # one long period beyond 16 KiB: match length > offset, source behind the ring
        p = int(rng.integers(16385, 60000))
        return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]
    if kind == 1:                                              # the same with 1 % noise: many separate far matches
        p = int(rng.integers(16400, 17500))
        a = np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n].copy()
        m = rng.random(n) < 0.01
        a[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
        return a
    if kind == 2:
        return rng.integers(0, 256, n, dtype=np.uint8)         # stays raw
    if kind == 3:
        a = np.zeros(n, np.uint8)
        idx = rng.integers(0, n, n // 40)
        a[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
        return a
    if kind == 4:                                              # slices copied from 16 KiB .. 64 KiB back, literals in between
        a = rng.integers(0, 256, n, dtype=np.uint8)
        pos = 40000 if n > 80000 else n // 3
        while pos < n - 4000:
            ln = int(rng.integers(4, 3000))
            back = int(rng.integers(16384, min(pos, 65535) + 1)) if pos > 16384 else pos
            a[pos:pos + ln] = a[pos - back:pos - back + ln]
            pos += ln + int(rng.integers(0, 40))
        return a
    if kind == 6:                                              # a period between the 8 KiB and the 16 KiB ring: behind one, inside the other
        p = int(rng.integers(8200, 16300))
        return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]
    p = int(rng.integers(1, 300))
    return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]


@pytest.mark.parametrize("pipeline,frame_bytes,nframes", [("lz4", 256 << 10, 800), ("lz4(blocksize_kb=64)", 64 << 10, 1100),
                                                         ("lz4(blocksize_kb=64,framestep_kb=256)", 256 << 10, 780),
                                                         ("lz4(blocksize_kb=64,framestep_kb=64)", 64 << 10, 3000)])     # (> 2560 compressed frames: the 8 KiB ring)
def test_decode_many_frames_small_ring(sqy, oracle, pipeline, frame_bytes, nframes):
    """more than 768 compressed frames select the 16 KiB-ring decode kernel: matches that reach further back than the ring
    (up to 64 KiB, also across the blocks of a multi-block frame, also longer than their offset) come from the output buffer"""
    rng = np.random.default_rng(nframes)
    d = np.concatenate([_frame_with_far_matches(rng, frame_bytes, f % 8) for f in range(nframes)] + [rng.integers(0, 256, 777, dtype=np.uint8)])
    vol = d.reshape(1, 1, -1)
    blob = oracle.pipeline_encode(pipeline, vol)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)
    rc, mine = sqy.encode(pipeline, vol, nthreads=2)
    assert rc == 0 and mine == blob


def test_decode_serial_layout_from_liblz4(sqy, oracle):
    """a blob in the reference's nthreads=1 layout (ONE block-linked frame, made with liblz4 itself) decodes too"""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not available")
    vol = synth.stack((20, 128, 128))
    planes = oracle.bitswap1_encode(vol).reshape(-1).view(np.uint8)
    payload = ref.lz4_encode_serial(planes).tobytes()
    name = "bitswap1(num_bits_per_plane=1)->lz4(accel=1,blocksize_kb=256,framestep_kb=256,n_chunks_of_input=0)"
    blob = oracle.header_pack(np.uint16, vol.shape, name, len(payload)) + payload
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)


def test_decode_rejects_garbage(sqy, oracle):
    vol = synth.stack((4, 16, 16))
    blob = bytearray(oracle.pipeline_encode("bitswap1->lz4", vol))
    rc, _ = sqy.decode(bytes(blob[:len(blob) // 2]))
    assert rc != 0
    h = oracle.header_unpack(bytes(blob))
    blob[h["size"]] ^= 0xff                                  # break the first frame's magic
    rc, _ = sqy.decode(bytes(blob))
    assert rc != 0


def test_decode_payload_full_of_frame_magics(sqy, oracle):
    """stored (incompressible) chunks whose bytes imitate LZ4 frame headers, some of them right behind four zero bytes:
    the parallel frame ranking collects them as candidates and must still find exactly the real frames"""
    rng = np.random.default_rng(21)
    n = 5 * (256 << 10) + 12345
    d = rng.integers(0, 256, n, dtype=np.uint8)
    fake = np.frombuffer(bytes([0, 0, 0, 0, 0x04, 0x22, 0x4D, 0x18, 0x40, 0x50, 0x77, 0x10, 0x00, 0x00, 0x00]), np.uint8)
    for p in rng.integers(0, n - 64, 4000):
        d[p:p + fake.size] = fake
    d[:fake.size - 4] = fake[4:]                              # one at the very start of the first chunk, too
    vol = d.reshape(1, 1, -1)
    blob = oracle.pipeline_encode("lz4", vol)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)
    # the same through the device's own encoder
    rc, blob2 = sqy.encode("lz4", vol, nthreads=2)
    assert rc == 0 and blob2 == blob


def test_decode_truncated_and_damaged_chunked_streams(sqy, oracle):
    """damage in the middle of a many-frame stream: the ranking gives up, the serial walk reports the error"""
    vol = synth.stack((24, 256, 256))
    blob = bytearray(oracle.pipeline_encode("bitswap1->lz4", vol))
    h = oracle.header_unpack(bytes(blob))
    assert sqy.decode(bytes(blob))[0] == 0
    cut = bytes(blob[:len(blob) - 7])                         # the last frame loses its end
    assert sqy.decode(cut)[0] != 0
    mid = h["size"] + (len(blob) - h["size"]) // 2
    bad = bytearray(blob)
    bad[mid:mid + 64] = bytes(64)
    rc, back = sqy.decode(bytes(bad))
    assert rc != 0 or not np.array_equal(back, vol)           # either refused or visibly different, never a crash


def _blob(oracle, dtype, shape, pipename, payload):
    return oracle.header_pack(dtype, shape, pipename, len(payload)) + payload


LZ4_NAME = "lz4(accel=1,blocksize_kb=256,framestep_kb=256,n_chunks_of_input=0)"
EMPTY_FRAME = bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x50, 0x77, 0, 0, 0, 0])


def test_decode_flood_of_empty_frames(sqy, oracle):
    """a payload of nothing but 11-byte block-less frames must be refused, not walk the frame index off its buffer"""
    payload = EMPTY_FRAME * 200000                            # 2.2 MB, far more frames than the volume can have blocks
    blob = _blob(oracle, np.uint8, (1, 1, 1000), LZ4_NAME, payload)
    rc, _ = sqy.decode(blob)
    assert rc == 11                                           # the sink failed: code + 10 (dynamic_pipeline.hpp:822)


def test_decode_header_arithmetic_is_checked(sqy, oracle):
    """extents whose product wraps to something small, zero extents, absurd ranks, payload sizes beyond the blob"""
    good = oracle.pipeline_encode("lz4", np.zeros((1, 1, 64), np.uint8))
    h = oracle.header_unpack(good)
    payload = good[h["size"]:]
    for shape in ((1 << 33, 1 << 31, 1), (0, 4, 4), (1 << 31, 1, 1), (70000, 70000, 1)):
        blob = _blob(oracle, np.uint8, shape, LZ4_NAME, payload)
        out = np.zeros(64, np.uint8)
        src = np.frombuffer(blob, np.uint8)
        import ctypes
        rc = sqy.lib().SQY_Decode_UI8(src.ctypes.data, ctypes.c_long(len(blob)), out.ctypes.data, ctypes.c_int(1))
        assert rc == 1, shape
    # payload_bytes larger than what is there
    hdr = oracle.header_pack(np.uint8, (1, 1, 64), LZ4_NAME, len(payload) + 1000)
    assert sqy.decode(hdr + payload)[0] == 1


def test_decode_short_and_missing_frames(sqy, oracle):
    """frames that decode to less than their share, or no frame at all, are errors -- never a silently half-written volume"""
    vol = synth.stack((8, 64, 64), np.uint8)                  # 32 KiB: one frame
    short = oracle.pipeline_encode("lz4", vol[:4])
    hs = oracle.header_unpack(short)
    blob = _blob(oracle, np.uint8, vol.shape, LZ4_NAME, short[hs["size"]:])
    assert sqy.decode(blob)[0] == 11
    blob = _blob(oracle, np.uint8, vol.shape, LZ4_NAME, b"")
    assert sqy.decode(blob)[0] != 0
    blob = _blob(oracle, np.uint8, vol.shape, LZ4_NAME, EMPTY_FRAME)
    assert sqy.decode(blob)[0] == 11
    # chunked layout where one frame is short: 3 chunks, the middle one re-encoded from fewer bytes
    big = synth.stack((3, 512, 512), np.uint8)                # 3 x 256 KiB
    frames = [oracle.lz4_encode_chunked(big[i].reshape(-1)) for i in range(3)]
    frames[1] = oracle.lz4_encode_chunked(big[1].reshape(-1)[:1000])
    blob = _blob(oracle, np.uint8, big.shape, LZ4_NAME, b"".join(f.tobytes() for f in frames))
    assert sqy.decode(blob)[0] == 11


def test_decode_composite_return_codes(sqy, oracle):
    """dynamic_pipeline.hpp:795-846: sink failure = code + 10; with lz4 as a tail filter behind the quantiser sink its failure
    is the plain code"""
    vol = synth.stack((8, 64, 64))
    blob = bytearray(oracle.pipeline_encode("bitswap1->lz4", vol))
    h = oracle.header_unpack(bytes(blob))
    blob[h["size"] + 1] ^= 0xff
    assert sqy.decode(bytes(blob))[0] == 11
    blob = bytearray(oracle.pipeline_encode("quantiser->lz4", vol))
    h = oracle.header_unpack(bytes(blob))
    blob[h["size"] + 1] ^= 0xff
    assert sqy.decode(bytes(blob))[0] == 1
    assert sqy.decode(b"no header here at all, just text" * 10)[0] == 1


@pytest.mark.parametrize("shape,dtype", [((2, 3, 5), np.uint16), ((5, 3, 2), np.uint16), ((2, 3, 2), np.uint8), ((40, 5, 4), np.uint16),
                                         ((30, 4, 8), np.uint8), ((12, 6, 3), np.uint16), ((70, 7, 16), np.uint16), ((9, 3, 3), np.uint8),
                                         ((33, 9, 15), np.uint16), ((18, 4, 17), np.uint8)])
def test_diff_decode_rows_that_spill(sqy, oracle, shape, dtype):
    """geometries in which a row's rewritten span runs over the row end (Z - 2 > X - 1, or the single-row case): the last voxels
    of a frame then depend on the first voxels of the SAME decoded frame"""
    if dtype == np.uint8 and max(shape) > 100:
        pytest.skip("8-bit diff: small extents only")
    rng = np.random.default_rng(sum(shape))
    for trial in range(3):
        vol = rng.integers(0, 65536 if dtype == np.uint16 else 256, shape).astype(dtype)
        try:
            blob = oracle.pipeline_encode("diff3x3x1->lz4", vol)
        except ValueError:
            pytest.skip("shape outside what the reference defines for diff3x3x1")
        assert np.array_equal(oracle.pipeline_decode(blob), vol)
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, vol), (shape, trial)
        rc, mine = sqy.encode("diff3x3x1->lz4", vol, nthreads=2)
        assert rc == 0 and mine == blob


@pytest.mark.parametrize("shape,dtype", [((9, 300, 64), np.uint16), ((64, 17, 64), np.uint16), ((3, 700, 128), np.uint16), ((20, 3, 64), np.uint16),
                                         ((16, 4, 128), np.uint16), ((5, 600, 8), np.uint16), ((24, 300, 24), np.uint16), ((31, 513, 192), np.uint16), ((50, 40, 2048), np.uint16), ((128, 3, 128), np.uint16),
                                         ((2, 64, 64), np.uint16), ((7, 2100, 64), np.uint16), ((64, 256, 64), np.uint16), ((33, 1030, 256), np.uint16)])
def test_diff_decode_one_launch_kernel(sqy, oracle, shape, dtype):
    """16-bit, Z <= X, X a multiple of 8: the strip kernel (one launch, neighbour strips hand their edge rows over)"""
    rng = np.random.default_rng(sum(shape))
    vol = rng.integers(0, 65536 if dtype == np.uint16 else 256, shape).astype(dtype)
    blob = oracle.pipeline_encode("diff3x3x1->bitswap1->lz4", vol)
    for _ in range(2):
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, vol), shape


def _medium_runs(rng, n):
    """short literal bursts between runs of 8..300 bytes with periods 1..70: matches of 4..300 bytes at small offsets,
    i.e. output bytes that depend on bytes produced a few lanes earlier in the same 64-byte step of the batch decoder"""
    out = np.empty(n + 400, np.uint8)
    pos = 0
    while pos < n:
        lit = int(rng.integers(0, 12))
        out[pos:pos + lit] = rng.integers(0, 256, lit, dtype=np.uint8)
        pos += lit
        p = int(rng.integers(1, 71))
        run = int(rng.integers(8, 300))
        pat = rng.integers(0, 256, p, dtype=np.uint8)
        out[pos:pos + run] = np.tile(pat, run // p + 1)[:run]
        pos += run
    return out[:n]


@pytest.mark.parametrize("seed,pipeline", [(1, "lz4"), (2, "lz4"), (3, "lz4(blocksize_kb=64)"), (4, "lz4(blocksize_kb=64,framestep_kb=256)")])
def test_decode_batches_with_dependencies_inside_a_step(sqy, oracle, seed, pipeline):
    rng = np.random.default_rng(seed)
    n = (3 << 20) + 12345 if seed < 3 else 900 * (64 << 10) + 77          # the second kind: > 768 frames -> the 16 KiB-ring kernel
    d = _medium_runs(rng, n)
    vol = d.reshape(1, 1, -1)
    blob = oracle.pipeline_encode(pipeline, vol)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)
    if seed == 1:                                                          # and the serial layout: one wave, 13 linked blocks
        blob1 = oracle.pipeline_encode(pipeline, vol, nthreads=1)
        rc, back = sqy.decode(blob1)
        assert rc == 0 and np.array_equal(back, vol)


@pytest.mark.parametrize("shape", [(2, 5000, 64), (3, 300, 4096), (5, 40, 8192), (64, 2049, 64), (17, 1023, 520), (8, 257, 8), (100, 33, 104)])
def test_diff_decode_one_launch_kernel_strip_geometries(sqy, oracle, shape):
    """strips of 1..20 rows, the last one short, rows from 16 bytes to 8 KiB; X = 8192 exceeds the exchange words a thread
    takes and uses the per-frame kernels"""
    rng = np.random.default_rng(sum(shape))
    vol = rng.integers(0, 65536, shape).astype(np.uint16)
    blob = oracle.pipeline_encode("diff3x3x1->lz4", vol)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol), shape


def test_frame_shuffle_decode_with_equal_metrics(sqy, oracle):
    """all-zero frames around the data: the encoder's map names frame 0 for every one of them (std::find), the frames nobody
    names come out as zeros -- like the oracle's, not as workspace bytes"""
    rng = np.random.default_rng(5)
    vol = np.zeros((34, 95, 354), np.uint16)
    vol[4:9:2] = rng.integers(0, 4000, (3, 95, 354))
    vol[29:] = rng.integers(0, 60000, (5, 95, 354))
    blob = oracle.pipeline_encode("frame_shuffle->lz4", vol)
    # (dirty the decode workspace first: a volume of the same size full of ones)
    rc, _ = sqy.decode(oracle.pipeline_encode("frame_shuffle->lz4", np.full(vol.shape, 0xFFFF, np.uint16)))
    assert rc == 0
    rc, back = sqy.decode(blob)
    assert rc == 0
    assert np.array_equal(back, oracle.pipeline_decode(blob))
    assert np.array_equal(back, vol)


# ---- frame_shuffle's inverse folded into the LZ4 decode (round 5) ----------------------------------------------------------------
# When every LZ4 chunk lies inside ONE frame of the shuffle the frames are decoded straight to their places (sqy_kernels.hip:
# lz4_decode_frame_out); any other geometry takes the two stages one after the other.  Frames of 1, 2, 4 chunks, a chunk of several
# frames and frames no chunk size divides (the fall-back), compressible, stored and mixed frames, maps that name a frame twice.
@pytest.mark.parametrize("dtype,shape,cfg", [(np.uint8, (12, 512, 512), ""), (np.uint8, (9, 1024, 512), ""), (np.uint16, (7, 512, 256), ""),
                                             (np.uint16, (6, 1024, 512), ""), (np.uint8, (40, 128, 128), ""), (np.uint8, (11, 300, 301), ""),
                                             (np.uint8, (16, 256, 256), "(framestep_kb=64)"), (np.uint16, (5, 512, 512), "(framestep_kb=128)"),
                                             (np.uint8, (6, 512, 512), "(framestep_kb=512)"), (np.uint8, (7, 512, 512), "(n_chunks_of_input=7)")])
@pytest.mark.parametrize("kind", ["stack", "noise", "mixed", "sparse", "equal"])
def test_frame_shuffle_inverse_folded_into_lz4(sqy, oracle, dtype, shape, cfg, kind):
    rng = np.random.default_rng(shape[0] * 7 + shape[2])
    hi = 256 if dtype == np.uint8 else 65536
    if kind == "stack":
        vol = synth.stack(shape, dtype)
    elif kind == "noise":
        vol = rng.integers(0, hi, shape).astype(dtype)
    elif kind == "mixed":
        vol = synth.stack(shape, dtype)
        vol[1::3] = rng.integers(0, hi, vol[1::3].shape).astype(dtype)
    elif kind == "sparse":                                  # compressed frames (the others are mostly stored)
        vol = ((rng.random(shape) < 0.01) * rng.integers(1, hi, shape)).astype(dtype)
        vol[::4] = (np.arange(vol[::4].size) % 251).reshape(vol[::4].shape).astype(dtype)
    else:                                                   # frames with equal metrics: the map names one of them for all
        vol = rng.integers(0, hi, shape).astype(dtype)
        vol[2] = vol[0]
        vol[3:5] = 0
        vol[shape[0] - 1] = 0
    pipeline = "frame_shuffle->lz4" + cfg
    blob = oracle.pipeline_encode(pipeline, vol, nthreads=2)
    want = oracle.pipeline_decode(blob)
    rc, _ = sqy.decode(oracle.pipeline_encode(pipeline, np.full(shape, hi - 1, dtype), nthreads=2))      # (dirty the workspace)
    assert rc == 0
    rc, back = sqy.decode(blob)
    assert rc == 0
    assert np.array_equal(back, want), (shape, cfg, kind)
    if kind != "equal":
        assert np.array_equal(back, vol)
    rc, mine = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=16 * shape[0] + 4096)
    assert rc == 0 and mine == blob


def test_frame_shuffle_inverse_folded_bad_map_is_refused(sqy, oracle):
    """a reorder_map in the header that names a frame outside the volume: refused before anything is written through it"""
    import base64, re
    vol = np.random.default_rng(3).integers(0, 256, (8, 512, 512), dtype=np.uint8)
    blob = oracle.pipeline_encode("frame_shuffle->lz4", vol, nthreads=2)
    m = re.search(rb"<verbatim>([A-Za-z0-9+/=]+)<\\/verbatim>", blob)
    assert m, "reorder_map not found in the header"
    raw = bytearray(base64.b64decode(m.group(1)))
    assert len(raw) == 8 * 8
    for slot in (0, 3, 7):
        r2 = bytearray(raw)
        r2[8 * slot:8 * slot + 8] = (8).to_bytes(8, "little")            # frame 8 of 8
        enc = base64.b64encode(bytes(r2))
        assert len(enc) == len(m.group(1))                                # (the header keeps its length)
        rc, _ = sqy.decode(blob[:m.start(1)] + enc + blob[m.end(1):])
        assert rc != 0, slot
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)


# ---- ONE block-linked frame decoded block-parallel (round 4) ---------------------------------------------------------------------
# The serial layout's frame was decoded by one wavefront in rounds 2-3.  Now every block is decoded at once with the history in front of
# it as an unknown (a 16-bit reference per byte), the last 64 KiB of every block are resolved in order, everything else at once
# (sqy_kernels.hip: lz4_blocks_decode_sym_kernel).  Streams chosen so that references are copied around inside a block, cross several
# block borders, reach 64 KiB back, come out of matches longer than their offset, and so that stored blocks sit between compressed ones.
def _linked_streams(n, seed):
    rng = np.random.default_rng(seed)
    B = 256 << 10
    yield "zeros", np.zeros(n, np.uint8)                                       # every byte of every block refers to the byte in front of it
    yield "period7", np.tile(np.arange(7, dtype=np.uint8), n // 7 + 1)[:n]
    p = np.tile(rng.integers(0, 256, 60001, dtype=np.uint8), n // 60001 + 1)[:n]
    yield "period60001", p                                                     # every match reaches ~59 KiB back: far sources, across borders
    a = np.zeros(n, np.uint8); idx = rng.integers(0, n, n // 50); a[idx] = rng.integers(1, 256, idx.size)
    yield "sparse", a
    yield "runs", np.repeat(rng.integers(0, 4, n // 64 + 1, dtype=np.uint8), 64)[:n]
    d = rng.integers(0, 256, n, dtype=np.uint8)
    for i in range(0, n, 2 * B):
        d[i:i + B] = 0
    yield "rawmix", d                                                          # stored and compressed blocks take turns
    c = rng.integers(0, 256, n, dtype=np.uint8)
    for i in range(66000, n - 300, 66000):
        c[i:i + 300] = c[i - 65000:i - 65000 + 300]
    yield "farrep", c
    e = rng.integers(0, 256, n, dtype=np.uint8)                                # a block's first bytes copied on and on inside the block:
    for b0 in range(B, n - B, B):                                              # references that travel through many matches
        e[b0:b0 + 500] = e[b0 - 700:b0 - 200]
        for j in range(1, 200):
            e[b0 + j * 1000:b0 + j * 1000 + 400] = e[b0 + (j - 1) * 1000 + 50:b0 + (j - 1) * 1000 + 450]
    yield "relay", e
    yield "planes", np.ascontiguousarray(sqy_oracle_planes(synth.stack((12, 512, 512)))).view(np.uint8).reshape(-1)[:n]
    yield "words", rng.integers(0, 256, (50, 12), dtype=np.uint8)[rng.integers(0, 50, n // 12 + 1)].reshape(-1)[:n]


def sqy_oracle_planes(vol):
    from oracle import sqy_oracle
    return sqy_oracle.bitswap1_encode_planes(vol.reshape(-1))


@pytest.mark.parametrize("cfg", ["", "blocksize_kb=64", "blocksize_kb=1024"])
@pytest.mark.parametrize("name", [s[0] for s in _linked_streams(1 << 20, 0)])
def test_serial_layout_decodes_block_parallel(sqy, oracle, options, name, cfg):
    n = 20 * (256 << 10) + 4567 if cfg != "blocksize_kb=1024" else 5 * (1 << 20) + 999
    data = dict(_linked_streams(n, 31))[name]
    vol = data.reshape(1, 1, -1)
    pipe = "lz4(%s)" % cfg if cfg else "lz4"
    blob = oracle.pipeline_encode(pipe, vol, nthreads=1)
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, back = sqy.decode(blob)
    sqy.profile_enable(False)
    names = set(sqy.profile_get().keys())
    assert rc == 0 and np.array_equal(back.reshape(-1), data), (name, cfg)
    assert "lz4_linked_decode" in names and "lz4_frames_decode" not in names, names   # the block-parallel path, no fall-back
    options("block_parallel", 0)                                  # the one-wavefront walk agrees
    rc, back2 = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back2.reshape(-1), data)


def test_serial_layout_damaged_streams_fall_back_to_the_walk(sqy, oracle, options):
    """damage inside a block, a block that does not decode to a full block, a cut stream: the block-parallel decode raises its flag and the
    one-wavefront walk gives the verdict -- the same return code and bytes as with the block-parallel path switched off"""
    rng = np.random.default_rng(3)
    n = 12 * (256 << 10) + 100
    a = np.zeros(n, np.uint8); idx = rng.integers(0, n, n // 30); a[idx] = rng.integers(1, 256, idx.size)
    vol = a.reshape(1, 1, -1)
    blob = oracle.pipeline_encode("lz4", vol, nthreads=1)
    h = oracle.header_unpack(blob)
    cases = []
    for at in (h["size"] + 200, h["size"] + (len(blob) - h["size"]) // 2, len(blob) - 300):
        b = bytearray(blob); b[at:at + 40] = bytes([0xF7] * 40); cases.append(bytes(b))
    cases.append(blob[:len(blob) - 9])
    for bad in cases:
        options("block_parallel", 1)
        rc1, back1 = sqy.decode(bad)
        options("block_parallel", 0)
        rc2, back2 = sqy.decode(bad)
        assert rc1 == rc2
        if rc1 == 0:
            assert np.array_equal(back1, back2)


@pytest.mark.parametrize("name", ["zeros", "period60001", "sparse", "rawmix", "relay", "planes", "runs"])
def test_serial_layout_long_frames_resolve_their_tails_as_a_scan(sqy, oracle, options, name):
    """frames of a few hundred blocks: the tails are not walked by one workgroup but composed per range of 64 blocks, chained, and walked by
    all ranges at once (lz4_linked_compose_tails_kernel ..); 64 KiB blocks so that a test-sized stream has 300 of them; the one walk
    (SQY_NO_TAIL_SCAN) and the one-wavefront decode (SQY_NO_BLOCK_PARALLEL) agree"""
    n = 300 * (64 << 10) + 777
    if name == "planes":
        data = np.ascontiguousarray(sqy_oracle_planes(synth.stack((40, 512, 512)))).view(np.uint8).reshape(-1)[:n]
    elif name == "relay":
        rng = np.random.default_rng(17)
        B = 64 << 10
        data = rng.integers(0, 256, n, dtype=np.uint8)
        for b0 in range(B, n - B, B):                                          # every block relays a piece of the block in front of it
            data[b0:b0 + 3000] = data[b0 - 3100:b0 - 100]
            data[b0 + B - 5000:b0 + B - 2000] = data[b0 + 10:b0 + 3010]
    else:
        data = dict(_linked_streams(n, 77))[name]
    vol = data.reshape(1, 1, -1)
    blob = oracle.pipeline_encode("lz4(blocksize_kb=64)", vol, nthreads=1)
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, back = sqy.decode(blob)
    sqy.profile_enable(False)
    names = set(sqy.profile_get().keys())
    assert rc == 0 and np.array_equal(back.reshape(-1), data), name
    assert "lz4_linked_decode" in names and "lz4_frames_decode" not in names, names
    options("tail_scan", 0)
    rc, back2 = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back2.reshape(-1), data)


def test_serial_layout_frame_cut_into_uneven_blocks_falls_back(sqy, oracle):
    """a valid block-linked frame whose blocks are NOT all full (another LZ4F writer that flushes early could produce it; sqeazy's encoders do
    not): the block-parallel decode notices (a block does not decode to one block size), the one-wavefront walk decodes it"""
    rng = np.random.default_rng(5)
    sizes = [256 << 10, (256 << 10) - 1000, 256 << 10, 256 << 10, 12345]       # (the total still looks like four full blocks and a rest)
    data = rng.integers(0, 256, sum(sizes), dtype=np.uint8)
    payload = bytearray([0x04, 0x22, 0x4D, 0x18, 0x40, 0x50, 0x77])          # magic, FLG (v1, blocks linked), BD (256 KiB), HC
    at = 0
    for sz in sizes:
        payload += int(sz | 0x80000000).to_bytes(4, "little") + data[at:at + sz].tobytes()   # stored blocks
        at += sz
    payload += bytes(4)
    blob = oracle.header_pack(np.uint8, (1, 1, data.size), LZ4_NAME, len(payload)) + bytes(payload)
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, back = sqy.decode(blob)
    sqy.profile_enable(False)
    names = set(sqy.profile_get().keys())
    assert rc == 0 and np.array_equal(back.reshape(-1), data)
    assert "lz4_linked_decode" in names and "lz4_frames_decode" in names, names      # tried, refused, walked


# ---- hand-made LZ4 sequences: offsets and lengths on every boundary the decoders' copy paths know --------------------------------
# (16 bytes per lane, 1 KiB per step, the 16 KiB ring of the small-ring kernels, 64 KiB reach, references across block borders).  The
# streams come from no encoder: a frame of linked 64 KiB blocks is written sequence by sequence, the expected bytes by a byte-wise
# reference decode in Python.  Decoded three ways: block-parallel, with one walk over the tails, by the one-wavefront walk.
def _handmade_linked_frame(seed, nblocks, block=64 << 10):
    import xxhash
    rng = np.random.default_rng(seed)
    OFFS = [1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 63, 64, 65, 255, 256, 1023, 1024, 1025, 4095, 16383, 16384, 16385, 16400, 32768, 40000, 65534, 65535]
    LENS = [4, 5, 15, 16, 17, 18, 19, 63, 64, 65, 255, 256, 257, 273, 274, 1023, 1024, 1025, 2047, 2048, 2049, 5000]
    out = bytearray()
    payload = bytearray([0x04, 0x22, 0x4D, 0x18, 0x40, 0x40])
    payload.append((xxhash.xxh32(bytes(payload[4:6]), seed=0).intdigest() >> 8) & 0xFF)
    for b in range(nblocks):
        last = b == nblocks - 1
        target = block if not last else int(rng.integers(2000, block - 100))
        start = len(out)
        comp = bytearray()
        while True:
            room = target - (len(out) - start)
            lit = int(rng.choice([0, 1, 2, 5, 14, 15, 16, 30, 269, 270, 271, 600])) if rng.random() < 0.8 else int(rng.integers(0, 40))
            if room < 64 or lit + 4 + 12 > room:                     # close the block: literals only (at least the 12 bytes LZ4 wants at a block's end)
                lit = room
                tok = min(lit, 15) << 4
                comp.append(tok)
                if lit >= 15:
                    r = lit - 15
                    comp += bytes([255] * (r // 255) + [r % 255])
                lits = rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
                comp += lits; out += lits
                break
            avail = min(len(out) + lit, 65535)                       # how far back a match may reach (the frame's own output only)
            offs = [o for o in OFFS if o <= avail]
            if not offs:
                lit = max(lit, 8); avail = min(len(out) + lit, 65535); offs = [o for o in OFFS if o <= avail]
            off = int(rng.choice(offs)) if rng.random() < 0.8 else int(rng.integers(1, avail + 1))
            maxml = room - lit - 12
            lens = [m for m in LENS if m <= maxml]
            ml = int(rng.choice(lens)) if rng.random() < 0.8 else int(rng.integers(4, min(maxml, 3000) + 1))
            tok = (min(lit, 15) << 4) | min(ml - 4, 15)
            comp.append(tok)
            if lit >= 15:
                r = lit - 15
                comp += bytes([255] * (r // 255) + [r % 255])
            lits = rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
            comp += lits; out += lits
            comp += off.to_bytes(2, "little")
            if ml - 4 >= 15:
                r = ml - 4 - 15
                comp += bytes([255] * (r // 255) + [r % 255])
            for _ in range(ml):                                      # byte-wise: overlapping matches extend themselves
                out.append(out[-off])
        assert len(out) - start == target
        payload += len(comp).to_bytes(4, "little") + comp
    payload += bytes(4)
    return bytes(payload), np.frombuffer(bytes(out), np.uint8)


@pytest.mark.parametrize("seed,nblocks", [(1, 5), (2, 9), (3, 40), (4, 300), (5, 3)])
def test_handmade_sequences_on_every_boundary(sqy, oracle, options, seed, nblocks):
    payload, want = _handmade_linked_frame(seed, nblocks)
    name = "lz4(accel=1,blocksize_kb=64,framestep_kb=256,n_chunks_of_input=0)"
    blob = oracle.header_pack(np.uint8, (1, 1, want.size), name, len(payload)) + payload
    assert np.array_equal(oracle.pipeline_decode(blob).reshape(-1), want)      # the oracle's decoder agrees with the byte-wise reference
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, back = sqy.decode(blob)
    sqy.profile_enable(False)
    names = set(sqy.profile_get().keys())
    assert rc == 0 and np.array_equal(back.reshape(-1), want), "block-parallel decode differs"
    assert "lz4_linked_decode" in names and "lz4_frames_decode" not in names, names
    options("tail_scan", 0)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), want), "one walk over the tails differs"
    options("block_parallel", 0)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), want), "one-wavefront decode differs"


@pytest.mark.parametrize("nframes", [12, 900, 2800])
def test_handmade_sequences_chunked_layout(sqy, oracle, nframes):
    """the same hand-made sequences as independent single-block frames (the chunked layout): 12 frames take the 64 KiB-ring kernel, 900 the
    16 KiB-ring one, 2800 the 8 KiB-ring one (matches behind the ring come from the output buffer); 30 distinct frames, repeated"""
    # (_handmade_linked_frame makes its LAST block short; a frame of the chunked layout must fill its chunk unless it is the stream's last:
    # of two-block frames the first block, which is full and names nothing in front of itself, becomes a frame of its own)
    full = []
    for s_ in range(30):
        payload, want = _handmade_linked_frame(200 + s_, 2, block=64 << 10)
        first_sz = int.from_bytes(payload[7:11], "little")
        full.append((payload[:11 + first_sz] + bytes(4), want[:64 << 10]))
    frames = [full[i % 30] for i in range(nframes - 1)] + [_handmade_linked_frame(100, 1, block=64 << 10)]
    payload = b"".join(f[0] for f in frames)
    want = np.concatenate([f[1] for f in frames])
    name = "lz4(accel=1,blocksize_kb=64,framestep_kb=64,n_chunks_of_input=0)"
    blob = oracle.header_pack(np.uint8, (1, 1, want.size), name, len(payload)) + payload
    assert np.array_equal(oracle.pipeline_decode(blob).reshape(-1), want)
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back.reshape(-1), want)


# ---- the chunked layout decoded by two wavefronts per frame (round 5) -----------------------------------------------------------------
# Frames of one block are decoded by lz4_frames_decode2_kernel: wave 0 parses the compressed bytes into units of up to 64 sequences, wave 1
# moves the bytes.  "decode_two_waves" = 0 sends the same blobs through the one-wavefront kernel (which also takes multi-block frames and
# the streams with more than 2560 compressed frames): both have to give the volume back, whatever the sequences look like -- streams of
# three-byte sequences (units taken 64 bytes a round), matches of a KiB and more (one after the other), chunks of zeros (a thousand length
# bytes in a row), literals of every length around the 4 that travel inside a record and the 3008 that fit the stage, matches behind the
# 16 KiB ring, the parser's stage running dry in the middle of a header, hand-made sequences on every boundary.
def _two_wave_streams(n, seed):
    rng = np.random.default_rng(seed)
    yield "zeros", np.zeros(n, np.uint8)
    yield "noise2bit", rng.integers(0, 4, n, dtype=np.uint8)                      # short sequences throughout
    a = np.zeros(n, np.uint8)
    idx = rng.integers(0, n, n // 200)
    a[idx] = rng.integers(1, 256, idx.size)
    yield "sparse", a                                                              # a few literals, long matches
    b = rng.integers(0, 256, n, dtype=np.uint8)
    pos = 5000
    while pos < n - 9000:                                                           # literal runs of 0 .. 9000 bytes between copies from near and far
        ln = int(rng.integers(4, 600))
        back = int(rng.integers(1, min(pos, 65535)))
        b[pos:pos + ln] = b[pos - back:pos - back + ln]
        pos += ln + int(rng.choice([0, 1, 2, 3, 4, 5, 14, 15, 16, 270, 3000, 3008, 3009, 4500, 9000]))
    yield "literal_runs", b
    c = np.repeat(rng.integers(0, 256, n // 700 + 1, dtype=np.uint8), 700)[:n].copy()
    c[::997] ^= 1
    yield "runs", c                                                                 # offset 1, matches of hundreds of bytes
    d = np.tile(rng.integers(0, 256, 20000, dtype=np.uint8), n // 20000 + 1)[:n].copy()
    d[rng.integers(0, n, n // 300)] ^= 0x55
    yield "period20000", d                                                          # every match behind the 16 KiB ring
    e = synth.stack((n // (64 * 64), 64, 64), np.uint8).reshape(-1)
    yield "stack8", np.resize(e, n)


@pytest.mark.parametrize("nframes,cfg", [(5, ""), (40, "(blocksize_kb=64,framestep_kb=64)"), (900, "(blocksize_kb=64,framestep_kb=64)")])
@pytest.mark.parametrize("name", [s[0] for s in _two_wave_streams(1 << 16, 0)])
def test_chunked_layout_two_wavefronts_and_one_agree(sqy, oracle, options, name, nframes, cfg):
    chunk = (64 << 10) if cfg else (256 << 10)
    n = nframes * chunk - 12345
    data = dict(_two_wave_streams(n, 77))[name]
    vol = data.reshape(1, 1, -1)
    blob = oracle.pipeline_encode("lz4" + cfg, vol, nthreads=2)
    for two in (1, 0):
        options("decode_two_waves", two)
        sqy.profile_reset(); sqy.profile_enable(True)
        rc, back = sqy.decode(blob)
        sqy.profile_enable(False)
        assert rc == 0, (name, two)
        assert np.array_equal(back.reshape(-1), data), (name, two)


def test_chunked_layout_two_wavefronts_damaged_blocks(sqy, oracle, options):
    """bytes of a compressed block overwritten (tokens, offsets, length bytes): either kernel refuses or gives different bytes, neither hangs
    nor writes outside the volume (the byte behind it stays as it was)"""
    rng = np.random.default_rng(5)
    n = 6 * (256 << 10)
    data = dict(_two_wave_streams(n, 3))["literal_runs"]
    blob = oracle.pipeline_encode("lz4", data.reshape(1, 1, -1), nthreads=2)
    h = oracle.header_unpack(blob)
    for trial in range(12):
        bad = bytearray(blob)
        at = h["size"] + 11 + int(rng.integers(0, len(blob) - h["size"] - 64))
        kind = trial % 4
        if kind == 0:
            bad[at] = 0xFF                                   # a token that promises 15+ literals and a 19+ match
        elif kind == 1:
            bad[at:at + 2] = b"\x00\x00"                     # an offset of zero, if it lands on one
        elif kind == 2:
            bad[at:at + 40] = b"\xff" * 40                   # a run of length bytes
        else:
            bad[at:at + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        for two in (1, 0):
            options("decode_two_waves", two)
            rc, back = sqy.decode(bytes(bad))
            assert rc != 0 or back.size == data.size


# ---- round 6 (round-5 advice) ---------------------------------------------------------------------------------------------------
def test_frame_shuffle_crafted_map_names_a_place_twice(sqy, oracle):
    """A reorder_map that sends two frames with DIFFERENT contents to one place (no encoder writes that: its duplicates are frames of
    equal bytes): the LZ4 frames are then not decoded straight to their places (several would decode into one at once, and the ring
    kernels read back from there) but to the stream, and the shuffle's inverse keeps the last frame named -- as the reference's loop does."""
    import base64
    rng = np.random.default_rng(9)
    for dtype, shape in ((np.uint8, (6, 512, 512)), (np.uint16, (5, 512, 256))):
        hi = 256 if dtype == np.uint8 else 4096
        vol = (rng.integers(0, hi, shape) * (rng.random(shape) < 0.3)).astype(dtype)      # compressible: matches reach far back
        blob = bytearray(oracle.pipeline_encode("frame_shuffle->lz4", vol))
        hs = oracle.header_unpack(bytes(blob))["size"]
        head = bytes(blob[:hs])
        a = head.index(b"<verbatim>") + len(b"<verbatim>")
        b = head.index(b"<\\/verbatim>")
        m = np.frombuffer(base64.b64decode(head[a:b]), dtype=np.uint64).copy()
        m[1] = m[0]; m[3] = m[0]                                                          # three frames to one place
        enc = base64.b64encode(m.tobytes())
        assert len(enc) == b - a
        blob[a:b] = enc
        want = oracle.pipeline_decode(bytes(blob))
        for _ in range(3):
            rc, back = sqy.decode(bytes(blob))
            assert rc == 0
            assert np.array_equal(back, want)


def _lz4_block_sequences(block):
    """(token position, literal length, offset position, match length) of every sequence of one LZ4 block"""
    seqs, i, n = [], 0, len(block)
    while i < n:
        tok = i
        t = block[i]; i += 1
        lit = t >> 4
        if lit == 15:
            while True:
                v = block[i]; i += 1; lit += v
                if v != 255:
                    break
        i += lit
        if i >= n:
            seqs.append((tok, lit, None, 0))
            break
        offp = i; i += 2
        ml = t & 15
        if ml == 15:
            while True:
                v = block[i]; i += 1; ml += v
                if v != 255:
                    break
        seqs.append((tok, lit, offp, ml + 4))
    return seqs


@pytest.mark.parametrize("pipeline,nframes,cfg", [("lz4", 6, ""), ("lz4", 900, "(blocksize_kb=64,framestep_kb=64)"), ("frame_shuffle->lz4", 8, "")])
def test_damaged_blocks_never_write_outside_the_volume(sqy, oracle, options, pipeline, nframes, cfg):
    """Decode through SQYAMD_Decode_UI8_Device into a buffer with canaries in front of and behind the volume; blocks damaged on purpose:
    an offset that reaches in front of the frame's first byte, a literal length that runs past the block's end, a match length that
    overflows the block -- and random bytes.  Both LZ4 decode kernels (two wavefronts per frame / one), the 64 KiB and the 16 KiB ring
    (6 and 900 compressed frames), and frames decoded straight to their places behind a frame_shuffle.  The call may refuse or return
    different voxels; it may not touch a byte outside the volume, and it must return."""
    import ctypes
    import torch
    rng = np.random.default_rng(31)
    chunk = (64 << 10) if cfg else (256 << 10)
    if pipeline == "lz4":
        n = nframes * chunk - 4321
        data = dict(_two_wave_streams(n, 13))["literal_runs"]
        vol = data.reshape(1, 1, -1)
    else:
        vol = dict(_two_wave_streams(nframes * chunk, 17))["literal_runs"].reshape(nframes, 512, 512)
        vol = vol + (np.arange(nframes, dtype=np.uint8) * 3)[:, None, None]           # distinct frame metrics: the map is a permutation
    blob = oracle.pipeline_encode(pipeline + cfg, vol, nthreads=2)
    hs = oracle.header_unpack(blob)["size"]
    # the first two compressed frames' blocks
    frames, off = [], hs
    while off < len(blob) and len(frames) < 3:
        size = int.from_bytes(blob[off + 7:off + 11], "little")
        raw, size = bool(size & 0x80000000), size & 0x7fffffff
        if not raw:
            frames.append((off + 11, size))
        off += 11 + size + 4
    assert frames
    dev = torch.device("cuda", 0)
    PAD = 1 << 16
    L = sqy.lib()
    L.SQYAMD_Decode_UI8_Device.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    d_out = torch.empty(PAD + vol.size + PAD, dtype=torch.uint8, device=dev)

    def damaged():
        for b0, sz in frames:
            seqs = _lz4_block_sequences(blob[b0:b0 + sz])
            first = next(s for s in seqs if s[2] is not None)
            mid = seqs[len(seqs) // 2]
            x = bytearray(blob); x[b0 + first[2]:b0 + first[2] + 2] = b"\xff\xff"; yield "offset in front of the frame", x
            x = bytearray(blob); x[b0 + mid[0]] |= 0xF0; x[b0 + mid[0] + 1:b0 + mid[0] + 1 + 64] = b"\xff" * 64; yield "literal length past the block", x
            if mid[2] is not None:
                x = bytearray(blob); x[b0 + mid[0]] |= 0x0F
                p = b0 + mid[2] + 2
                x[p:p + 600] = b"\xff" * 600; yield "match length overflows the block", x
            x = bytearray(blob); x[b0 + sz - 12:b0 + sz] = b"\x1f" + b"\x00" * 11; yield "a match in the last bytes", x
        for _ in range(6):
            x = bytearray(blob)
            at = hs + 11 + int(rng.integers(0, len(blob) - hs - 64))
            x[at:at + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8)); yield "random bytes", x

    for what, bad in damaged():
        d_src = torch.frombuffer(bytearray(bad), dtype=torch.uint8).to(dev)
        for two in (1, 0):
            options("decode_two_waves", two)
            d_out.fill_(0xC3)
            rc = L.SQYAMD_Decode_UI8_Device(ctypes.c_void_p(d_src.data_ptr()), ctypes.c_long(len(bad)), ctypes.c_void_p(d_out.data_ptr() + PAD),
                                            ctypes.c_long(vol.size), None)
            torch.cuda.synchronize()
            assert bool((d_out[:PAD] == 0xC3).all()) and bool((d_out[PAD + vol.size:] == 0xC3).all()), (what, two, "bytes outside the volume were written")
            if rc == 0:
                assert d_out[PAD:PAD + vol.size].numel() == vol.size


def _rank_launches(sqy):
    return sum(v[1] for k, v in sqy.profile_get().items() if k == "lz4_frame_rank")


@pytest.mark.parametrize("cfg", ["", "(blocksize_kb=64,framestep_kb=64)"])
def test_stored_tail_is_found_where_it_must_start(sqy, oracle, options, cfg):
    """round 6: the stored frames at the end of an LZ4 stream are looked for at n - (15 + last) - k * (15 + chunk) instead of by the scan
    for frame headers; the ranking still has to reach them from position 0.  Streams with a long tail, none, a compressed frame inside
    the noise, a short last frame, the first frame as part of the tail -- and one built to fool the look: a stored payload that holds a
    header, zeros in front and the end mark of ANOTHER frame behind, exactly where a tail frame would start (the ranking gives up, the
    whole stream is scanned: two launches)."""
    chunk = (64 << 10) if cfg else (256 << 10)
    rng = np.random.default_rng(61)
    noise = lambda n: rng.integers(0, 256, n, dtype=np.uint8)
    zeros = lambda n: np.zeros(n, np.uint8)
    runs = lambda n: np.repeat(rng.integers(0, 256, n // 100 + 1, dtype=np.uint8), 100)[:n]
    streams = {
        "long tail, short last": [zeros(chunk), runs(chunk)] + [noise(chunk) for _ in range(5)] + [noise(12345)],
        "all stored, whole chunks": [noise(chunk) for _ in range(6)],
        "no tail": [noise(chunk), noise(chunk), zeros(chunk)],
        "compressed inside the noise": [noise(chunk), noise(chunk), zeros(chunk), noise(chunk), noise(chunk), noise(chunk)],
        "short compressed last": [noise(chunk), noise(chunk), zeros(777)],
        "tail of one": [runs(chunk), zeros(chunk), noise(chunk)],
        "tiny last": [noise(chunk), noise(chunk), noise(1)],
    }
    for name, parts in streams.items():
        data = np.concatenate(parts)
        blob = oracle.pipeline_encode("lz4" + cfg, data.reshape(1, 1, -1), nthreads=2)
        for on in (1, 0):
            options("stored_tail_index", on)
            sqy.profile_reset(); sqy.profile_enable(True)
            rc, back = sqy.decode(blob)
            sqy.profile_enable(False)
            assert rc == 0 and np.array_equal(back.reshape(-1), data), (name, on)
            assert _rank_launches(sqy) == 1, (name, on)
    # the stream that fools the look: Z, A (noise), B (zeros: c bytes as a frame), C (noise, the last)
    parts = [noise(chunk), noise(chunk), zeros(chunk), noise(chunk - 4321)]
    data = np.concatenate(parts)
    blob = oracle.pipeline_encode("lz4" + cfg, data.reshape(1, 1, -1), nthreads=2)
    h = oracle.header_unpack(blob)
    payload = blob[h["size"]:]
    c = len(payload) - 2 * (15 + chunk) - (15 + chunk - 4321)
    assert 64 < c < chunk // 2
    fake = bytes(4) + payload[:7] + int(0x80000000 | chunk).to_bytes(4, "little")
    at = chunk + (c - 11 - 4)                                   # in A's bytes: a frame start c bytes behind A's own
    data[at:at + len(fake)] = np.frombuffer(fake, np.uint8)
    blob = oracle.pipeline_encode("lz4" + cfg, data.reshape(1, 1, -1), nthreads=2)
    payload = blob[h["size"]:]
    t2 = len(payload) - (15 + chunk - 4321) - (15 + chunk)
    assert payload[t2:t2 + 4] == bytes([0x04, 0x22, 0x4D, 0x18]) and t2 == 15 + chunk + c
    for on, launches in ((1, 2), (0, 1)):
        options("stored_tail_index", on)
        sqy.profile_reset(); sqy.profile_enable(True)
        rc, back = sqy.decode(blob)
        sqy.profile_enable(False)
        assert rc == 0 and np.array_equal(back.reshape(-1), data), on
        assert _rank_launches(sqy) == launches, on
    # bit planes: the bench stack's shape of blob (compressed head, stored tail), both ways
    vol = synth.stack((16, 256, 256))
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=2)
    assert rc == 0
    for on in (1, 0):
        options("stored_tail_index", on)
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, vol)


def test_stored_tail_random_streams(sqy, oracle, options):
    """random orders of stored and compressible chunks, random last lengths: the decode with and without the look at the stored tail"""
    chunk = 64 << 10
    cfg = "(blocksize_kb=64,framestep_kb=64)"
    rng = np.random.default_rng(62)
    for case in range(40):
        nch = int(rng.integers(2, 24))
        parts = []
        for i in range(nch):
            n = chunk if i + 1 < nch else int(rng.choice([1, 5, 15, 16, 17, 4095, chunk - 1, chunk, int(rng.integers(1, chunk + 1))]))
            kind = rng.integers(0, 4) if case % 3 else 0              # every third stream: noise only
            if kind == 0: parts.append(rng.integers(0, 256, n, dtype=np.uint8))
            elif kind == 1: parts.append(np.zeros(n, np.uint8))
            elif kind == 2: parts.append(np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), 50)[:n])
            else: parts.append(np.tile(rng.integers(0, 256, 300, dtype=np.uint8), n // 300 + 1)[:n])
        data = np.concatenate(parts)
        blob = oracle.pipeline_encode("lz4" + cfg, data.reshape(1, 1, -1), nthreads=2)
        for on in (1, 0):
            options("stored_tail_index", on)
            rc, back = sqy.decode(blob)
            assert rc == 0 and np.array_equal(back.reshape(-1), data), (case, on, nch, len(parts[-1]))
