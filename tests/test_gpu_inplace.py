"""GPU parity of the "frames in place" path (SQYAMD_PipelineEncode_UI16_DeviceAt and the host-pointer entry points on top of
it): a 16-bit bitswap1 in front of lz4 writes the plane stream as the bodies of its future LZ4 frames, the stored frames that
end the payload stay where they are, the frames in front are gathered up against them.  Blob bytes against the oracle."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(11)
    shape = (32, 128, 128)                                          # 1 MiB: four 256 KiB chunks, every chunk spans four bit planes
    yield "synth", "bitswap1->lz4", synth.stack(shape, np.uint16)                               # compressed head, stored tail
    yield "synth_big", "bitswap1->lz4", synth.stack((64, 256, 512), np.uint16)                  # 16 MiB, 64 chunks, duplicate chunks
    yield "zeros", "bitswap1->lz4", np.zeros(shape, np.uint16)                                  # nothing stored: everything is gathered
    yield "random", "bitswap1->lz4", rng.integers(0, 65536, shape, dtype=np.uint16)             # everything stored: nothing moves
    yield "stored_in_front", "bitswap1->lz4", (rng.integers(0, 2, (64, 256, 256)) * 0x8000).astype(np.uint16)   # plane 15 noise, the rest zero
    mixed = rng.integers(0, 2, (64, 256, 256)).astype(np.uint16) * 0x8000                       # stored, compressed, stored, compressed, stored
    mixed |= (rng.integers(0, 2, (64, 256, 256)).astype(np.uint16) << 7) | rng.integers(0, 4, (64, 256, 256)).astype(np.uint16)
    yield "stored_compressed_alternating", "bitswap1->lz4", mixed
    # holes: all-zero 1 KiB pieces of the plane stream are never written by the transpose, the LZ4 stage fills them in
    holes = rng.integers(0, 65536, (64, 256, 256), dtype=np.uint16)
    holes.reshape(-1, 8192)[100::256] = 0                                                          # stored chunks with zero pieces inside
    yield "holes_in_stored_chunks", "bitswap1->lz4", holes
    twice = holes.copy()
    twice[32:] = twice[:32]                                                                      # second chunk of every plane = the first: duplicates of stored chunks
    yield "holes_in_duplicates_of_stored_chunks", "bitswap1->lz4", twice
    low = (rng.integers(0, 256, (64, 256, 256)).astype(np.uint16))                               # planes 15..8 zero, the rest noise with gaps
    low.reshape(-1, 8192)[1::3] = 0
    low.reshape(-1, 8192)[5::11] = 3
    yield "holes_low_planes", "bitswap1->lz4", low
    yield "holes_diff", "diff3x3x1->bitswap1->lz4", low.reshape(64, 128, 512)
    yield "diff", "diff3x3x1->bitswap1->lz4", synth.stack((32, 128, 256), np.uint16)      # (diff writes only the 128 columns it can touch)
    yield "diff_wide_rows", "diff3x3x1->bitswap1->lz4", rng.integers(0, 65536, (20, 32, 1024), dtype=np.uint16)   # side buffer 1/8 of the rows, sums wrap
    yield "diff_deep", "diff3x3x1->bitswap1->lz4", synth.stack((200, 16, 256), np.uint16)       # hx = 198: two of the rows' two lanes come from the side buffer -> none left out
    yield "diff_two_lanes", "diff3x3x1->bitswap1->lz4", synth.stack((130, 16, 512), np.uint16)  # hx = 128: 129 columns -> 256 of 512
    # a head filter in front of diff3x3x1 -> bitswap1: the diff's side columns live outside the ping/pong rotation (round-3 advice:
    # the transpose's output used to land on the buffer its input lived in)
    yield "raster_diff_small", "raster_reorder->diff3x3x1->bitswap1->lz4", synth.stack((16, 32, 256), np.uint16)          # <= one chunk
    yield "raster_diff", "raster_reorder->diff3x3x1->bitswap1->lz4", synth.stack((32, 128, 256), np.uint16)
    yield "bitswap_diff_bitswap", "bitswap1->diff3x3x1->bitswap1->lz4", synth.stack((32, 128, 256), np.uint16)
    yield "frame_shuffle_diff", "frame_shuffle->diff3x3x1->bitswap1->lz4", synth.stack((32, 128, 256), np.uint16)
    yield "diff_bitswap_twice_small", "diff3x3x1->bitswap1->bitswap1->lz4", synth.stack((16, 32, 256), np.uint16)        # the second transpose reads its plain input
    yield "diff_bitswap_twice", "diff3x3x1->bitswap1->bitswap1->lz4", synth.stack((32, 128, 256), np.uint16)
    yield "small_chunks", "bitswap1->lz4(blocksize_kb=64,framestep_kb=64)", synth.stack(shape, np.uint16)
    yield "ragged_last_chunk", "bitswap1->lz4", synth.stack((33, 64, 128), np.uint16)           # 528 KiB: two chunks and a bit
    yield "not_in_place_odd_tiles", "bitswap1->lz4", synth.stack((3, 50, 70), np.uint16)        # no whole tiles: the ordinary path, offset 0
    yield "not_in_place_serial", "bitswap1->lz4", synth.stack(shape, np.uint16)                 # nthreads = 1 (below): one linked frame


@pytest.mark.parametrize("name,pipeline,vol", list(_cases()), ids=[c[0] for c in _cases()])
def test_blob_in_place_equals_oracle(sqy, oracle, name, pipeline, vol):
    import torch
    dev = torch.device("cuda", 0)
    nthreads = 1 if name.endswith("serial") else 2
    want = oracle.pipeline_encode(pipeline, vol, nthreads=nthreads)
    d_vol = torch.from_numpy(vol.copy()).to(dev)
    cap = sqy.max_compressed_length(pipeline, vol.shape, vol.dtype)
    out = torch.full((cap,), 0xA5, dtype=torch.uint8, device=dev)
    rc, off, n = sqy.encode_device_at(pipeline, d_vol.data_ptr(), vol.shape, vol.dtype, out.data_ptr(), cap, nthreads=nthreads)
    assert rc == 0
    assert 0 <= off and off + n <= cap
    got = bytes(out[off:off + n].cpu().numpy().tobytes())
    assert len(got) == len(want)
    assert got == want
    if name.startswith("not_in_place"):
        assert off == 0
    elif name not in ("zeros", "raster_diff_small", "diff_bitswap_twice_small", "diff_bitswap_twice"):
        assert off > 0
    # the plain device entry point (blob at the start of the destination) and the host-pointer one give the same bytes
    rc, n2 = sqy.encode_device(pipeline, d_vol.data_ptr(), vol.shape, vol.dtype, out.data_ptr(), cap, nthreads=nthreads)
    assert rc == 0 and bytes(out[:n2].cpu().numpy().tobytes()) == want
    rc, blob = sqy.encode(pipeline, vol, nthreads=nthreads)
    assert rc == 0 and blob == want
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol)


def test_in_place_needs_room(sqy, oracle):
    """a destination that holds the blob but not the frames in place: the ordinary path is taken, same bytes"""
    import torch
    dev = torch.device("cuda", 0)
    vol = synth.stack((32, 128, 128), np.uint16)
    want = oracle.pipeline_encode("bitswap1->lz4", vol)
    d_vol = torch.from_numpy(vol.copy()).to(dev)
    cap = len(want) + 64
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, off, n = sqy.encode_device_at("bitswap1->lz4", d_vol.data_ptr(), vol.shape, vol.dtype, out.data_ptr(), cap)
    assert rc == 0 and off == 0 and bytes(out[:n].cpu().numpy().tobytes()) == want


# ---- the noise digest (round 6) --------------------------------------------------------------------------------------------------
# The transpose leaves, for every probe of a search that starts with a chunk and finds nothing (probe 961 on), the bucket and tag of the five
# bytes there; the chunk's parse takes its batches from that digest as long as nothing has matched (sqy_kernels.hip: bitswap1_u16_regs,
# lz4_chunks_kernel).  Offered when every plane segment is a whole number of chunks.  Plane streams made by hand -- noise with repeats planted
# where the digest's bookkeeping has its edges -- turned back into voxels with the oracle's inverse transpose, encoded in place, compared
# with the oracle; the same with the digest switched off; and a call on other data in between (the digest buffer is the context's: what a
# former call left there must never be read).
def _digest_plane_streams(chunk, nchunks_per_plane):
    rng = np.random.default_rng(chunk + nchunks_per_plane)
    seg = chunk * nchunks_per_plane                                   # bytes per bit plane
    n = 16 * seg
    x = rng.integers(0, 256, n, dtype=np.uint8)
    # probe positions of the empty search inside a chunk (liblz4's step schedule)
    pos, p, st, nb = [], 1, 1, 64
    while True:
        pos.append(p)
        p2 = p + st; st = nb >> 6; nb += 1
        if p2 > chunk - 12 + 1:
            break
        p = p2
    pos = np.array(pos)
    late = pos[960:]                                                  # probes 961 ..: the digest's
    k = 0
    for c in range(0, n // chunk):
        base = c * chunk
        kind = c % 8
        if kind == 0:
            continue                                                   # pure noise: the whole chunk from the digest
        if kind == 1:                                                  # a repeat that starts exactly ON a late probe, its source 100..60000 bytes back
            q = int(late[(7 * c) % len(late)])
            d = int(rng.integers(100, min(q, 60000)))
            x[base + q:base + q + 12] = x[base + q - d:base + q - d + 12]
        elif kind == 2:                                                # .. one byte BEHIND a late probe (the probe itself finds nothing), and one in front
            q = int(late[(11 * c) % len(late)])
            x[base + q + 1:base + q + 9] = x[base + q + 1 - 3000:base + q + 9 - 3000]
            q2 = int(late[(13 * c + 5) % len(late)])
            x[base + q2 - 1:base + q2 + 7] = x[base + q2 - 1 - 77:base + q2 + 7 - 77]
        elif kind == 3:                                                # two late probes with the same five bytes (the second finds the first)
            i = (17 * c) % (len(late) - 40)
            a, b = int(late[i]), int(late[i + 30])
            x[base + b:base + b + 8] = x[base + a:base + a + 8]
        elif kind == 4:                                                # probes whose five bytes straddle a 1 KiB piece: every late probe in a piece's last 4 bytes
            for q in late[(late % 1024) >= 1020][:6]:
                q = int(q)
                x[base + q:base + q + 8] = x[base + q - 2048:base + q - 2048 + 8]
        elif kind == 5:                                                # an all-zero 1 KiB piece inside the noise (a hole: this chunk's digest is not used)
            pc = 20 + (c % 200)
            x[base + pc * 1024:base + (pc + 1) * 1024] = 0
        elif kind == 6:                                                # a match early in the chunk (in front of probe 961): the digest is never used
            x[base + 2000:base + 2040] = x[base + 1000:base + 1040]
        else:                                                          # the chunk's last bytes repeat (the search's tail, behind the last whole batch)
            x[base + chunk - 40:base + chunk - 8] = x[base + chunk - 3000:base + chunk - 3000 + 32]
        k += 1
    return x


@pytest.mark.parametrize("cfg,chunk", [("", 256 << 10), ("(blocksize_kb=64,framestep_kb=64)", 64 << 10)])
def test_noise_digest(sqy, oracle, options, cfg, chunk):
    import torch
    dev = torch.device("cuda", 0)
    pipe = "bitswap1->lz4" + cfg
    planes = _digest_plane_streams(chunk, 2)
    vol = oracle.bitswap1_decode(planes.view(np.uint16)).reshape(1, 1, -1)           # voxels whose bit planes are that stream
    assert np.array_equal(np.ascontiguousarray(oracle.bitswap1_encode(vol.reshape(-1))).view(np.uint8), planes)
    want = oracle.pipeline_encode(pipe, vol, nthreads=2)
    other = np.random.default_rng(3).integers(0, 4096, vol.shape, dtype=np.uint16)  # planes 15..12 zero: holes where `vol` has noise
    want_other = oracle.pipeline_encode(pipe, other, nthreads=2)
    cap = sqy.max_compressed_length(pipe, vol.shape, np.uint16)
    out = torch.full((cap,), 0x5A, dtype=torch.uint8, device=dev)
    d_vol, d_other = torch.from_numpy(vol.copy()).to(dev), torch.from_numpy(other).to(dev)
    for digest in (1, 0, 1):
        options("noise_digest", digest)
        for d, w in ((d_vol, want), (d_other, want_other), (d_vol, want)):           # the same context, other data in between
            rc, off, n = sqy.encode_device_at(pipe, d.data_ptr(), vol.shape, np.uint16, out.data_ptr(), cap, nthreads=2)
            assert rc == 0 and off > 0
            got = bytes(out[off:off + n].cpu().numpy().tobytes())
            assert got == w, "digest %d: blob differs from the oracle's" % digest
    rc, back = sqy.decode(want)
    assert rc == 0 and np.array_equal(back, vol)
