"""GPU, BASELINE.json sizes: every config's pipeline on an HBM-resident synthetic stack through the device C-ABI,
byte-compared with the CPU oracle on the same voxels, plus size-independent properties (decode round trip,
frame structure, header fields).  One C-ABI call stays below 2^31 voxels (the reference's int voxel count),
so the 2048^3 / 2048^2x1024 configs are exercised as the 2048x2048x256 slabs a rank encodes."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _encode_on_device(sqy, pipeline, shape, dtype, extra=0):
    import torch
    dev = torch.device("cuda", 0)
    vol = synth.stack_torch(shape, dtype, dev)
    cap = sqy.max_compressed_length(pipeline, shape, dtype) + extra
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, n = sqy.encode_device(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap, nthreads=0)
    assert rc == 0
    host_vol = vol.cpu().numpy()
    blob = out[:n].cpu().numpy().tobytes()
    del vol, out
    torch.cuda.empty_cache()
    return host_vol, blob


def _check(oracle, pipeline, vol, blob, lossless=True):
    want = oracle.pipeline_encode(pipeline, vol)
    h = oracle.header_unpack(blob)
    assert h["shape"] == vol.shape and h["bytes"] == len(blob) - h["size"]
    assert len(blob) == len(want)
    same = blob == want
    assert same, "HIP blob differs from the oracle blob"
    if lossless:
        back = oracle.pipeline_decode(blob)
        if "frame_shuffle" in pipeline:
            # frames with EQUAL float metrics all map to the first of them (std::find in the reference): the stage is
            # only invertible for the frames that occur in the map
            _, dmap = oracle.frame_shuffle_encode(vol)
            keep = np.unique(dmap.astype(np.int64))
            ok = np.array_equal(back[keep], vol[keep])
        else:
            ok = np.array_equal(back, vol)
        assert ok, "decode(encode(volume)) differs from the volume"


def test_config2_1024x1024x512_u16_bitswap1_lz4(sqy, oracle):
    shape = (512, 1024, 1024)
    vol, blob = _encode_on_device(sqy, "bitswap1->lz4", shape, np.uint16)
    _check(oracle, "bitswap1->lz4", vol, blob)
    # frame structure: 4096 independent single-block frames
    h = oracle.header_unpack(blob)
    body = np.frombuffer(blob, np.uint8)[h["size"]:]
    off, frames = 0, 0
    while off < body.size:
        assert body[off:off + 7].tobytes() == bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x50, 0x77])
        size = int.from_bytes(body[off + 7:off + 11].tobytes(), "little") & 0x7fffffff
        off += 11 + size
        assert body[off:off + 4].tobytes() == b"\0\0\0\0"
        off += 4
        frames += 1
    assert frames == 4096


def test_config3_slab_2048x2048x256_u16_diff_bitswap1_lz4(sqy, oracle):
    shape = (256, 2048, 2048)
    vol, blob = _encode_on_device(sqy, "diff3x3x1->bitswap1->lz4", shape, np.uint16)
    _check(oracle, "diff3x3x1->bitswap1->lz4", vol, blob)


def test_config4_1024cubed_u8_frame_shuffle_lz4(sqy, oracle):
    shape = (1024, 1024, 1024)
    vol, blob = _encode_on_device(sqy, "frame_shuffle->lz4", shape, np.uint8, extra=1 << 16)
    _check(oracle, "frame_shuffle->lz4", vol, blob)


def test_config5_slab_2048x2048x256_u16_quantiser_bitswap1_lz4(sqy, oracle):
    shape = (256, 2048, 2048)
    vol, blob = _encode_on_device(sqy, "quantiser->bitswap1->lz4", shape, np.uint16)
    _check(oracle, "quantiser->bitswap1->lz4", vol, blob, lossless=False)
    # tolerance of the lossy stage, as north_star asks: the decoded value is the centre of mass of the voxel's
    # bucket, so the error is bounded by the widest bucket; checked here as |error| <= max bucket span
    back = oracle.pipeline_decode(blob)
    err = np.abs(back.astype(np.int32) - vol.astype(np.int32))
    enc, dec = oracle.quantiser_build_luts(oracle.histogram(vol))
    lo = np.full(256, 65535, np.int64); hi = np.zeros(256, np.int64)
    used = np.nonzero(oracle.histogram(vol))[0]
    np.minimum.at(lo, enc[used], used); np.maximum.at(hi, enc[used], used)
    assert err.max() <= int((hi - lo).max())


@pytest.mark.parametrize("pipeline", ["diff3x3x1->bitswap1->lz4", "quantiser->bitswap1->lz4"])
def test_serial_layout_of_the_c3_and_c5_slabs(sqy, oracle, pipeline):
    """nthreads = 1 -- what the HDF5 filter (sqy_h5_filter.c) and the sqy tool pass by default, and every unchanged caller of the
    reference -- at slab size on the other two BASELINE pipelines: ONE block-linked frame (lz4_utils.hpp:99-173, lz4.hpp:227-234),
    the blob byte for byte the oracle's, and the decode of it.  These are the streams on which the block-parallel parse's table
    guesses fail: a few dozen short runs on the diff3x3x1 planes, the whole sequence-heavy top plane of the quantised stack as ONE
    run of 511 blocks, which is parsed in order by the kernel with the dense batches (seconds, not milliseconds: DESIGN.md 3)."""
    import torch
    dev = torch.device("cuda", 0)
    shape = (256, 2048, 2048)
    vol = synth.stack_torch(shape, np.uint16, dev)
    cap = sqy.max_compressed_length(pipeline, shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, n = sqy.encode_device(pipeline, vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap, nthreads=1)
    assert rc == 0
    host_vol = vol.cpu().numpy()
    blob = out[:n].cpu().numpy().tobytes()
    del vol, out
    torch.cuda.empty_cache()
    want = oracle.pipeline_encode(pipeline, host_vol, nthreads=1)
    assert len(blob) == len(want) and blob == want, "HIP blob (serial layout) differs from the oracle blob"
    h = oracle.header_unpack(blob)
    body = np.frombuffer(blob, np.uint8)[h["size"]:]
    assert body[:7].tobytes() == bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x50, 0x77]) and body[-4:].tobytes() == b"\0\0\0\0"     # one frame
    rc, back = sqy.decode(blob, nthreads=1)
    assert rc == 0
    if "quantiser" in pipeline:
        assert np.array_equal(back, oracle.pipeline_decode(blob))
    else:
        assert np.array_equal(back, host_vol)


def test_config1_256cubed_u16_host_abi(sqy, oracle):
    """configs[0] (the reference's CPU-runnable case) through the HOST-pointer ABI, nthreads = all cores"""
    vol = synth.stack((256, 256, 256))
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=0)
    assert rc == 0
    _check(oracle, "bitswap1->lz4", vol, blob)
    # nthreads = 1, what the reference's own C-ABI / Java tests and CLI pass: ONE block-linked frame of 128 blocks
    rc, blob1 = sqy.encode("bitswap1->lz4", vol, nthreads=1)
    assert rc == 0
    assert blob1 == oracle.pipeline_encode("bitswap1->lz4", vol, nthreads=1)
    rc, back = sqy.decode(blob1)
    assert rc == 0 and np.array_equal(back, vol)


@pytest.mark.parametrize("pipeline", ["diff3x3x1->bitswap1->lz4", "quantiser->bitswap1->lz4"])
def test_every_slab_of_the_sharded_volumes(sqy, oracle, pipeline):
    """configs[2] / configs[4] are encoded as z-slabs of 256 frames, one per rank: EVERY slab index (not only slab 0) against
    the oracle, at the full z geometry (2048 frames, 8 slabs; the shell sweeps through them) and a reduced 256 x 256 plane"""
    import torch
    dev = torch.device("cuda", 0)
    Y = X = 256
    for slab in range(8):
        vol = synth.stack_torch((256, Y, X), np.uint16, dev, z_offset=256 * slab, z_total=2048)
        cap = sqy.max_compressed_length(pipeline, (256, Y, X), np.uint16)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        rc, n = sqy.encode_device(pipeline, vol.data_ptr(), (256, Y, X), np.uint16, out.data_ptr(), cap, nthreads=0)
        assert rc == 0, slab
        host = vol.cpu().numpy()
        assert np.array_equal(host, synth.stack((2048, Y, X))[256 * slab:256 * (slab + 1)]) if slab == 3 else True
        blob = out[:n].cpu().numpy().tobytes()
        assert blob == oracle.pipeline_encode(pipeline, host), "slab %d differs from the oracle" % slab
        rc, back = sqy.decode(blob)
        assert rc == 0
        if pipeline.startswith("diff"):
            assert np.array_equal(back, host), slab
        else:
            assert np.array_equal(back, oracle.pipeline_decode(blob)), slab


# ---- the headline pinned to the reference itself --------------------------------------------------
# tests/golden/headline.json holds the payload digests the REFERENCE pieces produce for the BASELINE headline stacks (reference SSE
# bit-plane gather + liblz4 1.9.3 frames, oracle/gen_golden.py --headline, build container only).  The HIP path's payload must
# equal them through every device entry point -- including the one bench.py times, SQYAMD_PipelineEncode_UI16_DeviceAt with
# four calls in flight.
def _headline(name_prefix):
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "headline.json")) as f:
        H = json.load(f)
    return next(s for s in H["stacks"] if s["name"].startswith(name_prefix))


def _sha(b):
    import hashlib
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("prefix", ["C1", "C2", "north_star slab 0"])
def test_headline_payload_equals_the_reference_digest(sqy, oracle, prefix):
    import torch
    g = _headline(prefix)
    shape = tuple(g["shape_zyx"])
    dev = torch.device("cuda", 0)
    vol = synth.stack_torch(shape, np.uint16, dev, z_offset=g["z_offset"], z_total=g["z_total"])
    if prefix == "C1":
        assert _sha(vol.cpu().numpy().tobytes()) == g["voxels_sha256"]          # same voxels as the generator's
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.full((cap,), 0x5A, dtype=torch.uint8, device=dev)
    # blob at the start of the destination
    rc, n = sqy.encode_device("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap)
    assert rc == 0 and n == g["blob_bytes"]
    blob = out[:n].cpu().numpy().tobytes()
    hs = sqy.header_size(blob[:65536])
    assert hs == g["header_bytes"] and n - hs == g["payload_bytes"]
    assert _sha(blob[hs:]) == g["payload_sha256"], "_Device payload differs from the reference pieces' payload"
    assert _sha(blob) == g["blob_sha256"]
    del blob
    # frames in place (what bench.py calls)
    out.fill_(0xA5)
    rc, off, n = sqy.encode_device_at("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap)
    assert rc == 0 and n == g["blob_bytes"] and off > 0
    blob = out[off:off + n].cpu().numpy().tobytes()
    assert _sha(blob[hs:]) == g["payload_sha256"], "_DeviceAt payload differs from the reference pieces' payload"
    assert _sha(blob) == g["blob_sha256"]
    del vol, out
    torch.cuda.empty_cache()


def test_headline_four_calls_in_flight_equal_the_reference_digest(sqy):
    """exactly what bench.py times: four host threads, a stream and an output buffer each, SQYAMD_PipelineEncode_UI16_DeviceAt on the
    1 GiB stack, several rounds back to back; every blob of every thread hashed"""
    import threading
    import torch
    g = _headline("C2")
    shape = tuple(g["shape_zyx"])
    dev = torch.device("cuda", 0)
    vol = synth.stack_torch(shape, np.uint16, dev)
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    T, rounds = 4, 3
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    outs = [torch.full((cap,), 0xA5, dtype=torch.uint8, device=dev) for _ in range(T)]
    torch.cuda.synchronize()
    results, errors = [[] for _ in range(T)], []

    def caller(t):
        try:
            torch.cuda.set_device(0)
            for r in range(rounds):
                rc, off, n = sqy.encode_device_at("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, outs[t].data_ptr(), cap,
                                                  stream=streams[t].cuda_stream)
                assert rc == 0
                if r == rounds - 1 or t == r:                                  # the last blob of every thread, and one from the middle
                    results[t].append(_sha(outs[t][off:off + n].cpu().numpy().tobytes()))
        except Exception as e:   # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert not errors, errors
    for t in range(T):
        assert results[t] and all(h == g["blob_sha256"] for h in results[t]), "thread %d produced a blob that differs from the reference digest" % t


@pytest.mark.parametrize("prefix", ["C1", "C2", "north_star slab 0"])
def test_headline_serial_layout_equals_the_reference_digest(sqy, prefix):
    """nthreads = 1, what the reference's own callers pass: ONE block-linked frame (lz4_utils.hpp:99-173).  Round 4 parses its blocks
    block-parallel from verified table guesses; the blob must equal what the reference SSE gather + liblz4's LZ4F_compressUpdate
    sequence give for the headline stacks (tests/golden/headline.json "serial", oracle/gen_golden.py --headline-serial)"""
    import torch
    g = _headline(prefix)
    shape = tuple(g["shape_zyx"])
    dev = torch.device("cuda", 0)
    vol = synth.stack_torch(shape, np.uint16, dev, z_offset=g["z_offset"], z_total=g["z_total"])
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.full((cap,), 0x5A, dtype=torch.uint8, device=dev)
    for _ in range(2):                                                         # (the second call reuses the workspace: stale tables in it)
        rc, n = sqy.encode_device("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap, nthreads=1)
        assert rc == 0 and n == g["serial"]["blob_bytes"]
        blob = out[:n].cpu().numpy().tobytes()
        hs = g["serial"]["header_bytes"]
        assert _sha(blob[hs:]) == g["serial"]["payload_sha256"], "serial-layout payload differs from the reference pieces' payload"
        assert _sha(blob) == g["serial"]["blob_sha256"]
    rc, back = sqy.decode(blob)
    assert rc == 0 and np.array_equal(back, vol.cpu().numpy())
    del vol, out
    torch.cuda.empty_cache()


# ---- every full-size slab of the sharded volumes ---------------------------------------------------------------
# tests/golden/headline_slabs.json (oracle/gen_golden.py --headline-slabs, build container only): all eight 2048 x 2048 x 256 slabs of
# north_star's 2048^3 'bitswap1->lz4' volume from the REFERENCE pieces (SSE bit-plane gather + liblz4 1.9.3), and the slabs around the
# shell's centre of configs[2] / configs[4] from the oracle.  The shell sweeps through z: slabs 3 and 4 hold 12-bit shell voxels over whole
# frames, data slab 0 never sees.  Encoded HERE with ONE call for the whole volume (SQYAMD_PipelineEncode_Slabs_UI16_Device, what bench.py's
# north_star leg times), so that leg's "blobs equal the single calls" is no longer the only check of slabs 1..7.
def _slab_goldens(pipeline, z_total):
    import json, os
    path = os.path.join(os.path.dirname(__file__), "golden", "headline_slabs.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        H = json.load(f)
    return {e["slab"]: e for e in H["slabs"] if e["pipeline"] == pipeline and e["z_total"] == z_total}


@pytest.mark.parametrize("pipeline,z_total,want_slabs", [("bitswap1->lz4", 2048, tuple(range(8))), ("diff3x3x1->bitswap1->lz4", 2048, (3, 4)),
                                                          ("quantiser->bitswap1->lz4", 1024, (1, 2))])
def test_full_size_slabs_equal_the_golden_digests(sqy, pipeline, z_total, want_slabs):
    import torch
    G = _slab_goldens(pipeline, z_total)
    missing = [s for s in want_slabs if s not in G]
    assert not missing, "tests/golden/headline_slabs.json lacks slabs %s of %s" % (missing, pipeline)
    dev = torch.device("cuda", 0)
    shape = (z_total, 2048, 2048)
    nslabs = z_total // 256
    vol = torch.empty(shape, dtype=torch.uint16, device=dev)
    for s in range(nslabs):                                                     # built slab by slab (the generator's temporaries are 8x a slab)
        vol[256 * s:256 * (s + 1)] = synth.stack_torch((256, 2048, 2048), np.uint16, dev, z_offset=256 * s, z_total=z_total)
    cap = sqy.max_compressed_length(pipeline, (256, 2048, 2048), np.uint16)
    out = torch.full((cap * nslabs,), 0x5A, dtype=torch.uint8, device=dev)
    rc, offs, lens = sqy.encode_slabs_device(pipeline, vol.data_ptr(), shape, np.uint16, nslabs, out.data_ptr(), cap, inflight=3)
    assert rc == 0
    for s in want_slabs:
        g = G[s]
        if s == want_slabs[0]:
            assert _sha(vol[256 * s:256 * (s + 1)].cpu().numpy().tobytes()) == g["voxels_sha256"]      # same voxels as the generator's
        assert lens[s] == g["blob_bytes"], "slab %d: %d bytes, golden %d" % (s, lens[s], g["blob_bytes"])
        blob = out[offs[s]:offs[s] + lens[s]].cpu().numpy().tobytes()
        hs = g["header_bytes"]
        assert sqy.header_size(blob[:65536]) == hs
        assert _sha(blob[hs:]) == g["payload_sha256"], "slab %d of %s: payload differs from %s" % (s, pipeline, g["source"])
        assert _sha(blob) == g["blob_sha256"]
        del blob
    del vol, out
    torch.cuda.empty_cache()
