"""Round 5: sequence-heavy chunks parsed by several wavefronts at once (sqy_kernels.h: Lz4SegArgs; include/sqeazy_amd.h: option
"chunk_split").  On the frames-in-place path (a 16-bit bitswap1 in front of a chunked lz4) the key kernel nominates sparse chunks, four
wavefronts parse each from guessed tables, lz4_seg_verify_kernel accepts the chunk when every hand-over is the same anchor on equivalent
tables and sends it to the one-piece parse otherwise.  Whatever the guesses are worth, the bytes must be liblz4's: every case against
the oracle, with the option at 2 (nominated chunks, always), 3 (EVERY chunk that is not all zero: noise, low-entropy planes and
diff3x3x1 residuals fail the check and take the fall-back) and 0 (off); decode restores the stack."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _stacks():
    rng = np.random.default_rng(31)
    yield "synth", synth.stack((64, 512, 512), np.uint16)                                   # sparse planes 12 / 11: the guesses hold
    yield "synth+40000", (synth.stack((32, 512, 512), np.uint16).astype(np.uint32) + 40000).astype(np.uint16)
    yield "noise", rng.integers(0, 65536, (16, 512, 512), dtype=np.uint16)                  # nothing compresses
    yield "lowent", (rng.random((32, 512, 512)) < 0.05).astype(np.uint16) * 257             # sparse random bits: the parse never converges
    a = np.zeros((32, 512, 512), np.uint16)
    a[:, 100:140, :] = rng.integers(0, 4096, (32, 40, 512), dtype=np.uint16)
    a[:, 300:302, 17:400] = 65535
    yield "bands", a                                                                        # chunks with all-zero pieces and data
    b = np.zeros((16, 512, 512), np.uint16)
    b.reshape(-1)[::4099] = 1
    yield "dots", b                                                                         # matches that span segment borders, long zero runs


@pytest.mark.parametrize("mode", [2, 3, 0])
@pytest.mark.parametrize("pipeline", ["bitswap1->lz4", "diff3x3x1->bitswap1->lz4"])
def test_split_chunks_are_liblz4s_bytes(sqy, oracle, options, pipeline, mode):
    import torch
    dev = torch.device("cuda", 0)
    options("chunk_split", mode)
    for name, vol in _stacks():
        want = oracle.pipeline_encode(pipeline, vol)
        d_vol = torch.from_numpy(vol.view(np.int16)).to(dev)
        cap = sqy.max_compressed_length(pipeline, vol.shape, np.uint16)
        out = torch.full((cap,), 0xA5, dtype=torch.uint8, device=dev)                        # (holes: a destination full of garbage)
        sqy.profile_reset(); sqy.profile_enable(True)
        rc, off, n = sqy.encode_device_at(pipeline, d_vol.data_ptr(), vol.shape, np.uint16, out.data_ptr(), cap)
        sqy.profile_enable(False)
        names = set(sqy.profile_get().keys())
        assert rc == 0, (name, mode)
        blob = out[off:off + n].cpu().numpy().tobytes()
        assert blob == want, (pipeline, name, mode, len(blob), len(want))
        assert ("lz4_seg_verify" in names) == (mode != 0), (name, mode, names)
        if mode == 3 and name in ("noise", "lowent"):
            assert "lz4_chunks_dense" in names, (name, names)                                # the guesses failed: parsed again in one piece
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, vol), (pipeline, name, mode)


def test_split_through_the_host_abi_and_in_flight(sqy, oracle, options):
    """the host-pointer entry point takes the same path; with other calls in flight the default (1) leaves the chunks whole, forced (2)
    splits them in every call -- four threads, the same bytes"""
    import threading
    options("chunk_split", 2)
    vol = synth.stack((48, 512, 512), np.uint16)
    want = oracle.pipeline_encode("bitswap1->lz4", vol)
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=0)
    assert rc == 0 and blob == want
    got = [None] * 4

    def work(i):
        got[i] = sqy.encode("bitswap1->lz4", vol, nthreads=0)
    for mode in (2, 1):
        options("chunk_split", mode)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert all(g[0] == 0 and g[1] == want for g in got), mode


def test_split_headline_stack_equals_the_reference_digest(sqy, options):
    """the 1 GiB bench stack with its heavy chunks split: the blob's digest is the one the reference pieces give (tests/golden/headline.json)"""
    import hashlib
    import json
    import os
    import torch
    dev = torch.device("cuda", 0)
    options("chunk_split", 2)
    shape = (512, 1024, 1024)
    vol = synth.stack_torch(shape, np.uint16, dev)
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    sqy.profile_reset(); sqy.profile_enable(True)
    rc, off, n = sqy.encode_device_at("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap)
    sqy.profile_enable(False)
    assert rc == 0 and "lz4_seg_verify" in sqy.profile_get()
    dig = hashlib.sha256(out[off:off + n].cpu().numpy().tobytes()).hexdigest()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "headline.json")) as f:
        g = [x for x in json.load(f)["stacks"] if tuple(x["shape_zyx"]) == shape and x["z_offset"] == 0 and x["z_total"] == shape[0]][0]
    assert dig == g["blob_sha256"]
