"""The throughput entry point for plain C callers: SQYAMD_PipelineEncode_Slabs_*_Device -- a volume as N z-slab blobs with one
call, k slab calls in flight on library-owned streams.  Every blob must equal the blob of the single call on that slab (and
the oracle's)."""
import os
import subprocess

import numpy as np
import pytest

from sqeazy_amd import synth, multi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("pipeline,shape,dtype,nslabs,inflight", [
    ("bitswap1->lz4", (64, 256, 256), np.uint16, 4, 3),
    ("bitswap1->lz4", (37, 128, 256), np.uint16, 5, 2),            # uneven split: 8, 8, 7, 7, 7 frames
    ("diff3x3x1->bitswap1->lz4", (48, 128, 128), np.uint16, 3, 3),
    ("quantiser->bitswap1->lz4", (32, 128, 128), np.uint16, 4, 4),
    ("frame_shuffle->lz4", (64, 64, 128), np.uint8, 4, 3),
    ("bitswap1->lz4", (16, 64, 64), np.uint16, 16, 8),
])
def test_slabs_call_equals_single_calls(sqy, oracle, pipeline, shape, dtype, nslabs, inflight):
    import torch
    dev = torch.device("cuda", 0)
    vol = synth.stack(shape, dtype)
    d_vol = torch.from_numpy(vol.copy()).to(dev)
    biggest = (-(-shape[0] // nslabs),) + tuple(shape[1:])
    cap = (sqy.max_compressed_length(pipeline, biggest, dtype) + 255) & ~255
    out = torch.zeros(cap * nslabs, dtype=torch.uint8, device=dev)
    rc, offs, lens = sqy.encode_slabs_device(pipeline, d_vol.data_ptr(), shape, dtype, nslabs, out.data_ptr(), cap, inflight=inflight)
    assert rc == 0
    for i in range(nslabs):
        z0, nz = multi.slab_range(shape[0], i, nslabs)
        assert i * cap <= offs[i] and offs[i] + lens[i] <= (i + 1) * cap
        got = bytes(out[offs[i]:offs[i] + lens[i]].cpu().numpy().tobytes())
        assert got == oracle.pipeline_encode(pipeline, vol[z0:z0 + nz]), (pipeline, i)


def test_slabs_call_starts_behind_the_default_stream(sqy, oracle):
    """The slab calls run on streams of the library's own; what the caller has queued on the DEFAULT stream -- here: the kernels that make
    the volume and a fill of the output buffer, behind a pile of other work -- has to be in front of them (round 6: the full-size slab
    test's fill of d_dst overtook the first slabs' transposes and was parsed in their place: 8.6 MB blobs of 0x5A instead of 1.2 GB)."""
    import torch
    dev = torch.device("cuda", 0)
    pipeline, shape, nslabs = "bitswap1->lz4", (64, 256, 256), 4
    vol = synth.stack(shape, np.uint16)
    want = [oracle.pipeline_encode(pipeline, vol[multi.slab_range(shape[0], i, nslabs)[0]:sum(multi.slab_range(shape[0], i, nslabs))]) for i in range(nslabs)]
    cap = (sqy.max_compressed_length(pipeline, (shape[0] // nslabs,) + shape[1:], np.uint16) + 255) & ~255
    h_vol = torch.from_numpy(vol.copy()).pin_memory()
    ballast = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for fill in (0x5A, 0):
        d_vol = torch.zeros(shape, dtype=torch.uint16, device=dev)
        out = torch.empty(cap * nslabs, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for k in range(8):
            ballast.fill_(k)                                   # a few milliseconds of work in front
        d_vol.copy_(h_vol, non_blocking=True)                  # the volume arrives behind it ...
        out.fill_(fill)                                        # ... and so does the fill of the output
        rc, offs, lens = sqy.encode_slabs_device(pipeline, d_vol.data_ptr(), shape, np.uint16, nslabs, out.data_ptr(), cap, inflight=3)
        assert rc == 0
        for i in range(nslabs):
            assert bytes(out[offs[i]:offs[i] + lens[i]].cpu().numpy().tobytes()) == want[i], (fill, i)


def test_slabs_call_errors(sqy):
    import torch
    dev = torch.device("cuda", 0)
    vol = torch.zeros((4, 16, 16), dtype=torch.uint16, device=dev)
    out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    assert sqy.encode_slabs_device("bitswap1->lz4", vol.data_ptr(), (4, 16, 16), np.uint16, 5, out.data_ptr(), 1 << 16)[0] == 1     # more slabs than frames
    assert sqy.encode_slabs_device("no_such_stage->lz4", vol.data_ptr(), (4, 16, 16), np.uint16, 2, out.data_ptr(), 1 << 16)[0] == 1
    assert sqy.encode_slabs_device("bitswap1->lz4", vol.data_ptr(), (4, 16, 16), np.uint16, 2, out.data_ptr(), 64)[0] == 1            # no room


def test_plain_c_caller():
    """tools/slabs_c_test.c (gcc, HIP runtime API for device memory only): one Slabs call against the single calls, blobs equal"""
    exe = os.path.join(ROOT, "sqeazy_amd", "bin", "slabs_c_test")
    r = subprocess.run([exe, "64", "256", "512", "4", "bitswap1->lz4", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "blobs equal" in r.stdout
