"""Single-blob mode (sqeazy_amd/multi.py, SURVEY.md 8(e) "optional"): slab blobs made by the HIP path, re-ordered into ONE blob
that must be byte for byte what one call on the whole volume yields -- and what the oracle yields."""
import numpy as np
import pytest

from sqeazy_amd import multi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,shape,world", [(np.uint16, (16, 512, 1024), 2), (np.uint16, (32, 512, 1024), 4), (np.uint8, (16, 1024, 1024), 4),
                                               (np.uint16, (64, 1024, 1024), 8)])
def test_slab_blobs_reorder_into_the_whole_volume_blob(sqy, oracle, dtype, shape, world):
    import torch
    vol = synth.stack(shape, dtype)
    assert multi.single_blob_possible(shape, dtype, world)
    rc, whole = sqy.encode("bitswap1->lz4", vol, nthreads=0)
    assert rc == 0
    if np.prod(shape) <= (1 << 25):
        assert whole == oracle.pipeline_encode("bitswap1->lz4", vol)
    blobs, ranges = [], []
    for r in range(world):
        z0, nz = multi.slab_range(shape[0], r, world)
        rc, b = sqy.encode("bitswap1->lz4", vol[z0:z0 + nz], nthreads=0)
        assert rc == 0
        ranges.append(multi.plane_ranges(b, (nz,) + tuple(shape[1:]), dtype)[1])
        blobs.append(torch.frombuffer(bytearray(b), dtype=torch.uint8).to("cuda:0"))      # as the gather leaves them on the root GPU
    one = multi.assemble_single_blob(shape, dtype, blobs, ranges)
    assert one.is_cuda and bytes(one.cpu().numpy().tobytes()) == whole
    rc, back = sqy.decode(bytes(one.cpu().numpy().tobytes()))
    assert rc == 0 and np.array_equal(back, vol)
