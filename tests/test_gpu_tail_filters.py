"""diff3x3x1 and frame_shuffle as TAIL filters on the sink's `char` output (src/sqeazy_pipelines.hpp:64-77): signed bytes -- the 9-neighbour
sum is sign-extended into the unsigned short sum type before the division (diff_scheme_impl.hpp:24, traits.hpp:29), the frame metric
sums values from -128 to 127.  Blob bytes against the oracle, decode against the oracle's decode."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _vols():
    rng = np.random.default_rng(21)
    yield (rng.gamma(2.0, 300.0, (20, 30, 40)) + 50).astype(np.uint16)            # > 256 levels: bytes from 0 to 255 (negative as char)
    yield synth.stack((24, 64, 96), np.uint16)
    yield rng.integers(0, 65536, (16, 33, 35), dtype=np.uint16)                    # ragged, whole range


@pytest.mark.parametrize("pipeline", ["quantiser->diff3x3x1->lz4", "quantiser->diff3x3x1->bitswap1->lz4", "quantiser->frame_shuffle->lz4",
                                      "quantiser->frame_shuffle->bitswap1->lz4", "quantiser->diff3x3x1", "pass_through->frame_shuffle->lz4"])
def test_tail_filters_on_char(sqy, oracle, pipeline):
    assert sqy.pipeline_possible(pipeline, np.uint16)
    for vol in _vols():
        want = oracle.pipeline_encode(pipeline, vol)
        rc, blob = sqy.encode(pipeline, vol, nthreads=2, extra_capacity=16 * vol.shape[0] + 512)
        assert rc == 0, (pipeline, vol.shape)
        assert blob == want, (pipeline, vol.shape)
        rc, back = sqy.decode(blob)
        assert rc == 0 and np.array_equal(back, oracle.pipeline_decode(want)), (pipeline, vol.shape)


def test_tail_diff_needs_the_volume_shape(sqy):
    """behind a sink that does not write one byte per voxel the tail chain sees {1, 1, bytes} (dynamic_pipeline.hpp:658-666): diff3x3x1
    reads out of bounds there in the reference; refused.  8-bit extents above 127 overflow its char coordinates: refused as well."""
    vol = synth.stack((8, 16, 16), np.uint16)
    assert sqy.encode("pass_through->diff3x3x1->lz4", vol, nthreads=2)[0] == 1
    big = synth.stack((8, 16, 200), np.uint16)
    assert sqy.encode("quantiser->diff3x3x1->lz4", big, nthreads=2)[0] == 1
