"""GPU parity tests: the HIP path behind the C-ABI against the CPU oracle, byte for byte."""
import numpy as np
import pytest

from sqeazy_amd import synth

pytestmark = pytest.mark.gpu


def _cases_u16():
    rng = np.random.default_rng(7)
    yield "synth_32x64x64", synth.stack((32, 64, 64))
    yield "synth_64x256x256", synth.stack((64, 256, 256))
    yield "zeros", np.zeros((16, 128, 128), np.uint16)
    yield "random", rng.integers(0, 65536, (16, 128, 128), dtype=np.uint16)
    yield "ramp", (np.arange(24 * 100 * 52) % 32768).astype(np.uint16).reshape(24, 100, 52)
    yield "lowbits", rng.integers(0, 4, (8, 256, 256), dtype=np.uint16)
    yield "ragged_len", rng.integers(0, 300, (3, 7, 11), dtype=np.uint16)          # len % 16 != 0
    yield "tiny", rng.integers(0, 300, (1, 1, 5), dtype=np.uint16)
    yield "sparse", (rng.random((16, 128, 128)) < 0.01).astype(np.uint16) * 4095


@pytest.mark.parametrize("name,vol", list(_cases_u16()), ids=[c[0] for c in _cases_u16()])
def test_bitswap1_lz4_u16(sqy, oracle, name, vol):
    rc, blob = sqy.encode("bitswap1->lz4", vol, nthreads=2)
    assert rc == 0
    want = oracle.pipeline_encode("bitswap1->lz4", vol)
    assert len(blob) == len(want)
    assert blob == want


@pytest.mark.parametrize("name,vol", list(_cases_u16())[:6], ids=[c[0] for c in list(_cases_u16())[:6]])
def test_bitswap1_only_u16(sqy, oracle, name, vol):
    rc, blob = sqy.encode("bitswap1", vol, nthreads=2)
    assert rc == 0
    assert blob == oracle.pipeline_encode("bitswap1", vol)


@pytest.mark.parametrize("name,vol", list(_cases_u16()), ids=[c[0] for c in _cases_u16()])
def test_lz4_only_u16(sqy, oracle, name, vol):
    rc, blob = sqy.encode("lz4", vol, nthreads=2)
    assert rc == 0
    assert blob == oracle.pipeline_encode("lz4", vol)


def test_lz4_structured_bytes(sqy, oracle):
    """byte streams that stress the match finder: short periods, long runs, mixed literals"""
    rng = np.random.default_rng(11)
    n = 3 * (256 << 10) + 12345
    parts = []
    parts.append(np.tile(np.arange(7, dtype=np.uint8), n // 7 + 1)[:n])
    parts.append(np.repeat(rng.integers(0, 256, n // 64 + 1, dtype=np.uint8), 64)[:n])
    x = rng.integers(0, 256, n, dtype=np.uint8); x[rng.random(n) < 0.7] = 0
    parts.append(x)
    words = rng.integers(0, 256, (50, 12), dtype=np.uint8)
    parts.append(words[rng.integers(0, 50, n // 12 + 1)].reshape(-1)[:n])
    parts.append(rng.integers(0, 3, n, dtype=np.uint8))
    for i, p in enumerate(parts):
        vol = p.reshape(1, 1, -1)
        rc, blob = sqy.encode("lz4", vol, nthreads=2)
        assert rc == 0, i
        want = oracle.pipeline_encode("lz4", vol)
        assert blob == want, "stream %d differs (len %d vs %d)" % (i, len(blob), len(want))


def test_diff_bitswap_lz4_u16(sqy, oracle):
    for shape in ((16, 32, 48), (8, 8, 8), (40, 12, 20), (6, 8, 16)):
        vol = synth.stack(shape)
        rc, blob = sqy.encode("diff3x3x1->bitswap1->lz4", vol, nthreads=2)
        assert rc == 0
        assert blob == oracle.pipeline_encode("diff3x3x1->bitswap1->lz4", vol), shape
