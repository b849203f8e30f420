"""The `sqy` command line tool (sqeazy_amd/bin/sqy, csrc/sqy_cli.cpp): same verbs/options as the reference's tool
(/root/reference/src/cpp/src/sqy.cpp:183-330), driven through the C-ABI only.  CPU tests cover the TIFF reader/writer and
the verbs that need no device; the GPU tests compare its .sqy files with the oracle's bytes."""
import os
import subprocess

import numpy as np
import pytest

from sqeazy_amd import build as sqy_build
from sqeazy_amd import synth

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def sqy_bin():
    sqy_build.build()
    assert os.path.exists(sqy_build.CLI)
    return sqy_build.CLI


def run(binary, *args):
    p = subprocess.run([binary] + list(args), capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


def write_tiff(path, vol):
    frames = [PIL.fromarray(f) for f in vol]
    frames[0].save(path, save_all=True, append_images=frames[1:], compression=None)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8])
def test_scan_and_compare_tiff(sqy_bin, tmp_path, dtype):
    vol = synth.stack((5, 24, 40), dtype)
    a, b = str(tmp_path / "a.tif"), str(tmp_path / "b.tif")
    write_tiff(a, vol)
    rc, out, err = run(sqy_bin, "scan", a)
    assert rc == 0, err
    row = out.strip().splitlines()[-1].split(",")
    assert row[1] == "40x24x5" and int(row[2]) == 8 * np.dtype(dtype).itemsize
    assert int(row[3]) == vol.min() and int(row[4]) == vol.max() and abs(float(row[5]) - vol.mean()) < 1e-3 * vol.mean()
    rc, out, _ = run(sqy_bin, "cmp", a, a)
    assert rc == 0 and "equal" in out
    vol2 = vol.copy(); vol2[3, 7, 9] ^= 1
    write_tiff(b, vol2)
    rc, out, _ = run(sqy_bin, "compare", a, b)
    assert rc == 1 and "differ: 1 of" in out


def test_scan_sqy_header_and_errors(sqy_bin, tmp_path, oracle):
    vol = synth.stack((4, 16, 32), np.uint16)
    blob = oracle.pipeline_encode("bitswap1->lz4", vol)
    f = tmp_path / "x.sqy"
    f.write_bytes(blob)
    rc, out, _ = run(sqy_bin, "info", str(f))
    assert rc == 0 and '"pipename": "bitswap1(num_bits_per_plane=1)->lz4(' in out and '"rank": "3"' in out
    assert run(sqy_bin, "frobnicate", str(f))[0] == 1
    assert run(sqy_bin, "compress", str(tmp_path / "missing.tif"))[0] == 1
    rc, _, err = run(sqy_bin, "compress", "-p", "no_such_stage->lz4", str(f))
    assert rc == 1 and "unable to build pipeline" in err
    assert run(sqy_bin, "--help")[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,pipeline", [(np.uint16, "bitswap1->lz4"), (np.uint16, "diff3x3x1->bitswap1->lz4"), (np.uint8, "lz4"),
                                            (np.uint16, "quantiser->bitswap1->lz4")])
def test_compress_matches_oracle_and_round_trips(sqy_bin, tmp_path, oracle, dtype, pipeline):
    vol = synth.stack((24, 64, 96), dtype)
    tif, sqy, back = str(tmp_path / "s.tif"), str(tmp_path / "s.sqy"), str(tmp_path / "back.tif")
    write_tiff(tif, vol)
    rc, out, err = run(sqy_bin, "compress", "-p", pipeline, "-v", tif)
    assert rc == 0, err
    want = oracle.pipeline_encode(pipeline, vol, nthreads=1)      # the tool's default, as the reference's (src/sqy.cpp:190)
    assert open(sqy, "rb").read() == want
    rc, out, err = run(sqy_bin, "decompress", "-o", back, sqy)
    assert rc == 0, err
    im = PIL.open(back)
    got = np.stack([np.array(im.seek(i) or im) for i in range(im.n_frames)])
    assert got.shape == vol.shape
    if pipeline.startswith("quantiser"):
        assert got.dtype == np.uint16 and np.array_equal(got, oracle.pipeline_decode(want))
    else:
        assert got.dtype == vol.dtype and np.array_equal(got, vol)
        assert run(sqy_bin, "compare", tif, back)[0] == 0


@pytest.mark.gpu
def test_raw_input_bench_and_serial_layout(sqy_bin, tmp_path, oracle):
    vol = synth.stack((16, 32, 64), np.uint16)
    raw = tmp_path / "v.raw"
    raw.write_bytes(vol.tobytes())
    rc, _, err = run(sqy_bin, "enc", "-s", "16x32x64", "-t", "uint16", "-o", str(tmp_path / "v.sqy"), str(raw))
    assert rc == 0, err
    assert (tmp_path / "v.sqy").read_bytes() == oracle.pipeline_encode("bitswap1->lz4", vol, nthreads=1)
    rc, _, _ = run(sqy_bin, "dec", "-e", ".raw", "-o", str(tmp_path / "w.raw"), str(tmp_path / "v.sqy"))
    assert rc == 0 and (tmp_path / "w.raw").read_bytes() == vol.tobytes()
    rc, out, err = run(sqy_bin, "bench", "-r", "3", "-c", "-s", "16x32x64", str(raw))
    assert rc == 0, err
    lines = out.strip().splitlines()
    assert lines[0].startswith("id,shape,time_mus,final_bytes,ingest_bw_mbps") and len(lines) == 4
    # the reference's default (nthreads = 1, src/sqy.cpp:190): one block-linked frame over several blocks
    big = synth.stack((16, 128, 128), np.uint16)
    (tmp_path / "b.raw").write_bytes(big.tobytes())
    assert run(sqy_bin, "enc", "-s", "16x128x128", "-o", str(tmp_path / "b1.sqy"), str(tmp_path / "b.raw"))[0] == 0
    assert (tmp_path / "b1.sqy").read_bytes() == oracle.pipeline_encode("bitswap1->lz4", big, nthreads=1)
    assert run(sqy_bin, "enc", "-n", "0", "-s", "16x128x128", "-o", str(tmp_path / "b0.sqy"), str(tmp_path / "b.raw"))[0] == 0
    assert (tmp_path / "b0.sqy").read_bytes() == oracle.pipeline_encode("bitswap1->lz4", big)
