"""HDF5 filter plugin (sqeazy_amd/lib/libh5sqy_amd.so, csrc/sqy_h5_filter.c): filter id 711 with the sqy header in
cd_values, like the reference's plugin (/root/reference/src/cpp/inc/sqeazy_h5_filter.hpp:28-227).  Driven through the plain
HDF5 C API by tools/h5_roundtrip.c; the stored chunk must be the oracle's blob byte for byte."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from sqeazy_amd import build as sqy_build
from sqeazy_amd import synth


@pytest.fixture(scope="module")
def h5(tmp_path_factory):
    sqy_build.build()
    if not (os.path.exists(sqy_build.H5_PLUGIN) and os.path.exists(sqy_build.H5_TOOL)):
        if sqy_build.build_h5_plugin() is None:
            pytest.skip("no HDF5 C library in this image")
    link = [f for f in os.listdir(sqy_build.LIBDIR) if f.startswith("libhdf5.so")]
    if not link or not os.path.exists(os.path.join(sqy_build.LIBDIR, link[0])):
        sqy_build.build_h5_plugin()              # the symlink to the image's libhdf5 did not travel with the snapshot
    env = dict(os.environ, HDF5_PLUGIN_PATH=sqy_build.LIBDIR)
    return env, tmp_path_factory.mktemp("h5")


def run_tool(env, tmp, vol, pipeline):
    raw = tmp / "in.raw"
    raw.write_bytes(vol.tobytes())
    z, y, x = vol.shape
    p = subprocess.run([sqy_build.H5_TOOL, str(raw), str(z), str(y), str(x), vol.dtype.name, pipeline, str(tmp / "out.h5"), str(tmp / "chunk.bin")],
                       capture_output=True, text=True, env=env, timeout=300)
    return p.returncode, p.stdout + p.stderr


def test_plugin_exports_and_is_discovered(h5):
    env, tmp = h5
    import sqeazy_amd
    sqeazy_amd.lib()                              # resolves libsqeazy_amd.so for the plugin's NEEDED entry
    plug = ctypes.CDLL(sqy_build.H5_PLUGIN)
    assert plug.H5PLget_plugin_type() == 0        # H5PL_TYPE_FILTER
    plug.H5PLget_plugin_info.restype = ctypes.c_void_p
    info = plug.H5PLget_plugin_info()
    assert info and ctypes.cast(info, ctypes.POINTER(ctypes.c_int))[1] == 0o1307   # H5Z_class2_t{version, id, ...}
    # HDF5 finds the plugin through HDF5_PLUGIN_PATH; without a GPU the encode inside the filter is refused (exit 4),
    # a missing plugin would be exit 3
    import torch
    if not torch.cuda.is_available():
        rc, out = run_tool(env, tmp, synth.stack((4, 16, 32), np.uint16), "bitswap1->lz4")
        assert rc == 4, out


def test_header_helpers(oracle):
    import sqeazy_amd
    L = sqeazy_amd.lib()
    shape = (ctypes.c_long * 3)(4, 16, 32)
    n = ctypes.c_long(0)
    assert L.SQYAMD_Header_Build(b"bitswap1->lz4", 2, shape, 3, ctypes.c_long(1234), None, ctypes.byref(n)) == 0
    buf = ctypes.create_string_buffer(n.value)
    assert L.SQYAMD_Header_Build(b"bitswap1->lz4", 2, shape, 3, ctypes.c_long(1234), buf, ctypes.byref(n)) == 0
    want = oracle.header_pack(np.uint16, (4, 16, 32), oracle.Pipeline.from_string("bitswap1->lz4").name(), 1234) if hasattr(oracle, "Pipeline") else None
    hdr = buf.raw[:n.value]
    assert hdr.endswith(b"|01307#!") and b'"bytes": "1234"' in hdr
    if want is not None:
        assert hdr == want
    m = ctypes.c_long(0)
    assert L.SQYAMD_Header_Pipeline(hdr, len(hdr), None, ctypes.byref(m)) == 0
    name = ctypes.create_string_buffer(m.value)
    assert L.SQYAMD_Header_Pipeline(hdr, len(hdr), name, ctypes.byref(m)) == 0
    assert name.value == b"bitswap1(num_bits_per_plane=1)->lz4(accel=1,blocksize_kb=256,framestep_kb=256,n_chunks_of_input=0)"
    assert L.SQYAMD_Header_Build(b"nonsense->lz4", 2, shape, 3, ctypes.c_long(0), None, ctypes.byref(n)) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,pipeline", [(np.uint16, "bitswap1->lz4"), (np.uint16, "diff3x3x1->bitswap1->lz4"), (np.uint8, "frame_shuffle->lz4")])
def test_h5_chunk_is_the_oracle_blob_and_round_trips(h5, oracle, dtype, pipeline):
    env, tmp = h5
    vol = synth.stack((24, 64, 96), dtype)
    rc, out = run_tool(env, tmp, vol, pipeline)
    assert rc == 0, out
    assert "round trip equal" in out
    assert (tmp / "chunk.bin").read_bytes() == oracle.pipeline_encode(pipeline, vol, nthreads=1)
