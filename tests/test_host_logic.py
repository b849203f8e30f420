"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its host logic (pipeline grammar,
size bounds, header queries) agrees with the oracle.  No compute calls: there is no GPU here and no CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"SQY_FUNCTION_PREFIX\s+[\w\s\*]+?\b(SQY\w+|SQYAMD\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(sqy):
    L = sqy.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert getattr(L, name) is not None, name
    assert sorted(set(sqy.EXPORTED_SYMBOLS)) == declared


def test_version(sqy):
    assert sqy.version_triple() == (0, 5, 2)
    assert b"gfx950" in sqy.lib().SQYAMD_Version()


def test_run_time_options(sqy):
    """SQYAMD_Set_Option / SQYAMD_Get_Option (include/sqeazy_amd.h): the switches the environment sets once at load; names and ranges checked"""
    assert sqy.get_option("block_parallel") == 1 and sqy.get_option("tail_scan") == 1 and sqy.get_option("transpose_chain") == 1
    assert sqy.get_option("transpose_chain_caller_streams") == 0          # coupling caller streams is opt-in (round-4 advice)
    assert sqy.get_option("block_parallel_warmup") == 65536
    assert sqy.get_option("decode_two_waves") == 1
    assert sqy.get_option("no_such_option") == -1
    L = sqy.lib()
    assert L.SQYAMD_Set_Option(b"no_such_option", 1) == 1 and L.SQYAMD_Set_Option(None, 1) == 1
    assert L.SQYAMD_Set_Option(b"block_parallel", 2) == 1 and L.SQYAMD_Set_Option(b"block_parallel_warmup", -1) == 1
    assert L.SQYAMD_Set_Option(b"block_parallel_warmup", (1 << 30) + 1) == 1
    with sqy.option("block_parallel_warmup", 200000):
        assert sqy.get_option("block_parallel_warmup") == 200000
    assert sqy.get_option("block_parallel_warmup") == 65536


def test_pipeline_possible_matches_reference_rules(sqy, oracle):
    """tests/test_pipeline_interface.cpp:28-61 + the documented restriction to implemented stages"""
    for dt in (np.uint16, np.uint8):
        assert sqy.pipeline_possible("bitswap1->lz4", dt)
        assert not sqy.pipeline_possible("", dt)
        assert not sqy.pipeline_possible("bswap1_lz4", dt)
    L = sqy.lib()
    assert L.SQY_Pipeline_Possible(b"bitswap1->lz4", 2) and L.SQY_Pipeline_Possible(b"bitswap1->lz4", 1)
    assert not L.SQY_Pipeline_Possible(b"bitswap1->lz4", 4)
    supported = ["lz4", "bitswap1", "diff3x3x1->bitswap1->lz4", "frame_shuffle->lz4", "quantiser->bitswap1->lz4",
                 "raster_reorder->lz4", "raster_reorder(tile_size=4)->bitswap1->lz4", "tile_shuffle->lz4", "zcurve_reorder->lz4",
                 "zcurve_reorder(tile_size=8)->bitswap1->lz4", "bitshuffle->lz4", "bitshuffle(block_size=64)->lz4", "quantiser->bitshuffle->lz4",
                 "pass_through", "pass_through->lz4", "bitswap1->pass_through->bitswap1->lz4",
                 # every tail filter of sqeazy_pipelines.hpp:64-77 but the video codecs (round 5: the reorder / shuffle stages too)
                 "quantiser->tile_shuffle->lz4", "quantiser->raster_reorder->lz4", "quantiser->zcurve_reorder(tile_size=4)->bitswap1->lz4",
                 "pass_through->raster_reorder->tile_shuffle(tile_size=8)->zcurve_reorder->frame_shuffle->diff3x3x1->bitshuffle->bitswap1->lz4",
                 "bitswap1(num_bits_per_plane=1)->lz4(accel=1,blocksize_kb=256,framestep_kb=256,n_chunks_of_input=0)"]
    for p in supported:
        assert oracle.can_be_built_from(p) and sqy.pipeline_possible(p), p
    # valid for the reference, not implemented here: answered false (documented deviation)
    for p in ["lz4(accel=9)", "lz4->bitswap1", "lz4->raster_reorder", "raster_reorder(tile_size=0)->lz4",
              "bitshuffle(block_size=12)->lz4", "remove_background->lz4", "quantiser->tile_shuffle(tile_size=0)->lz4"]:
        assert oracle.can_be_built_from(p) and not sqy.pipeline_possible(p), p
    assert not sqy.pipeline_possible("quantiser->lz4", np.uint8)
    # malformed
    for p in ["bitswap1->", "->lz4", "diff->bitswap1->lz4", "bitswap1->lz4(", "bitswap1 -> lz4"]:
        assert not sqy.pipeline_possible(p), p


@pytest.mark.parametrize("pipeline", ["bitswap1->lz4", "lz4", "bitswap1", "diff3x3x1->bitswap1->lz4", "frame_shuffle->lz4",
                                      "quantiser->bitswap1->lz4", "lz4(blocksize_kb=64,framestep_kb=64)", "lz4(n_chunks_of_input=7)"])
def test_max_compressed_length_matches_oracle(sqy, oracle, pipeline):
    for shape, dt in (((256, 256, 256), np.uint16), ((3, 7, 11), np.uint16), ((1024, 1024, 512), np.uint16), ((33, 65, 129), np.uint8)):
        if dt == np.uint8 and pipeline.startswith("quantiser"):
            continue
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        want = oracle.pipeline_max_encoded_size(pipeline, nbytes, dt, nthreads=1)
        assert sqy.max_compressed_length(pipeline, shape, dt) == want
        assert sqy.max_compressed_length_bytes(pipeline, nbytes, dt) == want
        assert want > nbytes                                               # tests/test_pipeline_interface.cpp:66-93


def test_max_compressed_length_rejects_bad_pipeline(sqy):
    with pytest.raises(ValueError):
        sqy.max_compressed_length("bswap1_lz4", (8, 8, 8))


def test_header_queries_on_oracle_blob(sqy, oracle):
    """tests/test_pipeline_interface.cpp:95-208: header size / dims / shape / sizeof / decoded length"""
    from sqeazy_amd import synth
    for vol in (synth.stack((6, 10, 14)), synth.stack((5, 9, 13), np.uint8)):
        for pipeline in ("bitswap1->lz4", "frame_shuffle->lz4"):
            blob = oracle.pipeline_encode(pipeline, vol)
            h = oracle.header_unpack(blob)
            assert sqy.header_size(blob) == h["size"]
            assert sqy.decompressed_ndims(blob) == 3
            assert sqy.decompressed_shape(blob) == vol.shape
            assert sqy.decompressed_sizeof(blob) == vol.dtype.itemsize
            assert sqy.decompressed_length(blob) == vol.nbytes


def test_no_cpu_fallback(sqy):
    """without a HIP device every encode answers the reference's error code 1 -- never a CPU result"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sqeazy_amd import synth
    rc, blob = sqy.encode("bitswap1->lz4", synth.stack((4, 8, 16)), nthreads=2)
    assert rc == 1 and blob is None


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under sqeazy_amd/ may import, include or link it"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sqeazy_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|sqy_oracle|#include\s+\"[^\"]*oracle", text, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    so = os.path.join(ROOT, "sqeazy_amd", "lib", "libsqeazy_amd.so")
    assert b"sqo_" not in open(so, "rb").read()
