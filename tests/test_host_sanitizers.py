"""Host side under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU ("race detection / sanitizers";
VERDICT round 3, item 9).  Two sanitized builds, made here with g++ (no GPU, no hipcc):
  * tests/sanitize/host_fuzz.cpp + csrc/sqy_pipeline.cpp: the pipeline grammar, configuration strings, header pack / unpack of
    untrusted bytes, base64, the LZ4 block planner, the quantiser's host LUTs and LUT files, the frame / tile ordering
  * csrc/sqy_cli.cpp (the `sqy` tool): its TIFF reader and .raw / option handling on malformed files.  The tool is linked against the
    real libsqeazy_amd.so; without a GPU every encode ends in the library's error code 1 AFTER the input has been read and checked,
    which is all this test needs.
GPU AddressSanitizer is not available on this pool; the kernels are covered by the parity tests instead."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sqeazy_amd", "csrc")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:exitcode=97:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _no_report(r):
    assert r.returncode not in (97, 98) and r.returncode >= 0, (r.returncode, r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_pipeline_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    subprocess.check_call(["g++"] + SAN + [os.path.join(ROOT, "tests", "sanitize", "host_fuzz.cpp"), os.path.join(CSRC, "sqy_pipeline.cpp"), "-o", exe, "-lpthread"])
    for seed in ("1", "20261004"):
        r = subprocess.run([exe, seed], env=dict(ENV, SQY_SAN_TMP=str(tmp_path)), capture_output=True, text=True, timeout=300)
        _no_report(r)
        assert r.returncode == 0 and "host_fuzz ok" in r.stdout, (r.stdout, r.stderr[-2000:])


# ---- the sqy tool's file readers ---------------------------------------------------------------------------------------------------
def _tiff(frames, h, w, bits=16, big=False, order="<"):
    """an uncompressed grayscale TIFF stack, one IFD and one strip per frame (what the reference's writer emits)"""
    bo = b"II" if order == "<" else b"MM"
    data = (np.arange(frames * h * w) % 251).astype(np.uint16 if bits == 16 else np.uint8)
    if bits == 16 and order == ">":
        data = data.byteswap()
    raw = data.tobytes()
    fb = h * w * bits // 8
    if big:
        out = bytearray(bo + struct.pack(order + "HHHQ", 43, 8, 0, 16))
    else:
        out = bytearray(bo + struct.pack(order + "HI", 42, 8))
    # pixel data first, IFDs behind
    data_at = len(out)
    out += raw
    ifd_at = []
    for k in range(frames):
        if len(out) % 2:
            out += b"\0"
        ifd_at.append(len(out))
        tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1), (273, 16 if big else 4, 1, data_at + k * fb),
                (277, 3, 1, 1), (278, 4, 1, h), (279, 16 if big else 4, 1, fb)]
        if big:
            out += struct.pack(order + "Q", len(tags))
            for tag, typ, cnt, val in tags:
                out += struct.pack(order + "HHQ", tag, typ, cnt)
                out += struct.pack(order + {3: "H6x", 4: "I4x", 16: "Q"}[typ], val)
            out += struct.pack(order + "Q", 0)
        else:
            out += struct.pack(order + "H", len(tags))
            for tag, typ, cnt, val in tags:
                out += struct.pack(order + "HHI", tag, typ, cnt)
                out += struct.pack(order + {3: "H2x", 4: "I"}[typ], val)
            out += struct.pack(order + "I", 0)
    # chain the IFDs
    for k, at in enumerate(ifd_at):
        nxt = ifd_at[k + 1] if k + 1 < frames else 0
        n_entries = 9
        if big:
            struct.pack_into(order + "Q", out, at + 8 + 20 * n_entries, nxt)
        else:
            struct.pack_into(order + "I", out, at + 2 + 12 * n_entries, nxt)
    if big:
        struct.pack_into(order + "Q", out, 8, ifd_at[0])
    else:
        struct.pack_into(order + "I", out, 4, ifd_at[0])
    return bytes(out), ifd_at


@pytest.fixture(scope="module")
def sqy_san(tmp_path_factory):
    lib = os.path.join(ROOT, "sqeazy_amd", "lib", "libsqeazy_amd.so")
    if not os.path.exists(lib):
        pytest.skip("libsqeazy_amd.so not built")
    d = tmp_path_factory.mktemp("sqy_san")
    exe = str(d / "sqy_san")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++"] + SAN + [os.path.join(CSRC, "sqy_cli.cpp"), "-o", exe, "-L" + os.path.dirname(lib), "-lsqeazy_amd",
                                          "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath-link," + os.path.join(rocm, "lib")])
    return exe


def _run(exe, args, cwd):
    return subprocess.run([exe] + args, env=ENV, capture_output=True, text=True, timeout=120, cwd=cwd)


def test_sqy_tool_reads_malformed_tiff_files_under_asan_ubsan(sqy_san, tmp_path):
    rng = np.random.default_rng(5)
    cases = {}
    for big in (False, True):
        for order in ("<", ">"):
            good, ifds = _tiff(3, 8, 16, 16, big, order)
            key = "%s%s" % ("big" if big else "classic", "le" if order == "<" else "be")
            cases[key + "_good"] = good
            for cut in (0, 1, 3, 7, 9, 17, len(good) // 2, ifds[0] + 1, ifds[0] + 5, ifds[-1] + 10, len(good) - 1):
                cases["%s_cut%d" % (key, cut)] = good[:cut]
            b = bytearray(good)                                   # an IFD chain that loops back to its first directory
            n_entries = 9
            at = ifds[-1] + (8 + 20 * n_entries if big else 2 + 12 * n_entries)
            struct.pack_into(order + ("Q" if big else "I"), b, at, ifds[0])
            cases[key + "_loop"] = bytes(b)
            b = bytearray(good)                                   # a first IFD far outside the file
            struct.pack_into(order + ("Q" if big else "I"), b, 8 if big else 4, (1 << 40) + 12 if big else 0xfffffff0)
            cases[key + "_ifd_outside"] = bytes(b)
            esz, cnt_off, val_off = (20, 4, 12) if big else (12, 4, 8)
            first = ifds[0] + (8 if big else 2)
            for tag_index, what, value in ((0, "huge_width", 0xffffffff), (1, "huge_height", 0xffffffff), (2, "bits_7", 7), (3, "compressed", 5),
                                           (5, "strip_outside", 0xffffff00), (8, "strip_bytes_huge", 0xffffffff)):
                b = bytearray(good)
                pos = first + esz * tag_index + val_off
                typ = struct.unpack_from(order + "H", b, first + esz * tag_index + 2)[0]
                struct.pack_into(order + {3: "H", 4: "I", 16: "Q"}[typ], b, pos, value if typ != 3 else value & 0xffff)
                cases["%s_%s" % (key, what)] = bytes(b)
            for tag_index, what, count in ((5, "strip_count_huge", 0xffffffff), (0, "width_count_huge", 0xffffffff), (8, "bytecounts_zero", 0)):
                b = bytearray(good)
                struct.pack_into(order + ("Q" if big else "I"), b, first + esz * tag_index + cnt_off, count)
                cases["%s_%s" % (key, what)] = bytes(b)
            b = bytearray(good)                                   # number of directory entries: 65535
            struct.pack_into(order + ("Q" if big else "H"), b, ifds[0], 0xffff)
            cases[key + "_many_entries"] = bytes(b)
            for k in range(25):                                   # random byte flips in the directories
                b = bytearray(good)
                for _ in range(1 + int(rng.integers(0, 6))):
                    b[int(rng.integers(ifds[0], len(b)))] = int(rng.integers(0, 256))
                cases["%s_flip%d" % (key, k)] = bytes(b)
    cases["empty"] = b""
    cases["not_tiff"] = b"hello world, this is no image\n" * 10
    for name, blob in cases.items():
        path = tmp_path / (name + ".tif")
        path.write_bytes(blob)
        for verb in (["compress", "-p", "lz4"], ["scan"], ["compare"]):
            args = verb + [str(path)] + ([str(path)] if verb[0] == "compare" else [])
            r = _run(sqy_san, args, str(tmp_path))
            _no_report(r)
    # .raw input with shapes that do not fit the file, option edge cases
    raw = tmp_path / "v.raw"
    raw.write_bytes(bytes(1000))
    for shape in ("10x10x5", "0x0x0", "1000000x1000000x1000000", "abc", "-1x2x3", "10x10", "1x1x1x1x1x1x1x1x1x1x1x1x1x1x1x1x1x1x1x1"):
        r = _run(sqy_san, ["compress", "-p", "lz4", "-s", shape, "-t", "uint16", str(raw)], str(tmp_path))
        _no_report(r)
    # a file that claims to be a sqy blob
    for blob in (b"", b"{", b'{"pipename":"lz4","raw":{"type":"t","rank":"3","shape":{"dim_0":"5"}},"encoded":{"bytes":"99999999999"}}|', bytes(range(256)) * 4):
        p = tmp_path / "x.sqy"
        p.write_bytes(blob)
        for verb in ("decompress", "scan"):
            _no_report(_run(sqy_san, [verb, str(p)], str(tmp_path)))
