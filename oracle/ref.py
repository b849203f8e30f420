"""Loader for oracle/_ref/libsqy_ref.so -- the driver around the REAL reference pieces that build in
this image (reference SSE bit-plane gather + the image's liblz4 1.9.3).  TEST INFRASTRUCTURE ONLY.

The library is built by oracle/Makefile from /root/reference when that tree is present; on the GPU box
only the prebuilt file (shipped with the snapshot) can be used.  `available()` says whether it loads.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_TRIED = False

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)


def lib():
    global _LIB, _TRIED
    if not _TRIED:
        _TRIED = True
        path = os.path.join(_HERE, "_ref", "libsqy_ref.so")
        try:
            L = ctypes.CDLL(path)
            for f in ("ref_lz4f_compress_bound", "ref_lz4_encode_serial", "ref_lz4_encode_parallel", "ref_lz4_decode_frames"):
                getattr(L, f).restype = ctypes.c_size_t
            _LIB = L
        except OSError:
            _LIB = None
    return _LIB


def available():
    return lib() is not None


def lz4_version():
    return lib().ref_lz4_version()


def bitswap1_encode_u16(a, nthreads=1):
    """reference simd_segment_broadcast; needs len % 128 == 0 and 16-byte alignment."""
    flat = np.ascontiguousarray(a, dtype=np.uint16).reshape(-1)
    buf = np.zeros(flat.size + 8, dtype=np.uint16)
    off = (-buf.ctypes.data % 16) // 2
    src = buf[off:off + flat.size]
    src[:] = flat
    out = np.zeros(flat.size, dtype=np.uint16)
    rc = lib().ref_bitswap1_encode_u16(src.ctypes.data_as(_u16p), out.ctypes.data_as(_u16p),
                                       ctypes.c_size_t(flat.size), ctypes.c_int(nthreads))
    if rc:
        raise ValueError("reference would take its scalar branch (not buildable here)")
    return out


def _bytes(a):
    a = a if isinstance(a, np.ndarray) else np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


def lz4_encode_parallel(data, chunk=256 << 10, accel=1, block_id=5, nthreads=2):
    src = _bytes(data)
    nchunks = max(1, (src.size + chunk - 1) // chunk)
    stride = lib().ref_lz4f_compress_bound(ctypes.c_size_t(chunk), ctypes.c_int(accel), ctypes.c_int(block_id)) + 19
    cap = nchunks * stride + 64
    dst = np.zeros(cap, dtype=np.uint8)
    n = lib().ref_lz4_encode_parallel(src.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(src.size),
                                      dst.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(cap), ctypes.c_size_t(chunk),
                                      ctypes.c_int(accel), ctypes.c_int(block_id), ctypes.c_int(nthreads))
    if n == 0:
        raise RuntimeError("liblz4 frame compression failed")
    return dst[:n]


def lz4_encode_serial(data, framestep=256 << 10, accel=1, block_id=5):
    src = _bytes(data)
    cap = src.size + (src.size // (64 << 10) + 2) * 16 + 64
    dst = np.zeros(cap, dtype=np.uint8)
    n = lib().ref_lz4_encode_serial(src.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(src.size),
                                    dst.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(cap), ctypes.c_size_t(framestep),
                                    ctypes.c_int(accel), ctypes.c_int(block_id))
    if n == 0:
        raise RuntimeError("liblz4 frame compression failed")
    return dst[:n]


def lz4_block(data, cap=None, accel=1):
    src = _bytes(data)
    cap = src.size - 1 if cap is None else cap
    dst = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
    r = lib().ref_lz4_block_fast_continue(src.ctypes.data_as(ctypes.c_char_p), ctypes.c_int(src.size),
                                          dst.ctypes.data_as(ctypes.c_char_p), ctypes.c_int(cap), ctypes.c_int(accel))
    return dst[:r].tobytes() if r > 0 else None


def lz4_decode_frames(data, cap):
    src = _bytes(data)
    dst = np.zeros(cap + 8, dtype=np.uint8)
    n = lib().ref_lz4_decode_frames(src.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(src.size),
                                    dst.ctypes.data_as(ctypes.c_char_p), ctypes.c_size_t(cap))
    if n == ctypes.c_size_t(-1).value:
        raise ValueError("liblz4 rejected the stream")
    return dst[:n]


def lz4f_compress_bound(n, accel=1, block_id=5):
    return lib().ref_lz4f_compress_bound(ctypes.c_size_t(n), ctypes.c_int(accel), ctypes.c_int(block_id))
