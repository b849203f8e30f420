"""sqeazy_amd -- MI355X-native drop-in for sqeazy's pipeline encode path.

The product is ``sqeazy_amd/lib/libsqeazy_amd.so`` (C-ABI in ``include/sqeazy_amd.h``, hand-written HIP
kernels for gfx950).  This module is only the ctypes binding used by the tests and ``bench.py``; it mirrors
the reference's C-ABI names (``src/cpp/inc/sqeazy.h``) one to one and adds thin numpy / device-pointer helpers.

There is no CPU fallback: if the library is missing, importing works but every call raises, and every encode
on a machine without a HIP device returns the reference's error code 1.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsqeazy_amd.so")
_lib = None

EXPORTED_SYMBOLS = (
    "SQY_Header_Size", "SQY_Decompressed_NDims", "SQY_Decompressed_Shape", "SQY_Decompressed_Sizeof",
    "SQY_Version_Triple", "SQY_PipelineEncode_UI8", "SQY_PipelineEncode_UI16",
    "SQY_Pipeline_Max_Compressed_Length_UI8", "SQY_Pipeline_Max_Compressed_Length_UI16",
    "SQY_Pipeline_Max_Compressed_Length_3D_UI8", "SQY_Pipeline_Max_Compressed_Length_3D_UI16",
    "SQY_Pipeline_Possible_UI16", "SQY_Pipeline_Possible_UI8", "SQY_Pipeline_Possible",
    "SQY_Decompressed_Length", "SQY_Decode_UI16", "SQY_Decode_UI8",
    "SQYAMD_PipelineEncode_UI16_Device", "SQYAMD_PipelineEncode_UI8_Device",
    "SQYAMD_PipelineEncode_UI16_DeviceAt", "SQYAMD_PipelineEncode_UI8_DeviceAt",
    "SQYAMD_PipelineEncode_UI16_DeviceAt_Frames", "SQYAMD_PipelineEncode_UI8_DeviceAt_Frames",
    "SQYAMD_PipelineEncode_Slabs_UI16_Device", "SQYAMD_PipelineEncode_Slabs_UI8_Device",
    "SQYAMD_PipelineEncode_UI16_Cap", "SQYAMD_PipelineEncode_UI8_Cap",
    "SQYAMD_Decode_UI16_Device", "SQYAMD_Decode_UI8_Device",
    "SQYAMD_Profile_Enable", "SQYAMD_Profile_Reset", "SQYAMD_Profile_Get",
    "SQYAMD_Release_Workspace", "SQYAMD_Set_Option", "SQYAMD_Get_Option", "SQYAMD_Version", "SQYAMD_Header_Pipeline", "SQYAMD_Header_Build",
    "SQYAMD_Comm_UniqueId", "SQYAMD_Comm_Init", "SQYAMD_Comm_Destroy", "SQYAMD_Gather_Blobs",
)


class LibraryMissing(RuntimeError):
    pass


def lib():
    """the loaded libsqeazy_amd.so; raises LibraryMissing when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing("%s not built: run `python -m sqeazy_amd.build` (hipcc, gfx950)" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  When it is loaded first,
        # our DT_NEEDED libamdhip64.so.7 resolves to that same copy; loaded the other way round the process
        # ends up with two runtimes and the second one sees no device.  So bring torch's in first when torch
        # is installed (tests / bench use torch for device memory); plain C/C++ callers simply get /opt/rocm's.
        if os.environ.get("SQEAZY_AMD_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(LIB_PATH)
        c_long_p = ctypes.POINTER(ctypes.c_long)
        L.SQY_Pipeline_Possible_UI16.restype = ctypes.c_bool
        L.SQY_Pipeline_Possible_UI8.restype = ctypes.c_bool
        L.SQY_Pipeline_Possible.restype = ctypes.c_bool
        L.SQYAMD_Version.restype = ctypes.c_char_p
        L.SQYAMD_Set_Option.argtypes = [ctypes.c_char_p, ctypes.c_long]
        L.SQYAMD_Get_Option.argtypes = [ctypes.c_char_p]
        L.SQYAMD_Get_Option.restype = ctypes.c_long
        L.SQYAMD_Profile_Get.restype = ctypes.c_char_p
        L.SQYAMD_Profile_Get.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), c_long_p]
        for f in ("SQY_PipelineEncode_UI8", "SQY_PipelineEncode_UI16"):
            getattr(L, f).argtypes = [ctypes.c_char_p, ctypes.c_void_p, c_long_p, ctypes.c_uint, ctypes.c_void_p, c_long_p, ctypes.c_int]
        for f in ("SQYAMD_PipelineEncode_UI8_Device", "SQYAMD_PipelineEncode_UI16_Device"):
            getattr(L, f).argtypes = [ctypes.c_char_p, ctypes.c_void_p, c_long_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_long,
                                      c_long_p, ctypes.c_int, ctypes.c_void_p]
        for f in ("SQYAMD_PipelineEncode_UI8_DeviceAt", "SQYAMD_PipelineEncode_UI16_DeviceAt"):
            getattr(L, f).argtypes = [ctypes.c_char_p, ctypes.c_void_p, c_long_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_long,
                                      c_long_p, c_long_p, ctypes.c_int, ctypes.c_void_p]
        for f in ("SQYAMD_PipelineEncode_Slabs_UI8_Device", "SQYAMD_PipelineEncode_Slabs_UI16_Device"):
            getattr(L, f).argtypes = [ctypes.c_char_p, ctypes.c_void_p, c_long_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p, ctypes.c_long,
                                      c_long_p, c_long_p, ctypes.c_int, ctypes.c_int]
        for f in ("SQYAMD_PipelineEncode_UI8_Cap", "SQYAMD_PipelineEncode_UI16_Cap"):
            getattr(L, f).argtypes = [ctypes.c_char_p, ctypes.c_void_p, c_long_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_long, c_long_p,
                                      ctypes.c_int]
        for f in ("SQY_Decode_UI8", "SQY_Decode_UI16"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int]
        for f in ("SQYAMD_Decode_UI8_Device", "SQYAMD_Decode_UI16_Device"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        _lib = L
    return _lib


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.uint16:
        return "UI16"
    if dtype == np.uint8:
        return "UI8"
    raise TypeError("sqeazy pipelines take uint8 or uint16 voxels, got %s" % dtype)


def _longs(values):
    return (ctypes.c_long * len(values))(*[int(v) for v in values])


def version_triple():
    v = (ctypes.c_int * 3)()
    lib().SQY_Version_Triple(v)
    return tuple(v)


def pipeline_possible(pipeline, dtype=np.uint16):
    return bool(getattr(lib(), "SQY_Pipeline_Possible_" + _suffix(dtype))(pipeline.encode()))


def max_compressed_length(pipeline, shape, dtype=np.uint16):
    """SQY_Pipeline_Max_Compressed_Length_3D_*: note the in/out convention (*length in = strlen(pipeline))."""
    p = pipeline.encode()
    length = ctypes.c_long(len(p))
    rc = getattr(lib(), "SQY_Pipeline_Max_Compressed_Length_3D_" + _suffix(dtype))(p, _longs(shape), ctypes.c_uint(len(shape)),
                                                                                  ctypes.byref(length))
    if rc:
        raise ValueError("SQY_Pipeline_Max_Compressed_Length_3D returned %d for %r" % (rc, pipeline))
    return length.value


def max_compressed_length_bytes(pipeline, nbytes, dtype=np.uint16):
    p = pipeline.encode()
    length = ctypes.c_long(int(nbytes))
    rc = getattr(lib(), "SQY_Pipeline_Max_Compressed_Length_" + _suffix(dtype))(p, ctypes.c_long(len(p)), ctypes.byref(length))
    if rc:
        raise ValueError("SQY_Pipeline_Max_Compressed_Length returned %d for %r" % (rc, pipeline))
    return length.value


def encode(pipeline, volume, nthreads=0, extra_capacity=None):
    """SQY_PipelineEncode_UI8/UI16 on a host ndarray ({z,y,x}); returns (return code, blob bytes or None).

    With extra_capacity=None this is the reference protocol: dst holds exactly SQY_Pipeline_Max_Compressed_Length_3D
    bytes.  With a number, dst is that much larger and the explicit-capacity entry point is used."""
    vol = np.ascontiguousarray(volume)
    sfx = _suffix(vol.dtype)
    if not pipeline_possible(pipeline, vol.dtype):
        cap = 64
    else:
        cap = max(max_compressed_length(pipeline, vol.shape, vol.dtype), 64)
    dlen = ctypes.c_long(0)
    if extra_capacity is None:
        dst = np.empty(cap, dtype=np.uint8)
        rc = getattr(lib(), "SQY_PipelineEncode_" + sfx)(pipeline.encode(), vol.ctypes.data, _longs(vol.shape), ctypes.c_uint(vol.ndim),
                                                        dst.ctypes.data, ctypes.byref(dlen), ctypes.c_int(nthreads))
    else:
        cap += int(extra_capacity)
        dst = np.empty(cap, dtype=np.uint8)
        rc = getattr(lib(), "SQYAMD_PipelineEncode_%s_Cap" % sfx)(pipeline.encode(), vol.ctypes.data, _longs(vol.shape),
                                                                 ctypes.c_uint(vol.ndim), dst.ctypes.data, ctypes.c_long(cap),
                                                                 ctypes.byref(dlen), ctypes.c_int(nthreads))
    if rc:
        return rc, None
    return 0, dst[:dlen.value].tobytes()


def encode_device(pipeline, d_src, shape, dtype, d_dst, dst_capacity, nthreads=0, stream=None):
    """SQYAMD_PipelineEncode_*_Device on raw device pointers (ints); returns (rc, bytes written)."""
    sfx = _suffix(dtype)
    dlen = ctypes.c_long(0)
    rc = getattr(lib(), "SQYAMD_PipelineEncode_%s_Device" % sfx)(
        pipeline.encode(), ctypes.c_void_p(int(d_src)), _longs(shape), ctypes.c_uint(len(shape)), ctypes.c_void_p(int(d_dst)),
        ctypes.c_long(int(dst_capacity)), ctypes.byref(dlen), ctypes.c_int(nthreads), ctypes.c_void_p(stream or 0))
    return rc, dlen.value


def encode_device_at(pipeline, d_src, shape, dtype, d_dst, dst_capacity, nthreads=0, stream=None):
    """SQYAMD_PipelineEncode_*_DeviceAt: the blob may start anywhere inside the destination; returns (rc, offset, bytes)."""
    sfx = _suffix(dtype)
    dlen, doff = ctypes.c_long(0), ctypes.c_long(0)
    rc = getattr(lib(), "SQYAMD_PipelineEncode_%s_DeviceAt" % sfx)(
        pipeline.encode(), ctypes.c_void_p(int(d_src)), _longs(shape), ctypes.c_uint(len(shape)), ctypes.c_void_p(int(d_dst)),
        ctypes.c_long(int(dst_capacity)), ctypes.byref(doff), ctypes.byref(dlen), ctypes.c_int(nthreads), ctypes.c_void_p(stream or 0))
    return rc, doff.value, dlen.value


def encode_device_at_frames(pipeline, d_src, shape, dtype, d_dst, dst_capacity, every, nthreads=0, stream=None):
    """SQYAMD_PipelineEncode_*_DeviceAt_Frames: returns (rc, offset, bytes, frame_offsets) -- frame_offsets[i] = start of LZ4
    frame i * every relative to the blob start, the last entry is the blob length."""
    nvox = 1
    for d in shape:
        nvox *= int(d)
    max_entries = nvox * np.dtype(dtype).itemsize // (64 << 10) // max(int(every), 1) + 4
    fo = (ctypes.c_long * max_entries)()
    cnt = ctypes.c_int(0)
    dlen, doff = ctypes.c_long(0), ctypes.c_long(0)
    rc = getattr(lib(), "SQYAMD_PipelineEncode_%s_DeviceAt_Frames" % _suffix(dtype))(
        ctypes.c_char_p(pipeline.encode()), ctypes.c_void_p(int(d_src)), _longs(shape), ctypes.c_uint(len(shape)), ctypes.c_void_p(int(d_dst)),
        ctypes.c_long(int(dst_capacity)), ctypes.byref(doff), ctypes.byref(dlen), ctypes.c_int(nthreads), ctypes.c_void_p(stream or 0),
        ctypes.c_int(int(every)), fo, ctypes.c_int(max_entries), ctypes.byref(cnt))
    return rc, doff.value, dlen.value, list(fo[:cnt.value + 1])


def encode_slabs_device(pipeline, d_src, shape, dtype, nslabs, d_dst, slab_capacity, nthreads=0, inflight=3):
    """SQYAMD_PipelineEncode_Slabs_*_Device: a whole volume as nslabs z-slab blobs with one call; returns (rc, offsets, lengths)."""
    offs = (ctypes.c_long * nslabs)()
    lens = (ctypes.c_long * nslabs)()
    rc = getattr(lib(), "SQYAMD_PipelineEncode_Slabs_%s_Device" % _suffix(dtype))(
        pipeline.encode(), ctypes.c_void_p(int(d_src)), _longs(shape), ctypes.c_uint(len(shape)), ctypes.c_int(nslabs),
        ctypes.c_void_p(int(d_dst)), ctypes.c_long(int(slab_capacity)), offs, lens, ctypes.c_int(nthreads), ctypes.c_int(inflight))
    return rc, list(offs), list(lens)


def header_size(blob):
    n = ctypes.c_long(len(blob))
    lib().SQY_Header_Size(bytes(blob), ctypes.byref(n))
    return n.value


def decompressed_length(blob):
    n = ctypes.c_long(len(blob))
    lib().SQY_Decompressed_Length(bytes(blob), ctypes.byref(n))
    return n.value


def decompressed_ndims(blob):
    n = ctypes.c_long(len(blob))
    lib().SQY_Decompressed_NDims(bytes(blob), ctypes.byref(n))
    return n.value


def decompressed_shape(blob):
    nd = decompressed_ndims(blob)
    arr = (ctypes.c_long * max(nd, 1))()
    arr[0] = len(blob)
    lib().SQY_Decompressed_Shape(bytes(blob), arr)
    return tuple(arr[i] for i in range(nd))


def decompressed_sizeof(blob):
    n = ctypes.c_long(len(blob))
    lib().SQY_Decompressed_Sizeof(bytes(blob), ctypes.byref(n))
    return n.value


def decode(blob, nthreads=0):
    """SQY_Decode_UI8/UI16; returns (rc, ndarray or None)."""
    blob = bytes(blob)
    size = decompressed_sizeof(blob)
    shape = decompressed_shape(blob)
    if size not in (1, 2) or not shape:
        return 1, None
    dtype = np.uint16 if size == 2 else np.uint8
    out = np.empty(shape, dtype=dtype)
    src = np.frombuffer(blob, dtype=np.uint8)
    rc = getattr(lib(), "SQY_Decode_" + _suffix(dtype))(src.ctypes.data, ctypes.c_long(len(blob)), out.ctypes.data, ctypes.c_int(nthreads))
    return (rc, None) if rc else (0, out)


def set_option(name, value):
    """SQYAMD_Set_Option (run-time switches, include/sqeazy_amd.h); raises on an unknown name / a value out of range"""
    if lib().SQYAMD_Set_Option(name.encode(), ctypes.c_long(int(value))):
        raise ValueError("SQYAMD_Set_Option(%r, %r) refused" % (name, value))


def get_option(name):
    return int(lib().SQYAMD_Get_Option(name.encode()))


class option:
    """with sqeazy_amd.option("block_parallel", 0): ...  -- the option for the duration of the block (process-wide)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def profile_enable(on=True):
    lib().SQYAMD_Profile_Enable(ctypes.c_int(1 if on else 0))


def profile_reset():
    lib().SQYAMD_Profile_Reset()


def profile_get():
    """{kernel name: (total ms, launches)} since the last reset"""
    out = {}
    i = 0
    while True:
        ms = ctypes.c_double(0)
        n = ctypes.c_long(0)
        name = lib().SQYAMD_Profile_Get(ctypes.c_int(i), ctypes.byref(ms), ctypes.byref(n))
        if not name:
            break
        out[name.decode()] = (ms.value, n.value)
        i += 1
    return out
