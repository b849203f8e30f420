"""CPU oracle for the sqeazy hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product (``sqeazy_amd``) never imports it and fails loudly when its HIP library is missing.

Byte/integer arithmetic lives in ``sqy_oracle.c`` / ``sqy_oracle_float.c`` (plain C, built by
``oracle/Makefile``); this file composes the stages the way the reference's ``dynamic_pipeline`` does
and renders the sqy header.  Reference citations are relative to /root/reference/src/cpp/src.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SQY_VERSION = "0.5.2"      # build-time constant in the reference (sqeazy_header.hpp:172)
SQY_HEADREF = "mi355x"     # ditto (git describe --always at build time)
HEADER_END = b"|01307#!"   # sqeazy_header.hpp:586

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_szp = ctypes.POINTER(ctypes.c_size_t)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libsqy_oracle.so")
        src_newer = (not os.path.exists(path)) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(path)
            for f in ("sqy_oracle.c", "sqy_oracle_float.c", "sqy_oracle.h"))
        if src_newer:
            build()
        L = ctypes.CDLL(path)
        L.sqo_lz4_block_compress.restype = ctypes.c_int
        L.sqo_lz4_block_decompress.restype = ctypes.c_int
        for f in ("sqo_lz4_encode_chunked", "sqo_lz4_encode_serial", "sqo_lz4_max_encoded_size", "sqo_lz4_decode_frames",
                  "sqo_base64_encode", "sqo_diff3x3x1_offsets"):
            getattr(L, f).restype = ctypes.c_size_t
        L.sqo_xxh32.restype = ctypes.c_uint32
        _LIB = L
    return _LIB


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


# ------------------------------------------------------------------------------------------------
# stage-level wrappers
# ------------------------------------------------------------------------------------------------
def bitswap1_encode(a):
    a = np.ascontiguousarray(a)
    flat = a.reshape(-1)
    out = np.empty_like(flat)
    if flat.dtype == np.uint16:
        lib().sqo_bitswap1_encode_u16(_ptr(flat, _u16p), _ptr(out, _u16p), ctypes.c_size_t(flat.size))
    elif flat.dtype in (np.uint8, np.int8):
        lib().sqo_bitswap1_encode_u8(_ptr(flat, _u8p), _ptr(out, _u8p), ctypes.c_size_t(flat.size))
    else:
        raise TypeError(flat.dtype)
    return out.reshape(a.shape)


def setbits(destination, source, at, numbits, bits=16):
    """detail::setbits_of_integertype<T> (encoders/scalar_utils.hpp:75-80): `numbits` bits of `destination` from bit `at` on are
    replaced by the low bits of `source`; everything is truncated to the integer type (`bits` wide)"""
    m = (1 << bits) - 1
    ones = (((1 << numbits) - 1) << at) & m
    return ((ones | destination) ^ ((((~source) & m) << at) & ones)) & m


def remove_blanks(payload, n_bytes, stride):
    """lz4::remove_blanks (encoders/lz4_utils.hpp:175-190): chunk k was written at payload[k * stride, +n_bytes[k]); the
    chunks are moved together in place.  returns the number of bytes that are valid afterwards"""
    value = int(n_bytes[0])
    src = stride
    for n in n_bytes[1:]:
        n = int(n)
        payload[value:value + n] = payload[src:src + n].copy()
        value += n
        src += stride
    return value


def bitswap1_encode_planes(a, nthreads=1):
    flat = _c(a, np.uint16).reshape(-1)
    out = np.empty_like(flat)
    lib().sqo_bitswap1_encode_u16_planes(_ptr(flat, _u16p), _ptr(out, _u16p), ctypes.c_size_t(flat.size),
                                         ctypes.c_int(nthreads))
    return out


def bitswap1_decode(a):
    a = np.ascontiguousarray(a)
    flat = a.reshape(-1)
    out = np.empty_like(flat)
    if flat.dtype == np.uint16:
        lib().sqo_bitswap1_decode_u16(_ptr(flat, _u16p), _ptr(out, _u16p), ctypes.c_size_t(flat.size))
    else:
        lib().sqo_bitswap1_decode_u8(_ptr(flat, _u8p), _ptr(out, _u8p), ctypes.c_size_t(flat.size))
    return out.reshape(a.shape)


def _shape3(shape):
    return (ctypes.c_size_t * 3)(*[int(s) for s in shape])


def diff3x3x1_encode(a, char=False):
    """char=True: the tail filter form on the sink's `char` output (signed bytes: the sum is sign-extended before the division)"""
    a = np.ascontiguousarray(a)
    if a.ndim != 3:
        raise ValueError("diff3x3x1 needs a 3D shape (diff_scheme_impl.hpp:84-87)")
    out = np.empty_like(a)
    if char:
        rc = lib().sqo_diff3x3x1_encode_i8(_ptr(a.view(np.uint8), _u8p), _ptr(out.view(np.uint8), _u8p), _shape3(a.shape))
    elif a.dtype == np.uint16:
        rc = lib().sqo_diff3x3x1_encode_u16(_ptr(a, _u16p), _ptr(out, _u16p), _shape3(a.shape))
    else:
        rc = lib().sqo_diff3x3x1_encode_u8(_ptr(a, _u8p), _ptr(out, _u8p), _shape3(a.shape))
    if rc:
        raise ValueError("diff3x3x1: shape outside the reference's defined behaviour")
    return out


def diff3x3x1_decode(a, char=False):
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    if char:
        rc = lib().sqo_diff3x3x1_decode_i8(_ptr(a.view(np.uint8), _u8p), _ptr(out.view(np.uint8), _u8p), _shape3(a.shape))
    elif a.dtype == np.uint16:
        rc = lib().sqo_diff3x3x1_decode_u16(_ptr(a, _u16p), _ptr(out, _u16p), _shape3(a.shape))
    else:
        rc = lib().sqo_diff3x3x1_decode_u8(_ptr(a, _u8p), _ptr(out, _u8p), _shape3(a.shape))
    if rc:
        raise ValueError("diff3x3x1: shape outside the reference's defined behaviour")
    return out


def diff3x3x1_offsets(shape):
    hx = ctypes.c_size_t(0)
    n = lib().sqo_diff3x3x1_offsets(_shape3(shape), None, ctypes.c_size_t(0), ctypes.byref(hx))
    if n == ctypes.c_size_t(-1).value:
        raise ValueError("diff3x3x1: shape outside the reference's defined behaviour")
    out = np.zeros(max(n, 1), dtype=np.uint64)
    lib().sqo_diff3x3x1_offsets(_shape3(shape), _ptr(out, _szp), ctypes.c_size_t(n), ctypes.byref(hx))
    return out[:n], hx.value


def lz4_acceleration(accel):
    """liblz4's acceleration for sqeazy's `accel` (= LZ4F compressionLevel, encoders/lz4.hpp:103-113): a negative level -k means
    acceleration k + 1 (lz4frame.c 1.9.3, LZ4F_compressBlock), capped at 65537 (lz4.c); levels 0..2 mean 1"""
    return min(1 - int(accel), 65537) if accel < 0 else 1


class _Acceleration:
    """sets the C restatement's acceleration for the calls inside the block (module state: not thread safe, test infrastructure)"""

    def __init__(self, a):
        self.a = int(a)

    def __enter__(self):
        lib().sqo_lz4_set_acceleration(ctypes.c_int(self.a))

    def __exit__(self, *exc):
        lib().sqo_lz4_set_acceleration(ctypes.c_int(1))


def lz4_block_compress(data, cap=None, acceleration=1):
    src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else _c(data, np.uint8)
    n = src.size
    if cap is None:
        cap = n - 1
    dst = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
    with _Acceleration(acceleration):
        r = lib().sqo_lz4_block_compress(_ptr(src, _u8p), ctypes.c_int(n), _ptr(dst, _u8p), ctypes.c_int(cap))
    return dst[:r].tobytes() if r > 0 else None


def lz4_block_decompress(data, cap):
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    dst = np.zeros(cap + 8, dtype=np.uint8)
    r = lib().sqo_lz4_block_decompress(_ptr(src, _u8p), ctypes.c_int(src.size), _ptr(dst, _u8p), ctypes.c_int(cap))
    if r < 0:
        raise ValueError("corrupt lz4 block")
    return dst[:r].tobytes()


_BLOCK_ID = {64: 4, 256: 5, 1024: 6, 4096: 7}


def closest_blocksize_kb(kb):
    """encoders/lz4_utils.hpp:60-93 (closest_blocksize::of)"""
    sizes = [64, 256, 1024, 4096]
    import bisect
    i = bisect.bisect_left(sizes, kb)
    if i == len(sizes):
        return sizes[-1]
    if i == 0:
        return sizes[0]
    middle = sizes[i - 1] + (sizes[i] - sizes[i - 1]) // 2
    return sizes[i] if kb >= middle else sizes[i - 1]


class Lz4Config:
    """encoders/lz4.hpp:58-114 parameter logic + :132-141 config string + :146-156 bytes_per_chunk."""

    def __init__(self, cfg=""):
        self.accel, self.blocksize_kb, self.framestep_kb, self.n_chunks = 1, 256, 256, 0
        for k, v in parse_minors(cfg).items():
            if k == "accel":
                self.accel = int(float(v))
            elif k == "blocksize_kb":
                self.blocksize_kb = int(float(v))
            elif k == "framestep_kb":
                self.framestep_kb = int(float(v))
            elif k == "n_chunks_of_input":
                self.n_chunks = int(float(v))
        if self.framestep_kb < self.blocksize_kb:
            self.framestep_kb = self.blocksize_kb
        else:
            ratio = float(np.float32(self.framestep_kb) / np.float32(self.blocksize_kb))
            self.framestep_kb = int(np.floor(ratio + 0.5)) * self.blocksize_kb   # std::round: half away from zero
        if self.n_chunks != 0:
            self.framestep_kb = 0
        self.block_id = _BLOCK_ID[closest_blocksize_kb(self.blocksize_kb)]

    def config(self):
        return "accel=%d,blocksize_kb=%d,framestep_kb=%d,n_chunks_of_input=%d" % (
            self.accel, self.blocksize_kb, self.framestep_kb, self.n_chunks)

    def bytes_per_chunk(self, nbytes):
        value = (self.framestep_kb << 10) if self.framestep_kb else nbytes // self.n_chunks
        if value >= nbytes or self.n_chunks >= nbytes:
            value = nbytes
        return value

    def max_encoded_size(self, nbytes, nthreads=1):
        return lib().sqo_lz4_max_encoded_size(ctypes.c_size_t(nbytes), ctypes.c_size_t(self.bytes_per_chunk(nbytes)),
                                              ctypes.c_int(self.block_id), ctypes.c_int(nthreads))


def _lz4_dst(n, cfg, nframes):
    blocks = n // (64 << 10) + 2 * nframes + 2
    return np.zeros(n + 4 * blocks + 15 * nframes + 64, dtype=np.uint8)


def lz4_encode_chunked(data, cfg=None):
    """nthreads >= 2 layout (encoders/lz4.hpp:227-239 -> lz4_utils.hpp:193-274): every chunk its own frame; a chunk
    larger than one LZ4F block is a frame of block-linked blocks."""
    cfg = cfg or Lz4Config()
    src = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    src = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
    n = src.size
    chunk = cfg.bytes_per_chunk(n) if n else 1
    if cfg.accel >= 3:
        raise NotImplementedError("accel >= 3 selects LZ4HC in liblz4 (not restated)")
    nchunks = (n + chunk - 1) // chunk if n else 1
    dst = _lz4_dst(n, cfg, nchunks)
    with _Acceleration(lz4_acceleration(cfg.accel)):
        r = lib().sqo_lz4_encode_chunked(_ptr(src, _u8p), ctypes.c_size_t(n), _ptr(dst, _u8p), ctypes.c_size_t(chunk),
                                         ctypes.c_int(cfg.block_id))
    if r == 0:
        raise ValueError("lz4 configuration not encodable")
    return dst[:r]


def lz4_encode_serial(data, cfg=None, framestep=None):
    """nthreads == 1 layout (encoders/lz4.hpp:227-234 -> lz4_utils.hpp:99-173): ONE frame of block-linked blocks, fed to
    LZ4F_compressUpdate `framestep` bytes at a time."""
    cfg = cfg or Lz4Config()
    src = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    src = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
    n = src.size
    if cfg.accel >= 3:
        raise NotImplementedError("accel >= 3 selects LZ4HC in liblz4 (not restated)")
    if framestep is None:
        framestep = cfg.bytes_per_chunk(n) if n else 1
    dst = _lz4_dst(n, cfg, 1)
    with _Acceleration(lz4_acceleration(cfg.accel)):
        r = lib().sqo_lz4_encode_serial(_ptr(src, _u8p), ctypes.c_size_t(n), _ptr(dst, _u8p), ctypes.c_size_t(framestep),
                                        ctypes.c_int(cfg.block_id))
    if r == 0:
        raise ValueError("lz4 configuration not encodable")
    return dst[:r]


def lz4_encode(data, cfg=None, nthreads=2):
    """lz4_scheme::encode (encoders/lz4.hpp:214-242): the serial layout for exactly one thread, the chunked one otherwise."""
    return lz4_encode_serial(data, cfg) if nthreads == 1 else lz4_encode_chunked(data, cfg)


def lz4_decode_frames(data, cap):
    src = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), dtype=np.uint8)
    src = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
    dst = np.zeros(cap + 8, dtype=np.uint8)
    r = lib().sqo_lz4_decode_frames(_ptr(src, _u8p), ctypes.c_size_t(src.size), _ptr(dst, _u8p), ctypes.c_size_t(cap))
    if r == ctypes.c_size_t(-1).value:
        raise ValueError("corrupt lz4 frame stream")
    return dst[:r]


def histogram(a):
    flat = np.ascontiguousarray(a).reshape(-1)
    if flat.dtype == np.uint16:
        h = np.zeros(65536, dtype=np.uint32)
        lib().sqo_histogram_u16(_ptr(flat, _u16p), ctypes.c_size_t(flat.size), _ptr(h, _u32p))
    else:
        h = np.zeros(256, dtype=np.uint32)
        lib().sqo_histogram_u8(_ptr(flat.view(np.uint8), _u8p), ctypes.c_size_t(flat.size), _ptr(h, _u32p))
    return h


def quantiser_weighting(text):
    """(mode, numerator, denominator) the way quantiser_scheme::encode reads `weighting_function`
    (quantiser_scheme_impl.hpp:186-198, extract_ratio :25-47): a string containing "none" -> weights 1; otherwise every run of
    digits is an integer -- one: exponent n/1, two: n/d, anything else: 0/0 -- and "offset" anywhere selects offset_power_of.
    A string without "_" also gives 0/0.  0/0 and n/0 (NaN / infinite exponents) are rejected here: the reference computes
    garbage LUTs from them."""
    import re
    if "none" in text:
        return 0, 1, 1
    ints = [int(t) for t in re.findall(r"[0-9]+", text)] if "_" in text else []
    if len(ints) == 1:
        ints.append(1)
    if len(ints) != 2 or ints[1] == 0:
        raise ValueError("weighting_function=%s: the reference's exponent is not a finite number" % text)
    return (2 if "offset" in text else 1), ints[0], ints[1]


def quantiser_weights(histo, weighting="none"):
    histo = _c(histo, np.uint32)
    mode, num, den = quantiser_weighting(weighting)
    w = np.zeros(histo.size, dtype=np.float32)
    lib().sqo_quantiser_weights(_ptr(histo, _u32p), ctypes.c_size_t(histo.size), mode, num, den, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return w


def quantiser_build_luts(histo, weighting="none"):
    histo = _c(histo, np.uint32)
    enc = np.zeros(histo.size, dtype=np.uint8)
    dec = np.zeros(256, dtype=np.uint16)
    mode, num, den = quantiser_weighting(weighting)
    lib().sqo_quantiser_build_luts_w(_ptr(histo, _u32p), ctypes.c_size_t(histo.size), mode, num, den, _ptr(enc, _u8p), _ptr(dec, _u16p))
    return enc, dec


def quantiser_encode(a, weighting="none"):
    """quantiser_scheme<T,char>::encode (quantiser_scheme_impl.hpp:176-226).
    returns (bytes as uint8 array of a.shape, lut_decode as array of a.dtype[256])"""
    a = np.ascontiguousarray(a)
    enc, dec = quantiser_build_luts(histogram(a), weighting)
    flat = a.reshape(-1)
    out = np.empty(flat.size, dtype=np.uint8)
    if a.dtype == np.uint16:
        lib().sqo_quantiser_apply_u16(_ptr(flat, _u16p), ctypes.c_size_t(flat.size), _ptr(enc, _u8p), _ptr(out, _u8p))
    else:
        lib().sqo_quantiser_apply_u8(_ptr(flat, _u8p), ctypes.c_size_t(flat.size), _ptr(enc, _u8p), _ptr(out, _u8p))
    return out.reshape(a.shape), dec.astype(a.dtype)


def lut_to_file(path, lut):
    """quantiser::lut_to_file (quantiser_utils.hpp:490-498): one decimal value per line"""
    with open(path, "w") as f:
        for v in lut:
            f.write("%d\n" % int(v))


def lut_from_file(path, dtype=np.uint16):
    """quantiser::lut_from_file (quantiser_utils.hpp:501-515): whitespace-separated values into a 256-entry table"""
    lut = np.zeros(256, dtype=dtype)
    try:
        vals = open(path).read().split()
    except OSError:
        vals = []
    for i, t in enumerate(vals[:256]):
        lut[i] = int(t)
    return lut


def raster_reorder(a, tile_size=None, decode=False):
    """raster_reorder_scheme (encoders/raster_reorder_scheme_impl.hpp:97-147); default tile = 16 / sizeof(T)"""
    a = np.ascontiguousarray(a)
    if a.ndim != 3:
        raise ValueError("raster_reorder needs a 3D shape")
    ts = int(tile_size) if tile_size else 16 // a.dtype.itemsize
    out = np.empty_like(a)
    L = lib()
    L.sqo_raster_reorder.restype = ctypes.c_int
    rc = L.sqo_raster_reorder(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(out.ctypes.data), _shape3(a.shape),
                              ctypes.c_size_t(ts), ctypes.c_int(a.dtype.itemsize), ctypes.c_int(1 if decode else 0))
    if rc:
        raise ValueError("raster_reorder: geometry the reference leaves undefined (shape %r, tile %d)" % (a.shape, ts))
    return out


def _tiles_full(a, ts):
    """(ntiles, ts^3) view-copy of a volume whose extents are multiples of ts: tiles in (z,y,x) tile order, row-major inside"""
    Z, Y, X = a.shape
    return a.reshape(Z // ts, ts, Y // ts, ts, X // ts, ts).transpose(0, 2, 4, 1, 3, 5).reshape(-1, ts ** 3)


def _untile_full(tiles, shape, ts):
    Z, Y, X = shape
    return tiles.reshape(Z // ts, Y // ts, X // ts, ts, ts, ts).transpose(0, 3, 1, 4, 2, 5).reshape(shape)


def _tiled_offsets(shape, ts):
    """output offset of every voxel for 'tiles of ts^3 (smaller at the high ends) appended in (z,y,x) tile order, row-major
    inside each tile' -- the layout of zcurve_reorder (and of raster_reorder where that one is defined)"""
    Z, Y, X = shape
    z, y, x = np.indices(shape, dtype=np.int64)
    tz, ty, tx = z // ts, y // ts, x // ts
    ez = np.minimum(ts, Z - tz * ts)
    ey = np.minimum(ts, Y - ty * ts)
    ex = np.minimum(ts, X - tx * ts)
    tile_off = tz * ts * Y * X + ez * (ty * ts * X + ey * tx * ts)
    return tile_off + (z % ts) * ey * ex + (y % ts) * ex + (x % ts)


def zcurve_tile_ok(shape, ts):
    """geometries for which detail::zcurve is defined.  The 'Morton' code it applies inside a tile is
    morton_at_ct<log2(tile)>::from (zcurve_reorder_utils.hpp:30-67, morton.hpp:103-123): bit groups of log2(tile) bits are
    interleaved, and coordinates inside a tile have only ONE such group -- the in-tile order is plain row-major.  Tile sizes
    that are not 2..128 powers of two fall back to 1-bit Morton codes that run past the tile: undefined.  Shapes without
    remainder against their common power of two take encode_full, which needs tile_size to divide every extent."""
    if ts not in (2, 4, 8, 16, 32, 64, 128) or len(shape) != 3:
        return False
    common = 1 << min(int(d).bit_length() - 1 for d in shape)
    if all(d % common == 0 for d in shape) and any(d % ts for d in shape):
        return False
    return True


def zcurve_reorder(a, tile_size=2, decode=False):
    """zcurve_reorder_scheme (encoders/zcurve_reorder_scheme_impl.hpp:36-117, zcurve_reorder_utils.hpp:79-475)"""
    a = np.ascontiguousarray(a)
    ts = int(tile_size)
    if not zcurve_tile_ok(a.shape, ts):
        raise ValueError("zcurve_reorder: geometry the reference leaves undefined (shape %r, tile %d)" % (a.shape, ts))
    off = _tiled_offsets(a.shape, ts).reshape(-1)
    flat = a.reshape(-1)
    out = np.empty_like(flat)
    if decode:
        out[:] = flat[off]
    else:
        out[off] = flat
    return out.reshape(a.shape)


def tile_shuffle_encode(a, tile_size=32, char=False):
    """detail::tile_shuffle::encode_full (encoders/tile_shuffle_utils.hpp:104-224): tiles of tile^3 voxels, the metric of a
    tile is its SEQUENTIAL binary32 sum divided by the voxel count and CONVERTED TO THE VOXEL TYPE, tiles are appended in the
    order of the sorted metrics, slot i taking the FIRST tile whose metric equals sorted[i] (tiles with equal metrics all map
    to the first of them).  Shapes with a remainder take the reference's encode_with_remainder (Boost P^2 median over tiles
    read past their end, a thread-timing dependent map): not restated.  returns (volume-shaped output, decode_map)
    char=True: the tail filter form, tile_shuffle_scheme<char> on the sink's stream -- the sum adds SIGNED bytes and the metric is a
    (signed) char: `in_value_t` is char in :176-190, the float quotient converts to it by truncation and std::sort orders chars."""
    a = np.ascontiguousarray(a)
    if char:
        out, dmap = tile_shuffle_encode(a.view(np.int8), tile_size)
        return out.view(a.dtype), dmap
    ts = int(tile_size)
    if a.ndim != 3 or ts <= 0 or any(d % ts for d in a.shape) or any(d < ts for d in a.shape):
        raise ValueError("tile_shuffle: only shapes that are whole multiples of the tile are restated")
    tiles = _tiles_full(a, ts)
    sums = np.cumsum(tiles.astype(np.float32), axis=1, dtype=np.float32)[:, -1]          # accumulate = strictly sequential
    metric = (sums / np.float32(ts ** 3)).astype(a.dtype)
    order = np.sort(metric, kind="stable")
    first = {}
    for i, m in enumerate(metric.tolist()):
        first.setdefault(m, i)
    dmap = np.array([first[m] for m in order.tolist()], dtype=np.uint64)
    return tiles[dmap.astype(np.int64)].reshape(a.shape), dmap


def tile_shuffle_decode(a, dmap, tile_size=32):
    """detail::tile_shuffle::decode_with_remainder (:405-560): encoded tile i goes to slot decode_map[i] (a later i wins),
    slots nobody names stay zero"""
    a = np.ascontiguousarray(a)
    ts = int(tile_size)
    enc = a.reshape(-1, ts ** 3)
    tiles = np.zeros_like(enc)
    for i, t in enumerate(np.asarray(dmap).astype(np.int64).tolist()):
        tiles[t] = enc[i]
    return _untile_full(tiles, a.shape, ts)


def bitshuffle_block_elems(elem_size, block_size=0):
    """bshuf_default_block_size (bitshuffle_core.c): 8192 / elem_size rounded down to a multiple of 8, at least 128"""
    if block_size:
        return int(block_size)
    return max((8192 // elem_size) // 8 * 8, 128)


def bitshuffle(a, block_size=0, decode=False):
    """bitshuffle_scheme (encoders/bitshuffle_scheme_impl.hpp:91-100) = bshuf_bitshuffle(in, out, n, sizeof(T), block_size) of
    kiyo-masui/bitshuffle (the reference downloads psteinb's fork at configure time, src/cpp/CMakeLists.txt:361-367; the
    sources are NOT in the tree: restated from the published algorithm, PARITY UNPINNED).  Per block of `block_size` elements
    (default 8192 bytes worth; the last one rounded down to a multiple of 8 elements, the up to 7 elements behind it copied):
    bit r = 8 * byte + bit of every element, packed 8 elements per byte (element 8k+j at bit j), row after row."""
    a = np.ascontiguousarray(a)
    E = a.dtype.itemsize
    flat = a.reshape(-1).view(np.uint8)
    n = a.size
    bs = bitshuffle_block_elems(E, block_size)
    if bs % 8:
        raise ValueError("bitshuffle: block_size must be a multiple of 8")
    out = np.empty_like(flat)
    pos = 0

    def one(lo, cnt):
        seg = flat[lo * E:(lo + cnt) * E]
        if decode:
            rows = np.unpackbits(seg.reshape(E * 8, cnt // 8), axis=1, bitorder="little")      # (E*8, cnt)
            out[lo * E:(lo + cnt) * E] = np.packbits(rows.T, axis=1, bitorder="little").reshape(-1)
        else:
            bits = np.unpackbits(seg.reshape(cnt, E), axis=1, bitorder="little")               # (cnt, E*8)
            out[lo * E:(lo + cnt) * E] = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)

    while n - pos >= bs:
        one(pos, bs)
        pos += bs
    last = (n - pos) - (n - pos) % 8
    if last:
        one(pos, last)
        pos += last
    out[pos * E:] = flat[pos * E:]
    return out.view(a.dtype).reshape(a.shape)


def frame_shuffle_encode(a, char=False, chunk=1):
    """char=True: the tail filter form (frames of signed bytes: the metric sums values from -128 to 127).
    chunk = frame_chunk_size: `chunk` consecutive frames are one sort unit (frame_shuffle_utils.hpp:105-133, encode_full) -- the
    stage on Z / chunk frames of chunk * Y * X voxels; only whole multiples (the reference's remainder path is not restated)"""
    a = np.ascontiguousarray(a)
    if a.ndim != 3:
        raise ValueError("frame_shuffle needs a 3D shape")
    if chunk != 1:
        if chunk < 1 or a.shape[0] % chunk:
            raise NotImplementedError("frame_shuffle: frames are no whole multiple of frame_chunk_size (encode_with_remainder is not restated)")
        out, dmap = frame_shuffle_encode(a.reshape(a.shape[0] // chunk, a.shape[1] * chunk, a.shape[2]), char=char)
        return out.reshape(a.shape), dmap
    out = np.empty_like(a)
    dmap = np.zeros(a.shape[0], dtype=np.uint64)
    if char:
        i8p = ctypes.POINTER(ctypes.c_int8)
        rc = lib().sqo_frame_shuffle_encode_i8(a.view(np.int8).ctypes.data_as(i8p), out.view(np.int8).ctypes.data_as(i8p),
                                               _shape3(a.shape), _ptr(dmap, _u64p))
    elif a.dtype == np.uint16:
        rc = lib().sqo_frame_shuffle_encode_u16(_ptr(a, _u16p), _ptr(out, _u16p), _shape3(a.shape), _ptr(dmap, _u64p))
    else:
        rc = lib().sqo_frame_shuffle_encode_u8(_ptr(a.view(np.uint8), _u8p), _ptr(out.view(np.uint8), _u8p),
                                               _shape3(a.shape), _ptr(dmap, _u64p))
    if rc:
        raise MemoryError
    return out, dmap


def base64_encode(raw):
    src = np.frombuffer(bytes(raw), dtype=np.uint8)
    dst = ctypes.create_string_buffer(4 * ((src.size + 2) // 3) + 4)
    n = lib().sqo_base64_encode(_ptr(src, _u8p), ctypes.c_size_t(src.size), dst)
    return dst.raw[:n].decode("ascii")


def to_verbatim(arr):
    """parsing::range_to_verbatim (string_parsers.hpp:507-534)"""
    raw = np.ascontiguousarray(arr).tobytes()
    if not raw:
        return ""
    return "<verbatim>" + base64_encode(raw) + "</verbatim>"


# ------------------------------------------------------------------------------------------------
# pipeline grammar (string_parsers.hpp:285-471) -- separators '->' ',' '=' with <verbatim> protection
# ------------------------------------------------------------------------------------------------
def _split_outside_verbatim(s, sep):
    out, cur, i = [], "", 0
    while i < len(s):
        if s.startswith("<verbatim>", i):
            j = s.find("</verbatim>", i)
            if j < 0:
                return []
            cur += s[i:j + len("</verbatim>")]
            i = j + len("</verbatim>")
            continue
        if s.startswith(sep, i):
            out.append(cur)
            cur = ""
            i += len(sep)
            continue
        cur += s[i]
        i += 1
    out.append(cur)
    return out


def parse_pairs(pipeline):
    """pipeline_parser::to_pairs (string_parsers.hpp:354-395)"""
    if not pipeline:
        return []
    pairs = []
    for major in _split_outside_verbatim(pipeline, "->"):
        d = major.find("(")
        if d < 0:
            pairs.append((major, ""))
        else:
            pairs.append((major[:d], major[d + 1:-1]))
    return pairs


def parse_minors(cfg):
    """pipeline_parser::minors (string_parsers.hpp:433-467)"""
    out = {}
    if not cfg:
        return out
    for item in _split_outside_verbatim(cfg, ","):
        d = item.find("=")
        if d < 0:
            out[item] = item
        else:
            out[item[:d]] = item[d + 1:] if d + 1 < len(item) else item
    return out


HEAD_FILTERS = ("diff3x3x1", "bitswap1", "bitshuffle", "remove_background", "rmbkrd_neighbor5x5x5", "rmestbkrd", "raster_reorder",
                "tile_shuffle", "frame_shuffle", "zcurve_reorder")
SINKS = ("pass_through", "quantiser", "lz4")
TAIL_FILTERS = ("diff3x3x1", "bitswap1", "bitshuffle", "lz4", "raster_reorder", "tile_shuffle", "frame_shuffle", "zcurve_reorder")


def can_be_built_from(pipeline):
    """dynamic_pipeline::can_be_built_from (dynamic_pipeline.hpp:177-226) against the reference's full
    stage lists (sqeazy_pipelines.hpp:31-77)."""
    pairs = parse_pairs(pipeline)
    majors = _split_outside_verbatim(pipeline, "->") if pipeline else []
    if len(majors) != len(pairs):
        return False
    found, sink_matched = 0, False
    for name, _ in pairs:
        if not sink_matched and name in HEAD_FILTERS:
            found += 1
            continue
        if name in SINKS:
            found += 1
            sink_matched = True
            continue
        if name in TAIL_FILTERS:
            found += 1
    if found != len(pairs) or not pairs:
        return False
    rebuilt = 2 * (len(pairs) - 1)
    for name, cfg in pairs:
        rebuilt += len(name) + ((2 + len(cfg)) if cfg else 0)
    return rebuilt == len(pipeline)


# ------------------------------------------------------------------------------------------------
# header (sqeazy_header.hpp:146-193): Boost.PropertyTree write_json pretty output + delimiter,
# left-padded with spaces to a multiple of sizeof(T)
# ------------------------------------------------------------------------------------------------
_TYPE_NAME = {np.dtype(np.uint8): "uint8", np.dtype(np.uint16): "uint16", np.dtype(np.int8): "int8"}


def _json_escape(s):
    out = []
    for ch in s:
        c = ord(ch)
        if ch == '"':
            out.append('\\"')
        elif ch == "\\":
            out.append("\\\\")
        elif ch == "/":
            out.append("\\/")
        elif ch == "\b":
            out.append("\\b")
        elif ch == "\f":
            out.append("\\f")
        elif ch == "\n":
            out.append("\\n")
        elif ch == "\r":
            out.append("\\r")
        elif ch == "\t":
            out.append("\\t")
        elif c < 0x20:
            out.append("\\u%04X" % c)
        else:
            out.append(ch)
    return "".join(out)


def header_pack(dtype, shape, pipename, payload_bytes, version=SQY_VERSION, headref=SQY_HEADREF):
    dtype = np.dtype(dtype)
    q = lambda s: '"' + _json_escape(str(s)) + '"'
    lines = ["{",
             '    "pipename": %s,' % q(pipename),
             '    "raw": {',
             '        "type": %s,' % q(_TYPE_NAME[dtype]),
             '        "rank": %s,' % q(len(shape)),
             '        "shape": {']
    dims = ['            "dim": %s' % q(int(d)) for d in shape]
    lines.append(",\n".join(dims))
    lines += ["        }",
              "    },",
              '    "encoded": {',
              '        "bytes": %s' % q(int(payload_bytes)),
              "    },",
              '    "sqy": {',
              '        "version": %s,' % q(version),
              '        "headref": %s' % q(headref),
              "    }",
              "}"]
    # (latin-1: one byte per character -- Boost's writer passes bytes >= 0x80 through as they are, tests/test_header_tag_impl.cpp:87)
    text = ("\n".join(lines) + "\n").encode("latin-1") + HEADER_END
    if len(text) % dtype.itemsize:
        text = b" " * (dtype.itemsize - len(text) % dtype.itemsize) + text
    return text


def header_unpack(blob):
    """returns dict(pipename, type, shape, bytes, size) -- sqeazy_header.hpp:294-344"""
    import json
    blob = bytes(blob)
    pos = blob.find(HEADER_END)
    if pos < 0:
        raise ValueError("no sqy header")
    text = blob[:pos].decode("latin-1")

    dims = []

    def hook(pairs):
        d = {}
        for k, v in pairs:
            if k == "dim":
                dims.append(int(v))
            d[k] = v
        return d
    tree = json.loads(text, object_pairs_hook=hook)
    return dict(pipename=tree["pipename"], type=tree["raw"]["type"], shape=tuple(dims),
                bytes=int(tree["encoded"]["bytes"]), size=pos + len(HEADER_END),
                version=tree["sqy"]["version"], headref=tree["sqy"]["headref"])


# ------------------------------------------------------------------------------------------------
# pipeline level (dynamic_pipeline.hpp:560-690)
# ------------------------------------------------------------------------------------------------
class _Stage:
    def __init__(self, name, cfg):
        self.name, self.cfg_in = name, cfg
        self.extra = None
        if name == "lz4":
            self.lz4 = Lz4Config(cfg)
        if name == "quantiser":
            self.cmap = parse_minors(cfg)
        if name == "frame_shuffle":
            m = parse_minors(cfg)
            self.chunk = int(m.get("frame_chunk_size", "1"))
            self.map = m.get("reorder_map", "") if "reorder_map" in m else ""
        if name == "raster_reorder":
            m = parse_minors(cfg)
            self.tile = int(m["tile_size"]) if "tile_size" in m else None      # None: 16 / sizeof(T), known at encode time
        if name == "zcurve_reorder":
            m = parse_minors(cfg)
            self.tile = int(m["tile_size"]) if "tile_size" in m else 2         # zcurve_reorder_scheme_impl.hpp:40-55
        if name == "tile_shuffle":
            m = parse_minors(cfg)
            self.tile = int(m["tile_size"]) if "tile_size" in m else 32        # tile_shuffle_scheme_impl.hpp:26-46
            self.map = m.get("reorder_map", "")
        if name == "bitshuffle":
            m = parse_minors(cfg)
            self.block = int(m["block_size"]) if "block_size" in m else 0      # bitshuffle_scheme_impl.hpp:44-58

    def config(self):
        if self.name == "bitswap1":
            return "num_bits_per_plane=1"
        if self.name in ("diff3x3x1", "pass_through"):
            return ""
        if self.name == "lz4":
            return self.lz4.config()
        if self.name == "quantiser":
            return ",".join("%s=%s" % kv for kv in sorted(self.cmap.items()))
        if self.name == "frame_shuffle":
            return "frame_chunk_size=%d,reorder_map=%s" % (self.chunk, self.map)
        if self.name in ("raster_reorder", "zcurve_reorder"):
            return "tile_size=%d" % self.tile
        if self.name == "tile_shuffle":
            return "tile_size=%d,reorder_map=%s" % (self.tile, self.map)
        if self.name == "bitshuffle":
            return "block_size=%d" % self.block
        raise NotImplementedError(self.name)

    def full_name(self):
        c = self.config()
        return self.name + ("(" + c + ")" if c else "")


def pipeline_name(stages):
    return "->".join(s.full_name() for s in stages)


def build_stages(pipeline, dtype=None):
    stages = [_Stage(n, c) for n, c in parse_pairs(pipeline)]
    if dtype is not None:
        seen_sink = False
        for s in stages:                                     # defaults that depend on the voxel type (behind the sink: char)
            if s.name == "raster_reorder" and s.tile is None:
                s.tile = 16 // (1 if seen_sink else np.dtype(dtype).itemsize)
            if s.name in SINKS:
                seen_sink = True
    return stages


def pipeline_max_encoded_size(pipeline, nbytes, dtype, nthreads=1):
    """dynamic_pipeline::max_encoded_size (dynamic_pipeline.hpp:866-890)"""
    dtype = np.dtype(dtype)
    stages = build_stages(pipeline, dtype)
    hdr = header_pack(dtype, (nbytes,), pipeline_name(stages), nbytes * dtype.itemsize)
    sizes = []
    for s in stages:
        if s.name == "lz4":
            sizes.append(s.lz4.max_encoded_size(nbytes, nthreads))
        elif s.name == "quantiser":
            sizes.append(nbytes * dtype.itemsize + 256 * dtype.itemsize)
        else:
            sizes.append(nbytes)
    # head chain / sink / tail chain each contribute the max over their members; a max over all is equal
    return 2 * len(hdr) + (max(sizes) if sizes else 0)


def pipeline_encode(pipeline, vol, nthreads=2):
    """Whole-blob oracle for the supported stage set; `nthreads` as the caller passes it to SQY_PipelineEncode_* after
    the reference's "<= 0 means all cores" rule (1 = the serial LZ4 layout, anything else the chunked one).
    `vol` is a uint8/uint16 ndarray in {z,y,x} order.  Returns bytes."""
    vol = np.ascontiguousarray(vol)
    if not can_be_built_from(pipeline):
        raise ValueError("invalid pipeline")
    stages = build_stages(pipeline, vol.dtype)
    dtype = vol.dtype
    cur = vol
    seen_sink = False
    payload = None

    def tail_view(x):
        """what a 3-D tail filter sees (dynamic_pipeline.hpp:658-666): the sink's char stream in the volume's shape when the sink wrote
        one byte per voxel, else {1, 1, bytes}"""
        x = np.ascontiguousarray(x).reshape(-1).view(np.uint8)
        return x.reshape(vol.shape) if x.size == vol.size else x.reshape(1, 1, -1)

    for s in stages:
        is_sink = (not seen_sink) and s.name in SINKS
        if s.name == "bitswap1":
            cur = bitswap1_encode(cur)
        elif s.name == "diff3x3x1":
            # behind a sink the stream is `char`; it keeps the volume's shape only when the sink wrote one byte per voxel
            # (dynamic_pipeline.hpp:658-666: otherwise {1, 1, bytes}, which diff3x3x1 cannot take)
            if seen_sink:
                cur = tail_view(cur)
                if cur.shape != vol.shape:
                    raise ValueError("diff3x3x1: shape outside the reference's defined behaviour")
            cur = diff3x3x1_encode(cur, char=seen_sink)
        elif s.name == "frame_shuffle":
            if seen_sink:
                cur = tail_view(cur)
            cur, dmap = frame_shuffle_encode(cur, char=seen_sink, chunk=s.chunk)
            s.map = to_verbatim(dmap)
        elif s.name == "raster_reorder":
            if seen_sink:                      # raster_reorder_scheme<char> (sqeazy_pipelines.hpp:64-77): a pure reorder of bytes
                cur = tail_view(cur)
            if s.tile is None:
                s.tile = 16 // cur.dtype.itemsize
            cur = raster_reorder(cur, s.tile)
        elif s.name == "pass_through":
            cur = np.ascontiguousarray(cur).reshape(-1).view(np.uint8)     # pass_through_scheme_impl.hpp:66-79: a copy, re-typed to bytes
        elif s.name == "zcurve_reorder":
            if seen_sink:
                cur = tail_view(cur)
            cur = zcurve_reorder(cur, s.tile)
        elif s.name == "tile_shuffle":
            if seen_sink:
                cur = tail_view(cur)
            cur, dmap = tile_shuffle_encode(cur, s.tile, char=seen_sink)
            s.map = to_verbatim(dmap)
        elif s.name == "bitshuffle":
            cur = bitshuffle(cur, s.block)
        elif s.name == "quantiser":
            cur, dec = quantiser_encode(cur, s.cmap.get("weighting_function", "none"))
            if "decode_lut_path" in s.cmap:                    # quantiser_scheme_impl.hpp:200-204: to the file INSTEAD of the header
                lut_to_file(s.cmap["decode_lut_path"], dec)
            else:
                s.cmap["decode_lut_string"] = to_verbatim(dec)
        elif s.name == "lz4":
            payload = lz4_encode(cur.reshape(-1).view(np.uint8), s.lz4, nthreads)
            cur = payload
        else:
            raise NotImplementedError(s.name)
        if is_sink:
            seen_sink = True
    body = cur.reshape(-1).view(np.uint8)
    hdr = header_pack(dtype, vol.shape, pipeline_name(stages), body.size)
    return hdr + body.tobytes()


def pipeline_decode(blob):
    """inverse for round-trip checks (dynamic_pipeline.hpp:740-846); lossy for quantiser."""
    h = header_unpack(blob)
    dtype = np.dtype(h["type"])
    stages = build_stages(h["pipename"])
    body = np.frombuffer(bytes(blob), dtype=np.uint8)[h["size"]:h["size"] + h["bytes"]]
    n = int(np.prod(h["shape"]))
    sink_idx = next((i for i, s in enumerate(stages) if s.name in SINKS), None)
    cur = body

    def tail_view(x):
        x = np.ascontiguousarray(x).reshape(-1).view(np.uint8)
        return x.reshape(h["shape"]) if x.size == n else x.reshape(1, 1, -1)

    # tail filters^-1 then sink^-1 then head filters^-1
    order = list(range(len(stages)))[::-1]
    cur_dtype = np.uint8
    for i in order:
        s = stages[i]
        after_sink = sink_idx is not None and i > sink_idx
        work_dtype = np.uint8 if (after_sink or (sink_idx is not None and stages[sink_idx].name == "quantiser"
                                                 and i > sink_idx)) else dtype
        if s.name == "lz4":
            elem = 1 if (i != sink_idx or True) else dtype.itemsize
            quant = sink_idx is not None and stages[sink_idx].name == "quantiser"
            nbytes = n if quant else n * dtype.itemsize
            cur = lz4_decode_frames(cur, nbytes)
        elif s.name == "bitswap1":
            t = np.uint8 if after_sink else dtype
            cur = bitswap1_decode(np.ascontiguousarray(cur).view(t))
        elif s.name == "diff3x3x1":
            if after_sink:
                cur = diff3x3x1_decode(np.ascontiguousarray(cur).view(np.uint8).reshape(h["shape"]), char=True)
            else:
                cur = diff3x3x1_decode(np.ascontiguousarray(cur).view(dtype).reshape(h["shape"]))
        elif s.name == "quantiser":
            import base64
            if "decode_lut_path" in s.cmap:                    # quantiser_scheme_impl.hpp:83-85 (constructor): the file wins
                dec = lut_from_file(s.cmap["decode_lut_path"], dtype)
            else:
                lut = s.cmap["decode_lut_string"]
                lut = lut[len("<verbatim>"):-len("</verbatim>")]
                dec = np.frombuffer(base64.b64decode(lut), dtype=dtype)
            cur = dec[np.ascontiguousarray(cur).view(np.uint8)]
        elif s.name == "raster_reorder":
            cur = raster_reorder(tail_view(cur) if after_sink else np.ascontiguousarray(cur).view(dtype).reshape(h["shape"]), s.tile, decode=True)
        elif s.name == "pass_through":
            pass
        elif s.name == "zcurve_reorder":
            cur = zcurve_reorder(tail_view(cur) if after_sink else np.ascontiguousarray(cur).view(dtype).reshape(h["shape"]), s.tile, decode=True)
        elif s.name == "tile_shuffle":
            import base64
            m = s.map[len("<verbatim>"):-len("</verbatim>")]
            dmap = np.frombuffer(base64.b64decode(m), dtype=np.uint64)
            cur = tile_shuffle_decode(tail_view(cur) if after_sink else np.ascontiguousarray(cur).view(dtype).reshape(h["shape"]), dmap, s.tile)
        elif s.name == "bitshuffle":
            t = np.uint8 if after_sink else dtype
            cur = bitshuffle(np.ascontiguousarray(cur).view(t), s.block, decode=True)
        elif s.name == "frame_shuffle":
            import base64
            m = s.map[len("<verbatim>"):-len("</verbatim>")]
            dmap = np.frombuffer(base64.b64decode(m), dtype=np.uint64)
            if after_sink:
                v = tail_view(cur)
            else:
                v = np.ascontiguousarray(cur).view(dtype).reshape(h["shape"])
            # (frames with equal metrics: the map names one frame several times and others not at all -- those the reference leaves as
            # its output buffer had them, frame_shuffle_utils.hpp:337-344; here, and in the product, they are zeros)
            if v.shape[0] % s.chunk:
                raise ValueError("frame_shuffle: frame_chunk_size does not divide the frames")
            units = v.reshape(v.shape[0] // s.chunk, -1)                       # frame_chunk_size frames per sort unit
            out = np.zeros_like(units)
            out[dmap.astype(np.int64)] = units
            cur = out.reshape(v.shape)
        else:
            raise NotImplementedError(s.name)
    return np.ascontiguousarray(cur).view(dtype).reshape(h["shape"])
