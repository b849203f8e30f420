#!/usr/bin/env python3
"""Generates tests/golden/: golden vectors that pin the oracle (and through it the HIP path) to the REAL thing.

Run in the build container (needs /root/reference -> oracle/_ref/libsqy_ref.so and the image's liblz4 1.9.3):
    python oracle/gen_golden.py

Sources of truth
  * LZ4 bytes      liblz4.so.1.9.3 driven with sqeazy's call sequence (oracle/ref_driver.cpp cites
                   encoders/lz4.hpp:103-113, lz4_utils.hpp:99-274)
  * bitswap1 u16   the reference's own SSE gather, sqeazy::detail::simd_segment_broadcast
                   (encoders/sse_utils.hpp:1365-1433), compiled in place from /root/reference
  * everything the reference cannot produce here (needs Boost: scalar bitswap, diff3x3x1, quantiser, frame_shuffle,
    header) is NOT in this file's "reference" section; those stages are pinned by the reference's own KATs restated in
    tests/test_oracle_reference_kats.py and carry "oracle" hashes here only as regression anchors.

Output
  tests/golden/golden.json     {case: {input: generator spec, stage: {sha256, bytes, source}}}
  tests/golden/*.bin           a few small raw input/expected pairs (data only)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref, sqy_oracle as o   # noqa: E402
from sqeazy_amd import synth             # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def gen_bytes(kind, n, seed):
    """byte streams; the same function is imported by the tests"""
    rng = np.random.default_rng(seed)
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    if kind == "ff":
        return np.full(n, 255, np.uint8)
    if kind == "random":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "2level":
        return rng.integers(0, 2, n, dtype=np.uint8)
    if kind == "8level":
        return rng.integers(0, 8, n, dtype=np.uint8)
    if kind == "ramp":
        return (np.arange(n) % 251).astype(np.uint8)
    if kind == "slowramp":
        return ((np.arange(n) // 7) % 256).astype(np.uint8)
    if kind == "sparse":
        s = np.zeros(n, np.uint8)
        k = n // 50
        if k:
            s[rng.integers(0, n, k)] = rng.integers(1, 255, k)
        return s
    if kind == "words":
        words = rng.integers(0, 256, (50, 12), dtype=np.uint8)
        return words[rng.integers(0, 50, n // 12 + 1)].reshape(-1)[:n]
    if kind == "period7":
        return np.tile(np.arange(7, dtype=np.uint8), n // 7 + 1)[:n]
    if kind == "farrep":                                   # repeats just inside the 64 KiB window, across block borders
        c = rng.integers(0, 256, n, dtype=np.uint8)
        for i in range(66000, n - 300, 66000):
            c[i:i + 300] = c[i - 65000:i - 65000 + 300]
        return c
    if kind == "rawmix":                                   # every other 256 KiB incompressible (stored raw), the rest zeros
        d = rng.integers(0, 256, n, dtype=np.uint8)
        for i in range(0, n, 2 * (256 << 10)):
            d[i:i + (256 << 10)] = 0
        return d
    if kind == "periodic":
        b = np.tile(rng.integers(0, 256, 7000, dtype=np.uint8), n // 7000 + 1)[:n].copy()
        b[::1531] ^= 1
        return b
    raise KeyError(kind)


def main():
    assert ref.available(), "oracle/_ref/libsqy_ref.so missing: run `make -C oracle` where /root/reference exists"
    assert ref.lz4_version() == 10903, "golden vectors are pinned to liblz4 1.9.3"
    os.makedirs(GOLD, exist_ok=True)
    G = {"_meta": {"liblz4": "1.9.3 (LZ4_versionNumber 10903)", "reference_pieces": "encoders/sse_utils.hpp simd_segment_broadcast",
                   "generator": "oracle/gen_golden.py"}, "lz4_block": [], "lz4_frames": [], "bitswap1_u16": [], "pipelines": []}

    # ---- LZ4 block level: LZ4_compress_fast_continue(fresh stream, cap n-1) ----
    kinds = ["zeros", "ff", "random", "2level", "8level", "ramp", "slowramp", "sparse", "words", "period7"]
    for kind in kinds:
        for n in (1, 12, 13, 14, 100, 1000, 10000, 65536, 70000, 262144):
            d = gen_bytes(kind, n, 1000 + n)
            r = ref.lz4_block(d)
            mine = o.lz4_block_compress(d)
            assert r == mine, ("oracle differs from liblz4", kind, n)
            G["lz4_block"].append({"kind": kind, "n": n, "seed": 1000 + n, "csize": 0 if r is None else len(r),
                                   "sha256": None if r is None else sha(r), "source": "liblz4-1.9.3"})

    # ---- frames, chunked layout: encode_parallel ----
    for kind in kinds:
        for n in (0, 5, 10000, 262144, 262145, 600000, 2 * 262144 + 10000):
            d = gen_bytes(kind, n, 2000 + n)
            if n == 0:
                continue
            r = ref.lz4_encode_parallel(d, nthreads=2)
            r4 = ref.lz4_encode_parallel(d, nthreads=4)
            assert np.array_equal(r, r4), "chunked layout must not depend on the thread count (>= 2)"
            mine = o.lz4_encode_chunked(d)
            assert np.array_equal(r, mine), ("oracle frames differ from liblz4", kind, n)
            G["lz4_frames"].append({"kind": kind, "n": n, "seed": 2000 + n, "bytes": int(r.size), "sha256": sha(r.tobytes()),
                                    "source": "liblz4-1.9.3 via sqeazy's encode_parallel call sequence"})
    # other block sizes (BD / HC bytes)
    for kb, bid in ((64, 4), (1024, 6)):
        d = gen_bytes("8level", 3 * (kb << 10) + 777, 77)
        r = ref.lz4_encode_parallel(d, chunk=kb << 10, block_id=bid, nthreads=2)
        cfg = o.Lz4Config("blocksize_kb=%d,framestep_kb=%d" % (kb, kb))
        mine = o.lz4_encode_chunked(d, cfg)
        assert np.array_equal(r, mine), kb
        G["lz4_frames"].append({"kind": "8level", "n": int(d.size), "seed": 77, "config": "blocksize_kb=%d,framestep_kb=%d" % (kb, kb), "bytes": int(r.size),
                                "sha256": sha(r.tobytes()), "source": "liblz4-1.9.3"})
    # ---- block-linked frames: encode_serial (nthreads == 1) and chunks of several blocks ----
    G["lz4_linked"] = []
    lkinds = ["zeros", "random", "2level", "8level", "ramp", "sparse", "words", "farrep", "rawmix", "periodic"]
    for kind in lkinds:
        for n in (300000, 2 * (1 << 20) + 12345):
            d = gen_bytes(kind, n, 3000 + n)
            for cfg, step, bid in (("", 256 << 10, 5), ("blocksize_kb=64", 256 << 10, 4), ("framestep_kb=1024", 1 << 20, 5),
                                   ("n_chunks_of_input=3", n // 3, 5), ("blocksize_kb=64,n_chunks_of_input=7", n // 7, 4)):
                c = o.Lz4Config(cfg)
                assert c.bytes_per_chunk(n) == min(step, n) and c.block_id == bid, (cfg, c.bytes_per_chunk(n), step)
                r1 = ref.lz4_encode_serial(d, framestep=c.bytes_per_chunk(n), block_id=bid)
                m1 = o.lz4_encode_serial(d, c)
                assert np.array_equal(r1, m1), ("oracle serial frame differs from liblz4", kind, n, cfg)
                r2 = ref.lz4_encode_parallel(d, chunk=c.bytes_per_chunk(n), block_id=bid, nthreads=2)
                m2 = o.lz4_encode_chunked(d, c)
                assert np.array_equal(r2, m2), ("oracle multi-block chunks differ from liblz4", kind, n, cfg)
                G["lz4_linked"].append({"kind": kind, "n": n, "seed": 3000 + n, "config": cfg,
                                        "serial_bytes": int(r1.size), "serial_sha256": sha(r1.tobytes()),
                                        "chunked_bytes": int(r2.size), "chunked_sha256": sha(r2.tobytes()),
                                        "source": "liblz4-1.9.3 via sqeazy's encode_serial / encode_parallel call sequences"})
    # LZ4F constants the reference's tests pin (tests/test_lz4_sandbox.cpp:387-430)
    G["_meta"]["LZ4F_compressBound_256k"] = int(ref.lz4f_compress_bound(262144))
    G["_meta"]["LZ4F_HEADER_SIZE_MAX"] = int(ref.lib().ref_lz4f_header_size_max())

    # ---- bitswap1 u16 through the reference's SSE kernel ----
    for name, arr in (("ramp128", np.arange(128, dtype=np.uint16)),
                      ("random_64k", np.random.default_rng(5).integers(0, 65536, 1 << 16, dtype=np.uint16)),
                      ("synth_16x64x64", synth.stack((16, 64, 64)).reshape(-1)),
                      ("lowbits_4096", np.random.default_rng(6).integers(0, 16, 4096, dtype=np.uint16))):
        r = ref.bitswap1_encode_u16(arr, nthreads=2)
        assert np.array_equal(r, o.bitswap1_encode(arr)), name
        G["bitswap1_u16"].append({"name": name, "len": int(arr.size), "sha256": sha(r.tobytes()), "source": "reference sse_utils.hpp"})

    # ---- whole pipelines: payload from liblz4 where the stage chain is reference-backed, oracle otherwise ----
    vols = {"synth_u16_32x64x64": synth.stack((32, 64, 64)), "synth_u16_24x100x52": synth.stack((24, 100, 52)),
            "synth_u8_48x64x96": synth.stack((48, 64, 96), np.uint8)}
    for vname, vol in vols.items():
        for pipe in ("bitswap1->lz4", "lz4", "diff3x3x1->bitswap1->lz4", "frame_shuffle->lz4", "quantiser->bitswap1->lz4"):
            if vol.dtype == np.uint8 and pipe.startswith(("quantiser", "diff")):
                continue
            blob = o.pipeline_encode(pipe, vol)
            entry = {"volume": vname, "pipeline": pipe, "bytes": len(blob), "sha256": sha(blob), "source": "oracle"}
            blob1 = o.pipeline_encode(pipe, vol, nthreads=1)
            entry["nthreads1_bytes"], entry["nthreads1_sha256"] = len(blob1), sha(blob1)
            if pipe in ("bitswap1->lz4", "lz4") and vol.dtype == np.uint16 and vol.size % 128 == 0:
                # fully reference-backed payload: reference SSE bitswap + liblz4 frames
                stream = ref.bitswap1_encode_u16(vol, 2).view(np.uint8) if pipe.startswith("bitswap1") else vol.reshape(-1).view(np.uint8)
                payload = ref.lz4_encode_parallel(stream, nthreads=2).tobytes()
                h = o.header_unpack(blob)
                assert blob[h["size"]:] == payload
                entry["payload_sha256"] = sha(payload)
                payload1 = ref.lz4_encode_serial(stream).tobytes()
                assert blob1[o.header_unpack(blob1)["size"]:] == payload1
                entry["nthreads1_payload_sha256"] = sha(payload1)
                entry["source"] = "payload: reference SSE bitswap + liblz4 1.9.3; header: oracle"
            G["pipelines"].append(entry)

    # ---- small raw fixtures (data only) ----
    d = gen_bytes("sparse", 10000, 42)
    d.tofile(os.path.join(GOLD, "sparse_10000.in.bin"))
    ref.lz4_encode_parallel(d, nthreads=2).tofile(os.path.join(GOLD, "sparse_10000.lz4frames.bin"))
    v = synth.stack((4, 16, 32))
    v.tofile(os.path.join(GOLD, "synth_u16_4x16x32.in.bin"))
    ref.bitswap1_encode_u16(v, 1).tofile(os.path.join(GOLD, "synth_u16_4x16x32.bitswap1.bin"))
    ref.lz4_encode_parallel(ref.bitswap1_encode_u16(v, 1).view(np.uint8), nthreads=2).tofile(
        os.path.join(GOLD, "synth_u16_4x16x32.bitswap1_lz4_payload.bin"))

    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "golden.json"), "with", sum(len(v) for k, v in G.items() if k != "_meta"), "vectors")


def headline():
    """tests/golden/headline.json: payload digests of the BASELINE headline stacks (configs[0] 256^3 and configs[1] 1024x1024x512 uint16,
    'bitswap1->lz4') produced by the REFERENCE pieces themselves -- the reference's SSE bit-plane gather (encoders/sse_utils.hpp:1365-1433,
    compiled in place) followed by liblz4 1.9.3 frames through sqeazy's encode_parallel call sequence (encoders/lz4_utils.hpp:193-274) --
    plus the slab 0 of the 2048^3 north_star volume.  The GPU tests and bench.py compare the HIP path's payload with these digests, so
    the headline is pinned to the reference and not only to the restated oracle.  (The sqy header in front of the payload is the oracle's:
    the reference's header code needs Boost.)"""
    assert ref.available() and ref.lz4_version() == 10903
    H = {"_meta": {"generator": "oracle/gen_golden.py --headline", "bitswap1": "reference simd_segment_broadcast (sse_utils.hpp:1365-1433), 16 threads",
                   "lz4": "liblz4 1.9.3 (LZ4_versionNumber 10903) via encode_parallel (lz4_utils.hpp:193-274), 256 KiB chunks",
                   "stack": "sqeazy_amd.synth.stack(shape, uint16[, frames z_offset.. of z_total])"}, "stacks": []}
    cases = (("C1 256x256x256 u16", (256, 256, 256), 0, None), ("C2 1024x1024x512 u16", (512, 1024, 1024), 0, None),
             ("north_star slab 0 of 2048^3 u16 (2048x2048x256)", (256, 2048, 2048), 0, 2048))
    for name, shape, z0, ztot in cases:
        vol = synth.stack(shape, np.uint16) if ztot is None else None
        if vol is None:
            # frames [z0, z0 + Z) of a (ztot, Y, X) volume without building the whole volume
            Z, Y, X = shape
            per = Y * X
            noise = synth._noise(z0 * per, Z * per, synth.SEED).reshape(Z, Y, X)
            sh = synth._shell(z0, Z, ztot, Y, X)
            vol = (100 + (noise >> 2) + sh * 6000).astype(np.uint16)
        planes = ref.bitswap1_encode_u16(vol, 16)
        payload = ref.lz4_encode_parallel(planes.view(np.uint8), nthreads=8)
        mine = o.pipeline_encode("bitswap1->lz4", vol)
        h = o.header_unpack(mine)
        assert mine[h["size"]:] == payload.tobytes(), ("oracle payload differs from the reference pieces", name)
        H["stacks"].append({"name": name, "shape_zyx": list(shape), "z_offset": z0, "z_total": ztot or shape[0], "pipeline": "bitswap1->lz4",
                            "voxels_sha256": sha(vol.tobytes()), "payload_bytes": int(payload.size), "payload_sha256": sha(payload.tobytes()),
                            "blob_bytes": len(mine), "blob_sha256": sha(mine), "header_bytes": h["size"],
                            "source": "payload: reference SSE bitswap + liblz4 1.9.3; header: oracle"})
        print(name, payload.size, H["stacks"][-1]["payload_sha256"][:16], flush=True)
        del vol, planes, payload, mine
    with open(os.path.join(GOLD, "headline.json"), "w") as f:
        json.dump(H, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "headline.json"))


def headline_serial():
    """adds to tests/golden/headline.json the digests of the same stacks in the layout every default caller of the reference gets
    (nthreads = 1: ONE block-linked frame, lz4_utils.hpp:99-173): reference SSE bit-plane gather + liblz4 1.9.3 LZ4F_compressUpdate
    sequence; asserts the oracle's serial payload equals it on the way"""
    assert ref.available() and ref.lz4_version() == 10903
    path = os.path.join(GOLD, "headline.json")
    with open(path) as f:
        H = json.load(f)
    H["_meta"]["lz4_serial"] = "liblz4 1.9.3 via encode_serial (lz4_utils.hpp:99-173): one block-linked frame, one LZ4F_compressUpdate per framestep"
    for st in H["stacks"]:
        shape, z0, ztot = tuple(st["shape_zyx"]), st["z_offset"], st["z_total"]
        if ztot == shape[0]:
            vol = synth.stack(shape, np.uint16)
        else:
            Z, Y, X = shape
            per = Y * X
            noise = synth._noise(z0 * per, Z * per, synth.SEED).reshape(Z, Y, X)
            sh = synth._shell(z0, Z, ztot, Y, X)
            vol = (100 + (noise >> 2) + sh * 6000).astype(np.uint16)
        assert sha(vol.tobytes()) == st["voxels_sha256"]
        planes = ref.bitswap1_encode_u16(vol, 16).view(np.uint8)
        cfg = o.Lz4Config("")
        payload = ref.lz4_encode_serial(planes, framestep=cfg.bytes_per_chunk(planes.size))
        mine = o.pipeline_encode("bitswap1->lz4", vol, nthreads=1)
        h = o.header_unpack(mine)
        assert mine[h["size"]:] == payload.tobytes(), ("oracle serial payload differs from the reference pieces", st["name"])
        st["serial"] = {"payload_bytes": int(payload.size), "payload_sha256": sha(payload.tobytes()), "blob_bytes": len(mine), "blob_sha256": sha(mine),
                        "header_bytes": h["size"]}
        print(st["name"], "serial", payload.size, st["serial"]["payload_sha256"][:16], flush=True)
        del vol, planes, payload, mine
    with open(path, "w") as f:
        json.dump(H, f, indent=1, sort_keys=True)
    print("wrote", path)


def _slab_volume(shape, z0, ztot, frames_per_step=16):
    """frames [z0, z0 + Z) of a (ztot, Y, X) synthetic uint16 volume, built a few frames at a time (1 Gi voxels at once would need ~25 GiB)"""
    Z, Y, X = shape
    per = Y * X
    vol = np.empty(shape, dtype=np.uint16)
    for a in range(0, Z, frames_per_step):
        nz = min(frames_per_step, Z - a)
        noise = synth._noise((z0 + a) * per, nz * per, synth.SEED).reshape(nz, Y, X)
        sh = synth._shell(z0 + a, nz, ztot, Y, X)
        vol[a:a + nz] = (100 + (noise >> 2) + sh * 6000).astype(np.uint16)
    return vol


def headline_slabs():
    """tests/golden/headline_slabs.json (round 6): every z-slab a rank of the sharded BASELINE volumes encodes, at
    FULL plane size -- the shell sweeps through z, slabs 3 and 4 carry data (12-bit shell voxels over whole frames) that slab 0 never sees.
      * north_star 2048^3 u16 'bitswap1->lz4': all eight 2048 x 2048 x 256 slabs; payload = the REFERENCE pieces (SSE bit-plane gather +
        liblz4 1.9.3 through encode_parallel), the oracle's blob asserted equal on the way;
      * configs[2] 2048^3 'diff3x3x1->bitswap1->lz4', slabs 3 and 4, and configs[4] 2048 x 2048 x 1024 'quantiser->bitswap1->lz4', slabs 1
        and 2 of its four (the two around the shell's centre): the ORACLE's blob (those stages of the reference need Boost: unbuildable here).
    One slab at a time; resumes from what the file already holds."""
    assert ref.available() and ref.lz4_version() == 10903
    path = os.path.join(GOLD, "headline_slabs.json")
    H = {"_meta": {"generator": "oracle/gen_golden.py --headline-slabs",
                   "north_star": "payload: reference simd_segment_broadcast (sse_utils.hpp:1365-1433) + liblz4 1.9.3 via encode_parallel (lz4_utils.hpp:193-274); header: oracle",
                   "c3_c5": "blob: oracle (oracle/sqy_oracle.c / sqy_oracle_float.c); the reference's diff / quantiser stages include Boost and cannot be built here",
                   "stack": "frames [z_offset, z_offset + 256) of sqeazy_amd.synth's (z_total, 2048, 2048) uint16 volume"}, "slabs": []}
    if os.path.exists(path):
        with open(path) as f:
            H = json.load(f)
    have = {(e["pipeline"], e["z_total"], e["z_offset"]) for e in H["slabs"]}
    shape = (256, 2048, 2048)
    todo = [("bitswap1->lz4", 2048, s) for s in range(8)] + [("diff3x3x1->bitswap1->lz4", 2048, s) for s in (3, 4)] + \
           [("quantiser->bitswap1->lz4", 1024, s) for s in (1, 2)]
    for pipeline, ztot, slab in todo:
        z0 = 256 * slab
        if (pipeline, ztot, z0) in have:
            continue
        vol = _slab_volume(shape, z0, ztot)
        mine = o.pipeline_encode(pipeline, vol)
        h = o.header_unpack(mine)
        e = {"name": "%s slab %d of %d (2048x2048x256 of %dx2048x2048)" % (pipeline, slab, ztot // 256, ztot), "pipeline": pipeline, "slab": slab,
             "shape_zyx": list(shape), "z_offset": z0, "z_total": ztot, "voxels_sha256": sha(vol.tobytes()), "blob_bytes": len(mine),
             "blob_sha256": sha(mine), "header_bytes": h["size"], "payload_bytes": len(mine) - h["size"], "payload_sha256": sha(mine[h["size"]:])}
        if pipeline == "bitswap1->lz4":
            planes = ref.bitswap1_encode_u16(vol, 8)
            payload = ref.lz4_encode_parallel(planes.view(np.uint8), nthreads=8)
            assert mine[h["size"]:] == payload.tobytes(), ("oracle payload differs from the reference pieces", e["name"])
            e["source"] = "payload: reference SSE bitswap + liblz4 1.9.3 (oracle blob asserted equal); header: oracle"
            del planes, payload
        else:
            e["source"] = "oracle"
        H["slabs"].append(e)
        print(e["name"], e["payload_bytes"], e["payload_sha256"][:16], flush=True)
        with open(path, "w") as f:
            json.dump(H, f, indent=1, sort_keys=True)
        del vol, mine
    print("wrote", path)


def accel():
    """tests/golden/accel.json: liblz4 1.9.3 with an acceleration above 1 -- sqeazy's lz4(accel=-k), a negative LZ4F compression level
    (encoders/lz4.hpp:103-113) -- block level, the chunked layout and the serial block-linked layout; asserts oracle == liblz4 on the way"""
    assert ref.available() and ref.lz4_version() == 10903
    A = {"_meta": {"generator": "oracle/gen_golden.py --accel", "liblz4": "1.9.3 (LZ4_versionNumber 10903)",
                   "mapping": "sqeazy accel = LZ4F compressionLevel; level -k -> LZ4_compress_fast_continue acceleration k + 1 (lz4frame.c), capped at 65537"},
         "cases": []}
    for kind in ["zeros", "random", "2level", "8level", "ramp", "slowramp", "sparse", "words", "period7", "farrep", "rawmix", "periodic"]:
        for n in (10000, 262144, 600000):
            d = gen_bytes(kind, n, 4000 + n)
            for level in (-1, -3, -64, -70000):
                a = o.lz4_acceleration(level)
                rb = ref.lz4_block(d, accel=a)
                assert rb == o.lz4_block_compress(d, acceleration=a), ("block", kind, n, level)
                cfg = o.Lz4Config("accel=%d" % level)
                rc = ref.lz4_encode_parallel(d, accel=level, nthreads=2)
                assert np.array_equal(rc, o.lz4_encode_chunked(d, cfg)), ("chunked", kind, n, level)
                rs = ref.lz4_encode_serial(d, framestep=cfg.bytes_per_chunk(n), accel=level)
                assert np.array_equal(rs, o.lz4_encode_serial(d, cfg)), ("serial", kind, n, level)
                A["cases"].append({"kind": kind, "n": n, "seed": 4000 + n, "accel": level, "acceleration": a,
                                   "block_csize": 0 if rb is None else len(rb), "block_sha256": None if rb is None else sha(rb),
                                   "chunked_bytes": int(rc.size), "chunked_sha256": sha(rc.tobytes()),
                                   "serial_bytes": int(rs.size), "serial_sha256": sha(rs.tobytes()), "source": "liblz4-1.9.3"})
    with open(os.path.join(GOLD, "accel.json"), "w") as f:
        json.dump(A, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "accel.json"), "with", len(A["cases"]), "cases")


if __name__ == "__main__":
    if "--accel" in sys.argv:
        accel()
    elif "--headline-slabs" in sys.argv:
        headline_slabs()
    elif "--headline-serial" in sys.argv:
        headline_serial()
    elif "--headline" in sys.argv:
        headline()
        headline_serial()
    else:
        main()
        accel()
        headline()
        headline_serial()
