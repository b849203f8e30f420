#!/usr/bin/env python3
"""bench.py -- encode GB/s (input voxels) for the 'bitswap1->lz4' uint16 pipeline on N x MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W.  For N > 1 the driver starts one rank per GPU with
torch.distributed.run; called WITHOUT that launcher and N > 1 this script starts the ranks itself (a torchrun child
process, before anything here touches the GPU) and exits with the child's code.

A "step" is one pass of the hot path over one z-slab: every rank encodes a 1024x1024x512 uint16 synthetic stack
(BASELINE.json configs[1]; 1 GiB, already resident in HBM) with ONE C-ABI call.  Weak scaling: per-GPU work is fixed.
W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides, MAX
over ranks per block; blocks are repeated until >= 1 s has been timed and the MEDIAN block is reported (`ms_per_step`,
`value`; min / max next to it).  value = (N * input bytes * K) / block time, GB = 1e9 bytes.

N > 1: slabs are independent sqeazy blobs.  `value` is measured with the blobs left on the GPUs that made them (a sharded
container; only the 8-byte sizes -- its index -- are all_gathered over RCCL per step).  `with_gather` is the same run with
north_star's final RCCL gather of the compressed slabs to rank 0 inside every step, posted from a gather thread so that it
overlaps the next steps' encodes (double-buffered root ingress).

The JSON line also carries
  roofline      the dominant kernel (largest share of device time, HIP events on the launch stream inside the timed
                region) priced at the path's ALGORITHMIC bytes per call (2 B per voxel read once + payload bytes written
                once: SURVEY.md 8(d)) against 8 TB/s; `alone_*` = the same with one call at a time; `traffic` = PMC bytes
                per launch from the committed rocprofv3 passes of THIS library build (profiles/, matched by sha256), else null
  cpu_baseline  the reference's own SSE bit-plane gather + liblz4 1.9.3 frames (oracle/_ref) on the host cores, the two
                C calls alone on the full 1 GiB stack: one thread and all cores
  host_abi      the same stack through the reference-protocol entry point SQY_PipelineEncode_UI16 (host pointers, PCIe)
  config.secondary   device time and roofline fraction of the other BASELINE configs and of north_star's target
                (2048^3 uint16 as 8 sequential 2048x2048x256 slab calls on this GPU); N = 1 only
  build         sha256 and compile flags of libsqeazy_amd.so
"""
import argparse
import ctypes
import hashlib
import json
import os
import queue
import socket
import statistics
import subprocess
import sys
import threading
import time

import numpy as np

# HIP deals streams to hardware queues in creation order, FOUR queues by default: with the default stream in use that leaves three
# caller streams a queue of their own, and a fourth caller shares one (its kernels never overlap the other's: measured 1250 GB/s
# with four calls in flight against 1515 with three).  Eight queues, set before the HIP runtime starts: four calls in flight,
# 1660 GB/s.  A deployment setting like the caller's thread count; the runtime reads it once, at its first call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PIPELINE = "bitswap1->lz4"
SHAPE = (512, 1024, 1024)          # {z,y,x}: 1024x1024x512 voxels, uint16 -> 1 GiB
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SLABS_INFLIGHT = 3                 # slab calls in flight inside one Slabs call (--slabs-inflight)


def lib_identity():
    import sqeazy_amd
    from sqeazy_amd import build
    h = hashlib.sha256()
    with open(sqeazy_amd.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return {"library": os.path.relpath(sqeazy_amd.LIB_PATH, ROOT), "sha256": h.hexdigest(),
            "hipcc_flags": "--offload-arch=%s %s" % (build.ARCH, " ".join(build.FLAGS))}


def measured_traffic(sha, kernel, launches_per_call):
    """PMC bytes per launch of `kernel` from the newest profiles/*_pmc_hbm.json that was collected on this very build, and the sum over
    the kernels of ONE timed encode call (`launches_per_call`: kernel -> launches per call in the timed region; the PMC file of a full
    bench run also holds the decode and serial-layout kernels of the other legs, which are no part of the step)"""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_pmc_hbm.json"):
            try:
                j = json.load(open(os.path.join(pdir, name)))
            except Exception:
                continue
            if j.get("library_sha256") == sha and kernel in j.get("traffic", {}):
                # (one timing scope of the library may cover several small kernels: their names in the rocprofv3 summary)
                alias = {"lz4_dedupe": ["lz4_dedupe_clear", "lz4_dedupe_key"], "lz4_frame_scan": ["lz4_frame_scan", "lz4_tail_marks"],
                         "lz4_frame_gather": ["lz4_stash_raw", "lz4_frame_gather", "lz4_inplace_finish"], "lz4_inplace_tail": ["lz4_inplace_tail_fused"]}
                per_call, missing = 0.0, []
                for k, n in launches_per_call.items():
                    names = [x for x in alias.get(k, [k]) if x in j["traffic"]]
                    if not names:
                        missing.append(k)
                    per_call += n * sum(j["traffic"][x] for x in names)
                best = {"bytes_per_launch": j["traffic"][kernel], "per_call_all_kernels": int(round(per_call)),
                        "kernels_missing": sorted(missing), "source": "profiles/" + name}
    return best


def cpu_baseline(vol_host):
    """the reference's SSE bit-plane gather + liblz4 frames on the host, the two C calls alone (no Python staging in the timed part)"""
    from oracle import ref, sqy_oracle
    cores = os.cpu_count() or 1
    flat = np.ascontiguousarray(vol_host).reshape(-1)
    nbytes = flat.nbytes
    if not ref.available():
        t0 = time.perf_counter()
        planes = sqy_oracle.bitswap1_encode_planes(flat, nthreads=min(cores, 16))
        enc = sqy_oracle.lz4_encode_chunked(planes.view(np.uint8))
        dt = time.perf_counter() - t0
        return {"value": round(nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": min(cores, 16), "kind": "port",
                "sample": "full %d MiB stack, C restatement (oracle/_ref not loadable)" % (nbytes >> 20), "payload_ratio": round(nbytes / enc.size, 3)}
    L = ref.lib()
    u16p, chp = ctypes.POINTER(ctypes.c_uint16), ctypes.c_char_p
    # 16-byte aligned source / plane buffers, output sized as encode_parallel wants it; all touched before the clock starts
    raw = np.zeros(flat.size + 8, np.uint16)
    src = raw[(-raw.ctypes.data % 16) // 2:][:flat.size]
    src[:] = flat
    planes = np.zeros(flat.size, np.uint16)
    chunk = 256 << 10
    stride = L.ref_lz4f_compress_bound(ctypes.c_size_t(chunk), ctypes.c_int(1), ctypes.c_int(5)) + 19
    cap = ((nbytes + chunk - 1) // chunk) * stride + 64
    dst = np.zeros(cap, np.uint8)
    rows = {}
    for label, nt_bsw, nt_lz4, reps in (("1_thread", 1, 1, 1), ("all_cores", min(cores, 16), cores, 3)):
        best, nout = None, 0
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = L.ref_bitswap1_encode_u16(src.ctypes.data_as(u16p), planes.ctypes.data_as(u16p), ctypes.c_size_t(flat.size), ctypes.c_int(nt_bsw))
            t1 = time.perf_counter()
            # one thread: the layout a 1-thread reference run produces (encode_serial); else encode_parallel
            if nt_lz4 == 1:
                nout = L.ref_lz4_encode_serial(planes.ctypes.data_as(chp), ctypes.c_size_t(nbytes), dst.ctypes.data_as(chp), ctypes.c_size_t(cap),
                                               ctypes.c_size_t(chunk), ctypes.c_int(1), ctypes.c_int(5))
            else:
                nout = L.ref_lz4_encode_parallel(planes.ctypes.data_as(chp), ctypes.c_size_t(nbytes), dst.ctypes.data_as(chp), ctypes.c_size_t(cap),
                                                 ctypes.c_size_t(chunk), ctypes.c_int(1), ctypes.c_int(5), ctypes.c_int(nt_lz4))
            t2 = time.perf_counter()
            if rc or not nout:
                raise RuntimeError("reference pieces failed (rc %d, %d bytes)" % (rc, nout))
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        rows[label] = {"value": round(nbytes / best[0] / 1e9, 4), "bitswap_s": round(best[1], 3), "lz4_s": round(best[2], 3),
                       "threads": nt_lz4, "payload_bytes": int(nout)}
    return {"value": rows["all_cores"]["value"], "unit": "GB/s", "cores": cores, "kind": "reference",
            "sample": "the full %dx%dx%d uint16 bench stack (%d MiB); timed: simd_segment_broadcast (reference SSE, <= 16 threads) + "
                      "liblz4 1.9.3 frames through sqeazy's call sequence, the two C calls only; best of 3 (all cores), 1 run (1 thread)" % (
                          SHAPE[2], SHAPE[1], SHAPE[0], nbytes >> 20),
            "one_thread": rows["1_thread"], "all_cores": rows["all_cores"]}


def host_abi(vol_host, shape):
    """SQY_PipelineEncode_UI16 on caller memory (pageable, touched): H2D + kernels + D2H, as a C caller sees it"""
    import sqeazy_amd
    L = sqeazy_amd.lib()
    cap = sqeazy_amd.max_compressed_length(PIPELINE, shape, np.uint16)
    dst = np.zeros(cap, np.uint8)
    shp = (ctypes.c_long * 3)(*shape)
    n = ctypes.c_long(0)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        rc = L.SQY_PipelineEncode_UI16(PIPELINE.encode(), ctypes.c_void_p(vol_host.ctypes.data), shp, 3, ctypes.c_void_p(dst.ctypes.data), ctypes.byref(n), 0)
        dt = time.perf_counter() - t0
        if rc:
            raise RuntimeError("SQY_PipelineEncode_UI16 returned %d" % rc)
        best = dt if best is None else min(best, dt)
    res = {"value": round(vol_host.nbytes / best / 1e9, 3), "unit": "GB/s", "ms_per_call": round(best * 1e3, 2),
           "entry_point": "SQY_PipelineEncode_UI16 (host pointers: %.2f GB up, kernels, %.2f GB down), best of 4" % (vol_host.nbytes / 1e9, n.value / 1e9)}
    # what a caller of the reference gets who changes nothing: nthreads = 1 (one block-linked frame) and SQY_Decode_UI16 of that blob
    try:
        best1 = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.SQY_PipelineEncode_UI16(PIPELINE.encode(), ctypes.c_void_p(vol_host.ctypes.data), shp, 3, ctypes.c_void_p(dst.ctypes.data), ctypes.byref(n), 1)
            dt = time.perf_counter() - t0
            if rc:
                raise RuntimeError("SQY_PipelineEncode_UI16(nthreads = 1) returned %d" % rc)
            best1 = dt if best1 is None else min(best1, dt)
        back = np.empty_like(vol_host)
        bestd = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.SQY_Decode_UI16(ctypes.c_void_p(dst.ctypes.data), ctypes.c_long(n.value), ctypes.c_void_p(back.ctypes.data), ctypes.c_int(1))
            dt = time.perf_counter() - t0
            if rc:
                raise RuntimeError("SQY_Decode_UI16 returned %d" % rc)
            bestd = dt if bestd is None else min(bestd, dt)
        res["nthreads_1"] = {"encode_value": round(vol_host.nbytes / best1 / 1e9, 3), "encode_ms": round(best1 * 1e3, 2),
                             "decode_value": round(vol_host.nbytes / bestd / 1e9, 3), "decode_ms": round(bestd * 1e3, 2), "unit": "GB/s",
                             "round_trip_equal": bool(np.array_equal(back, vol_host)),
                             "what": "SQY_PipelineEncode_UI16(.., nthreads = 1) and SQY_Decode_UI16 of its blob, host pointers, best of 3"}
    except Exception as e:   # reported, never required
        res["nthreads_1"] = {"error": repr(e)}
    return res


def host_abi_two_calls(vol_host, shape):
    """Two SQY_PipelineEncode_UI16 calls in flight from two host threads (the reference's entry points are re-entrant).  Inside ONE call the two
    transfers cannot overlap -- the blob starts at dst[0], its stored tail (99.5 % of it) sits behind the compressed head, whose size is known
    only when the whole volume has been uploaded and parsed (every bit plane spans all voxels) -- but the upload of one call overlaps the
    download of the other (PCIe is full duplex): the rate a caller with two buffers gets.  Run LAST: the second context's streams (its own and
    its four copy lanes') take hardware queues, and kernels of later legs that are meant to overlap then share one (the decode's copy of the
    stored frames next to its LZ4 kernel: 0.63 -> 0.83 ms)."""
    import sqeazy_amd
    L = sqeazy_amd.lib()
    cap = sqeazy_amd.max_compressed_length(PIPELINE, shape, np.uint16)
    shp = (ctypes.c_long * 3)(*shape)
    try:
        dst = np.zeros(cap, np.uint8)
        vol2 = vol_host.copy()
        dst2 = np.zeros(cap, np.uint8)
        errs = []

        def one(v, d, reps):
            nn = ctypes.c_long(0)
            for _ in range(reps):
                if L.SQY_PipelineEncode_UI16(PIPELINE.encode(), ctypes.c_void_p(v.ctypes.data), shp, 3, ctypes.c_void_p(d.ctypes.data), ctypes.byref(nn), 0):
                    errs.append("rc")
        best2 = None
        for _ in range(2):
            ths = [threading.Thread(target=one, args=(v, d, 3)) for v, d in ((vol_host, dst), (vol2, dst2))]
            t0 = time.perf_counter()
            [t.start() for t in ths]
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            best2 = dt if best2 is None else min(best2, dt)
        if errs:
            raise RuntimeError("SQY_PipelineEncode_UI16 failed with two calls in flight")
        return {"value": round(6 * vol_host.nbytes / best2 / 1e9, 3), "unit": "GB/s", "ms_per_call": round(best2 / 6 * 1e3, 2),
                "what": "two host threads, three SQY_PipelineEncode_UI16 calls each on buffers of their own, aggregate; best of 2"}
    except Exception as e:   # reported, never required
        return {"error": repr(e)}


def secondary_configs(dev):
    """device time of one call for the other BASELINE configs and north_star's target, roofline fraction of the whole call"""
    import torch
    import sqeazy_amd
    from sqeazy_amd import synth
    out = {}

    def one(pipeline, shape, dtype, algo_per_voxel, vol=None, reps=3, extra=0, decode=False, nthreads=0):
        v = vol if vol is not None else synth.stack_torch(shape, dtype, dev)
        cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype) + extra
        buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        rc, off, m = sqeazy_amd.encode_device_at(pipeline, v.data_ptr(), shape, dtype, buf.data_ptr(), cap, nthreads=nthreads)
        if rc:
            raise RuntimeError("%s returned %d" % (pipeline, rc))
        best, prof = None, {}
        for _ in range(reps):
            sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc, off, m = sqeazy_amd.encode_device_at(pipeline, v.data_ptr(), shape, dtype, buf.data_ptr(), cap, nthreads=nthreads)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            sqeazy_amd.profile_enable(False)
            if best is None or dt < best:
                best, prof = dt, sqeazy_amd.profile_get()
        nvox = int(np.prod(shape))
        hdr = sqeazy_amd.header_size(bytes(buf[off:off + 65536].cpu().numpy().tobytes()))
        algo = algo_per_voxel * nvox + (m - hdr)
        res = {"ms_per_call": round(best * 1e3, 3), "input_GBps": round(nvox * np.dtype(dtype).itemsize / best / 1e9, 1),
               "algorithmic_bytes": int(algo), "roofline_frac": round(algo / best / 1e9 / HBM_PEAK_GBS, 5), "blob_bytes": int(m),
               "kernels_ms": {k: round(a / max(c, 1), 3) for k, (a, c) in prof.items()}}
        if decode:
            # .. and back (SQYAMD_Decode_*_Device on that blob, device to device), for the record
            try:
                nb = nvox * np.dtype(dtype).itemsize
                back = torch.empty(nb, dtype=torch.uint8, device=dev)
                dfn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device if np.dtype(dtype) == np.uint16 else sqeazy_amd.lib().SQYAMD_Decode_UI8_Device
                dbest, dprof = None, {}
                for _ in range(3):
                    sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    drc = dfn(ctypes.c_void_p(buf.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    sqeazy_amd.profile_enable(False)
                    if drc:
                        raise RuntimeError("decode returned %d" % drc)
                    if dbest is None or dt < dbest:
                        dbest, dprof = dt, sqeazy_amd.profile_get()
                # algorithmic bytes of a decode: the blob read once, the volume written once
                res["decode"] = {"ms_per_call": round(dbest * 1e3, 3), "output_GBps": round(nb / dbest / 1e9, 1),
                                 "algorithmic_bytes": int(m + nb), "roofline_frac": round((m + nb) / dbest / 1e9 / HBM_PEAK_GBS, 5),
                                 "kernels_ms": {k: round(a / max(c, 1), 3) for k, (a, c) in dprof.items()}}
                if "quantiser" not in pipeline and "frame_shuffle" not in pipeline:
                    res["decode"]["round_trip_equal"] = bool((back.view(torch.uint16 if np.dtype(dtype) == np.uint16 else torch.uint8).reshape(shape) == v).all().item())
                del back
            except Exception as e:   # reported, never required
                res["decode"] = {"error": repr(e)}
        del buf
        if vol is None:
            del v
        torch.cuda.empty_cache()
        return res, best

    # the bench stack with every voxel raised by 40000: the upper bit planes are no longer all zero (planes 15 and 12..10 are all
    # ones, 14 and 13 zero), so the transposes' "holes" -- all-zero 1 KiB pieces that are never written -- cover two planes instead
    # of five (round-3 advice: a number on data whose upper planes are not zero)
    try:
        v40 = synth.stack_torch(SHAPE, np.uint16, dev)
        v40.view(torch.int16).add_(40000 - 65536)            # (+ 40000 mod 2^16; torch has no uint16 add)
        out["C2 stack + 40000 (upper bit planes not zero) 1024x1024x512 u16 bitswap1->lz4"], _ = one(PIPELINE, SHAPE, np.uint16, 2, vol=v40)
        del v40
        torch.cuda.empty_cache()
    except Exception as e:   # reported, never required
        out["C2 stack + 40000"] = {"error": repr(e)}
    try:
        out["C2 1024x1024x512 u16 bitswap1->lz4 (the default run's stack, one call at a time, with its decode)"], _ = one(PIPELINE, SHAPE, np.uint16, 2, decode=True)
    except Exception as e:   # reported, never required
        out["C2 with decode"] = {"error": repr(e)}
    out["C3_slab 2048x2048x256 u16 diff3x3x1->bitswap1->lz4"], _ = one("diff3x3x1->bitswap1->lz4", (256, 2048, 2048), np.uint16, 2, decode=True)
    out["C4 1024x1024x1024 u8 frame_shuffle->lz4"], _ = one("frame_shuffle->lz4", (1024, 1024, 1024), np.uint8, 2, extra=1 << 16, decode=True)
    out["C5_slab 2048x2048x256 u16 quantiser->bitswap1->lz4"], _ = one("quantiser->bitswap1->lz4", (256, 2048, 2048), np.uint16, 4, decode=True)
    # the same slabs in the layout every unchanged caller of the reference asks for (nthreads = 1: ONE block-linked LZ4 frame; the HDF5 filter
    # and the sqy tool default to it), so that a regression of that layout shows in the line (round-4 advice)
    for key, pipeline, apv in (("C3_slab serial layout (nthreads = 1) 2048x2048x256 u16 diff3x3x1->bitswap1->lz4", "diff3x3x1->bitswap1->lz4", 2),
                               ("C5_slab serial layout (nthreads = 1) 2048x2048x256 u16 quantiser->bitswap1->lz4", "quantiser->bitswap1->lz4", 4)):
        try:
            out[key], _ = one(pipeline, (256, 2048, 2048), np.uint16, apv, reps=1, decode=True, nthreads=1)
        except Exception as e:   # reported, never required
            out[key] = {"error": repr(e)}
    # north_star's target on one GPU: 2048^3 uint16, bitswap1->lz4, eight sequential 2 GiB slab calls (inputs resident when each call starts)
    total_t, total_algo, total_in, slabs = 0.0, 0, 0, []
    for i in range(8):
        v = synth.stack_torch((256, 2048, 2048), np.uint16, dev, z_offset=256 * i, z_total=2048)
        r, t = one(PIPELINE, (256, 2048, 2048), np.uint16, 2, vol=v, reps=2)
        del v
        torch.cuda.empty_cache()
        total_t += t; total_algo += r["algorithmic_bytes"]; total_in += 2 * 256 * 2048 * 2048
        slabs.append(r["ms_per_call"])
    out["north_star 2048^3 u16 bitswap1->lz4, 8 sequential 2048x2048x256 calls"] = {
        "ms_total": round(total_t * 1e3, 2), "ms_per_slab": slabs, "input_GBps": round(total_in / total_t / 1e9, 1),
        "algorithmic_bytes": int(total_algo), "roofline_frac": round(total_algo / total_t / 1e9 / HBM_PEAK_GBS, 5)}
    # a whole volume as z-slab blobs with ONE C call (SQYAMD_PipelineEncode_Slabs_UI16_Device: three slab calls in flight on
    # library-owned streams; every slab resident, like the default run)
    def volume_in_flight(pipeline, nslabs, z_total, algo_per_voxel, key):
        try:
            vol = torch.empty((nslabs * 256, 2048, 2048), dtype=torch.uint16, device=dev)
            for i in range(nslabs):
                vol[256 * i:256 * (i + 1)] = synth.stack_torch((256, 2048, 2048), np.uint16, dev, z_offset=256 * i, z_total=z_total)
            cap = (sqeazy_amd.max_compressed_length(pipeline, (256, 2048, 2048), np.uint16) + 255) & ~255
            buf = torch.empty(cap * nslabs, dtype=torch.uint8, device=dev)
            best, sizes = None, []
            for _ in range(3):                                 # (the first pass lets every context allocate its workspace)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rc, offs, sizes = sqeazy_amd.encode_slabs_device(pipeline, vol.data_ptr(), (nslabs * 256, 2048, 2048), np.uint16, nslabs,
                                                                 buf.data_ptr(), cap, inflight=SLABS_INFLIGHT)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                if rc:
                    raise RuntimeError("SQYAMD_PipelineEncode_Slabs_UI16_Device returned %d" % rc)
                best = dt if best is None or dt < best else best
            nvox = nslabs * 256 * 2048 * 2048
            algo = algo_per_voxel * nvox + sum(sizes)          # (payload + ~700 B of header per slab)
            out[key] = {"ms_total": round(best * 1e3, 2), "input_GBps": round(2 * nvox / best / 1e9, 1),
                        "roofline_frac": round(algo / best / 1e9 / HBM_PEAK_GBS, 5),
                        "entry_point": "SQYAMD_PipelineEncode_Slabs_UI16_Device, one call, %d slab calls in flight" % SLABS_INFLIGHT}
            del vol, buf
            torch.cuda.empty_cache()
        except Exception as e:   # reported, never required
            out[key] = {"error": repr(e)}

    volume_in_flight(PIPELINE, 8, 2048, 2, "north_star 2048^3 u16 bitswap1->lz4, ONE Slabs call (8 slabs, three in flight)")
    volume_in_flight("diff3x3x1->bitswap1->lz4", 8, 2048, 2, "C3 2048^3 u16 diff3x3x1->bitswap1->lz4, ONE Slabs call (8 slabs, three in flight)")
    volume_in_flight("quantiser->bitswap1->lz4", 4, 1024, 4, "C5 2048x2048x1024 u16 quantiser->bitswap1->lz4, ONE Slabs call (4 slabs, three in flight)")
    # the same north_star volume from a plain C program (tools/slabs_c_test.c, a child process: its own HIP runtime, no torch
    # streams sharing the hardware queues with the library's) -- what a C caller of the entry point gets
    try:
        exe = os.path.join(ROOT, "sqeazy_amd", "bin", "slabs_c_test")
        if os.path.exists(exe):
            torch.cuda.synchronize()
            r = subprocess.run([exe, "2048", "2048", "2048", "8", PIPELINE, str(SLABS_INFLIGHT)], capture_output=True, text=True, timeout=180)
            ms = None
            for ln in r.stdout.splitlines():
                if ln.startswith("one call,"):
                    ms = float(ln.split(":")[1].split("ms")[0])
            if r.returncode == 0 and ms:
                nvox = 2048 ** 3
                payload = None
                for ln in r.stdout.splitlines():
                    if "blob bytes" in ln:
                        payload = int(ln.split(";")[1].split("blob bytes")[0])
                if payload:
                    out["north_star 2048^3 u16 bitswap1->lz4, ONE Slabs call from a plain C program (tools/slabs_c_test.c, child process)"] = {
                        "ms_total": round(ms, 2), "input_GBps": round(2 * nvox / (ms / 1e3) / 1e9, 1),
                        "roofline_frac": round((2 * nvox + payload) / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                        "blobs_equal_single_calls": "blobs equal" in r.stdout}
    except Exception as e:   # reported, never required
        out["slabs_c_test"] = {"error": repr(e)}
    return out


class StepRunner:
    """The step loop of the timed region: `inflight` caller threads that live as long as the run (an application's encoder threads do),
    thread t takes steps t, t + inflight, ..; the main thread takes the finished steps IN STEP ORDER and, for N > 1, either exchanges the
    blob sizes (sharded container) or hands the blob to the gatherer (north_star's gather to rank 0, overlapped), then gives the buffer
    back to its thread.  Everything device- or library-specific is passed in, so that tests/test_multi_gpu_gloo.py can drive this very loop
    under two gloo processes with the CPU oracle standing in for the encoder:
      encode(t, b) -> (offset, bytes)      one step of caller thread t into its buffer b (two buffers per thread); raises on failure
      blob_view(t, b, offset)              the tensor that starts at the blob inside buffer b of thread t (what the gatherer is handed)
      exchange(nbytes, slot)               N > 1 without gather: enqueue the all_gather of this step's blob size
      gatherer                             multi.SlabGatherer or None
      thread_init()                        run once by every caller thread (torch.cuda.set_device)"""

    def __init__(self, inflight, encode, dist_on=False, blob_view=None, exchange=None, gatherer=None, thread_init=None):
        self.inflight, self.encode, self.dist_on = inflight, encode, dist_on
        self.blob_view, self.exchange, self.gatherer, self.thread_init = blob_view, exchange, gatherer, thread_init
        self.last_at = (0, 0, 0)                 # (thread, buffer, offset) of the blob of the last step taken
        self.last_blob = [None] * inflight       # per caller thread: (buffer, offset, bytes) of the last blob it produced
        self.job_q = [queue.Queue() for _ in range(inflight)]
        self.callers = [threading.Thread(target=self._caller, args=(t,), daemon=True) for t in range(inflight)]
        for th in self.callers:
            th.start()

    def _caller(self, t):
        if self.thread_init:
            self.thread_init()
        inflight = self.inflight
        while True:
            job = self.job_q[t].get()
            if job is None:
                return
            k, free_qt, done_q, stop, errors = job
            try:
                if free_qt is None:
                    # one GPU, nothing consumes the blobs between the steps: the thread's steps back to back, its two buffers taking turns
                    i = 0
                    for s_ in range(t, k, inflight):
                        off, n = self.encode(t, i & 1)
                        self.last_blob[t] = (i & 1, off, n)
                        i += 1
                    if i:
                        done_q.put((-3, t, self.last_blob[t][0], self.last_blob[t][2], self.last_blob[t][1]))
                else:
                    for s_ in range(t, k, inflight):
                        b = free_qt.get()
                        if b is None or stop.is_set():
                            break
                        off, n = self.encode(t, b)
                        self.last_blob[t] = (b, off, n)
                        done_q.put((s_, t, b, n, off))
            except Exception as e:   # pragma: no cover
                errors.append(e)
                done_q.put((-1, t, 0, 0, 0))
            done_q.put((-2, t, 0, 0, 0))          # this thread's share of the block is over

    def run_steps(self, k, gather=False):
        """k steps: thread t encodes steps t, t+inflight, ...; the main thread takes them in step order and either exchanges the
        sizes (sharded container) or hands the blob to the gatherer (overlapped gather to rank 0), then recycles the buffer"""
        inflight = self.inflight
        done_q = queue.Queue()
        handoff = self.dist_on                  # N > 1: the main thread takes every step's blob (size exchange / gather) before its buffer is reused
        free_q = [queue.Queue() for _ in range(inflight)]
        for fq in free_q:
            fq.put(0)
            fq.put(1)
        errors = []
        stop = threading.Event()
        for t in range(inflight):
            self.job_q[t].put((k, free_q[t] if handoff else None, done_q, stop, errors))
        pending, nxt, last_n, finished = {}, 0, 0, 0
        while finished < inflight:
            s_, t, b, n, off = done_q.get()
            if s_ == -2:
                finished += 1
                continue
            if s_ == -3:                        # (one GPU) a thread's block is done: remember where its last blob sits
                self.last_at = (t, b, off)
                last_n = n
                continue
            if s_ < 0:
                stop.set()
                for fq in free_q:                    # wake every caller thread that waits for a buffer, then fail loudly
                    fq.put(None)
                continue
            pending[s_] = (t, b, n, off)
            while nxt in pending:
                t2, b2, n2, off2 = pending.pop(nxt)
                self.last_at = (t2, b2, off2)
                if self.dist_on and gather:
                    # posted, not waited for: the buffer goes back to its caller thread when the gather of this step is done
                    # (the blob sits at off2 inside its buffer: frames in place)
                    self.gatherer.post(self.blob_view(t2, b2, off2), n2, on_done=lambda q=free_q[t2], bb=b2: q.put(bb))
                else:
                    if self.dist_on:
                        self.exchange(n2, nxt)      # enqueued; the closing fence waits for it
                    free_q[t2].put(b2)
                last_n = n2
                nxt += 1
        if self.dist_on and gather:
            self.gatherer.drain()
        if errors:
            raise errors[0]
        return last_n

    def close(self):
        for q in self.job_q:
            q.put(None)
        for th in self.callers:
            th.join(timeout=10)


def per_rank_rows(local_blocks, steps, world, device, device_index):
    """every rank's own median ms per step (before / after the closing fence) and what the backend says the world is, gathered to every
    rank: makes a first real N > 1 run readable -- a slow rank, a rank that waits at the fence, a world that is not the one asked for"""
    import torch
    import torch.distributed as dist
    loc = local_blocks or [(0.0, 0.0)]
    mine = torch.tensor([statistics.median(a for a, _ in loc) / steps * 1e3, statistics.median(b for _, b in loc) / steps * 1e3,
                         float(device_index)], dtype=torch.float64, device=device)
    allr = torch.zeros(world * 3, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allr, mine)
    rows = allr.reshape(world, 3).tolist()
    return {"ms_per_step_own": [round(r[0], 4) for r in rows], "ms_per_step_fenced": [round(r[1], 4) for r in rows],
            "device_index": [int(r[2]) for r in rows], "world_size_rccl": int(dist.get_world_size()), "backend": dist.get_backend(),
            "world_size_env": int(os.environ.get("WORLD_SIZE", "1"))}


def gather_stats_delta(g0, g1):
    """what the gather thread did between two snapshots of SlabGatherer.stats"""
    ng = max(g1["gathers"] - g0["gathers"], 1)
    return {"gathers_timed": g1["gathers"] - g0["gathers"], "bytes_gathered_per_step": int((g1["bytes"] - g0["bytes"]) / ng),
            "gather_ms_per_step": round((g1["seconds"] - g0["seconds"]) / ng * 1e3, 4)}


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks as a torchrun child (nothing in this process has touched the GPU yet)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    global SLABS_INFLIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="main measurement only (no CPU baseline, host ABI, secondary configs): profiling runs")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="keep timing blocks of --steps steps until this much has been timed")
    ap.add_argument("--frames", type=int, default=SHAPE[0], help="z extent per GPU (default: the BASELINE config)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the second measurement with the RCCL gather to rank 0 inside the step")
    ap.add_argument("--slabs-inflight", type=int, default=SLABS_INFLIGHT)
    ap.add_argument("--inflight", type=int, default=4,
                    help="C-ABI calls in flight per GPU (host threads, one stream + workspace each; the C-ABI is re-entrant like "
                         "the reference's).  1 = strictly one call after the other.  A call is a chain of dependent kernels -- transpose "
                         "(HBM), duplicate search, LZ4 parse (latency-bound, HBM mostly idle), frame gather -- so several calls keep "
                         "every unit busy (measured in round 3, GPU_MAX_HW_QUEUES=8: 3 -> 1518, 4 -> 1665, 5 -> 1350, 6 -> 1550 GB/s; "
                         "with the runtime's default of four hardware queues: 3 -> 1515, 4 -> 1250)")
    args = ap.parse_args()
    SLABS_INFLIGHT = max(1, args.slabs_inflight)

    # (SQY_BENCH_FORCE_SPAWN=1: rehearsal of the self-launch on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("SQY_BENCH_FORCE_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import sqeazy_amd
    from sqeazy_amd import multi, synth

    sqeazy_amd.lib()   # fails loudly when the HIP library has not been built
    # The caller threads' streams below carry nothing but these calls: the library may chain the bit-plane transposes of the calls in
    # flight across them (a hipStreamWaitEvent between caller streams; off unless the caller says so, include/sqeazy_amd.h).  +3 %.
    # (round-5 advice: scoped -- `with sqeazy_amd.option(...)` around the legs with calls in flight only, the line says what `value` was
    # measured under, and the same leg with the option at its default sits next to it as `default_options`)
    CHAIN = ("transpose_chain_caller_streams", 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path (RCCL init, size exchange, gather, barrier, max-reduce) on a one-GPU box: a group of one
    dist_on = world > 1 or os.environ.get("SQY_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()          # the ranks RCCL actually sees

    shape = (args.frames, SHAPE[1], SHAPE[2])
    # rank r holds frames [r*Z, (r+1)*Z) of an (N*Z, Y, X) synthetic stack
    vol = synth.stack_torch(shape, np.uint16, dev, z_offset=rank * shape[0], z_total=world * shape[0])
    nbytes = vol.numel() * 2
    # one copy of the input per call in flight (4 GiB of HBM at the default): no two calls of the timed region read the same addresses,
    # so nothing the caches keep of one call's input can serve another's
    vols = [vol] + [vol.clone() for _ in range(max(1, args.inflight) - 1)]
    cap = sqeazy_amd.max_compressed_length(PIPELINE, shape, np.uint16)
    index_rows = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(8)] if dist_on else []   # container index of the last steps
    sys.setswitchinterval(1e-4)      # caller threads hand the GIL over promptly (default 5 ms would show up as whole milliseconds per step)
    inflight = max(1, args.inflight)
    # every caller thread owns a stream and two output buffers (the gather of step s may still read one while s+inflight encodes)
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
    outs = [[torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(inflight)]
    gatherer = multi.SlabGatherer(world * cap, dev) if (dist_on and not args.no_gather) else None
    torch.cuda.synchronize()

    # the C call of every (thread, buffer), marshalled once: what a C caller's loop looks like (no Python object is built per step)
    entry = sqeazy_amd.lib().SQYAMD_PipelineEncode_UI16_DeviceAt
    pipe_b = PIPELINE.encode()
    shape_c = (ctypes.c_long * 3)(*shape)

    def prepared(t, b):
        doff, dlen = ctypes.c_long(0), ctypes.c_long(0)
        args = (pipe_b, ctypes.c_void_p(vols[t].data_ptr()), shape_c, ctypes.c_uint(3), ctypes.c_void_p(outs[t][b].data_ptr()), ctypes.c_long(cap),
                ctypes.byref(doff), ctypes.byref(dlen), ctypes.c_int(0), ctypes.c_void_p(streams[t].cuda_stream))
        return args, doff, dlen

    calls = [[prepared(t, b) for b in range(2)] for t in range(inflight)]

    def encode_step(t, b):
        args, doff, dlen = calls[t][b]
        rc = entry(*args)
        if rc:
            raise RuntimeError("SQYAMD_PipelineEncode_UI16_DeviceAt returned %d" % rc)
        return doff.value, dlen.value

    # the caller threads live as long as the run; a block of steps is handed to them as a job (StepRunner above: the same loop the
    # world-size-2 gloo test drives with the CPU oracle as the encoder)
    runner = StepRunner(inflight, encode_step, dist_on=dist_on, blob_view=lambda t, b, off: outs[t][b][off:],
                        exchange=(lambda n, slot: multi.exchange_sizes(n, dev, out=index_rows[slot % len(index_rows)], sync=False)) if dist_on else None,
                        gatherer=gatherer, thread_init=lambda: torch.cuda.set_device(local_rank))
    run_steps = runner.run_steps
    last_blob = runner.last_blob

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_blocks(gather):
        """blocks of exactly --steps steps, each fenced on both sides, max over ranks per block, until --min-seconds are timed"""
        times, payload = [], 0
        local = []                                          # this rank's own clock per block (N > 1 diagnostics)
        while sum(times) < args.min_seconds and len(times) < 200:
            fence()
            t0 = time.perf_counter()
            payload = run_steps(args.steps, gather)
            t_own = time.perf_counter() - t0                # up to the last step's return on THIS rank, before the closing fence
            fence()
            dt = time.perf_counter() - t0
            local.append((t_own, dt))
            if dist_on:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            times.append(dt)
        local_blocks[bool(gather)] = local
        return times, payload

    local_blocks = {}

    def per_rank_report(gather):
        if not dist_on:
            return None
        return per_rank_rows(local_blocks.get(bool(gather)), args.steps, world, dev, torch.cuda.current_device())

    run_steps(inflight)          # untimed priming: every caller thread's context allocates its HBM workspace once
    if args.warmup:
        run_steps(args.warmup)
    sqeazy_amd.profile_reset()
    sqeazy_amd.profile_enable(True)
    with sqeazy_amd.option(*CHAIN):
        times, payload = timed_blocks(False)
        options_timed = {k: sqeazy_amd.get_option(k) for k in ("transpose_chain", "transpose_chain_caller_streams", "block_parallel")}
    sqeazy_amd.profile_enable(False)
    prof = sqeazy_amd.profile_get()
    nblocks = len(times)
    # the same leg as an unchanged caller gets it: every option at its default (the transposes of caller streams are not chained)
    default_leg = None
    if world == 1 and not args.quick:
        fence()
        t0 = time.perf_counter()
        run_steps(args.steps)
        fence()
        dt0 = time.perf_counter() - t0
        default_leg = {"value": round(nbytes * args.steps / dt0 / 1e9, 1), "unit": "GB/s", "ms_per_step": round(dt0 / args.steps * 1e3, 4),
                       "options": {k: sqeazy_amd.get_option(k) for k in ("transpose_chain", "transpose_chain_caller_streams")},
                       "timing": "one block of %d steps" % args.steps}
    # What was timed is also checked (after the clock stopped): the LAST blob of every caller thread is hashed and compared with the
    # digest the reference pieces themselves give for this stack (tests/golden/headline.json: reference SSE bit-plane gather + liblz4
    # 1.9.3 frames, oracle/gen_golden.py --headline; a data file, nothing of oracle/ runs here)
    digests = []
    timed_blob_bytes, timed_header_bytes = None, None      # of a blob of the timed region itself, read before any other leg reuses a buffer
    for t in range(inflight):
        if last_blob[t] is not None:
            b, off, n = last_blob[t]
            raw = outs[t][b][off:off + n].cpu().numpy().tobytes()
            digests.append(hashlib.sha256(raw).hexdigest())
            if timed_blob_bytes is None:
                timed_blob_bytes, timed_header_bytes = n, sqeazy_amd.header_size(raw[:65536])
            del raw
    verify = {"verified": False, "blobs_hashed": len(digests), "blob_sha256": digests[0] if digests else None,
              "threads_agree": len(set(digests)) == 1}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "headline.json")) as f:
            for g in json.load(f)["stacks"]:
                if tuple(g["shape_zyx"]) == tuple(shape) and g["z_offset"] == rank * shape[0] and g["z_total"] == world * shape[0]:
                    verify["reference_blob_sha256"] = g["blob_sha256"]
                    verify["payload_sha256"] = g["payload_sha256"]
                    verify["against"] = "tests/golden/headline.json (%s): %s" % (g["name"], g["source"])
                    verify["verified"] = bool(digests) and all(d == g["blob_sha256"] for d in digests)
    except Exception as e:   # reported, never fatal for the measurement
        verify["error"] = repr(e)
    if "against" not in verify:
        verify["against"] = "no reference digest for this shape (only the BASELINE stack has one); threads compared with each other"

    per_rank = per_rank_report(False)
    gather_times, gather_stats, per_rank_gather = None, None, None
    if gatherer is not None:
        with sqeazy_amd.option(*CHAIN):
            run_steps(max(2, inflight), True)
            g0 = dict(gatherer.stats)
            gather_times, _ = timed_blocks(True)
            gather_stats = gather_stats_delta(g0, dict(gatherer.stats))
        per_rank_gather = per_rank_report(True)

    # other operating points of the same step, for comparison with earlier rounds (round-3 advice): fewer calls in flight, and the entry
    # point that leaves the blob at the start of the destination (no frames in place: the payload is gathered)
    alt = {}
    if world == 1 and not args.quick:
        def alt_point(n_inflight, fn_name):
            fn = getattr(sqeazy_amd.lib(), fn_name)
            errs = []

            def th(t):
                try:
                    torch.cuda.set_device(local_rank)
                    dlen, doff = ctypes.c_long(0), ctypes.c_long(0)
                    for _ in range(t, args.steps, n_inflight):
                        if fn_name.endswith("DeviceAt"):
                            rc = fn(pipe_b, ctypes.c_void_p(vols[t].data_ptr()), shape_c, ctypes.c_uint(3), ctypes.c_void_p(outs[t][0].data_ptr()), ctypes.c_long(cap),
                                    ctypes.byref(doff), ctypes.byref(dlen), ctypes.c_int(0), ctypes.c_void_p(streams[t].cuda_stream))
                        else:
                            rc = fn(pipe_b, ctypes.c_void_p(vols[t].data_ptr()), shape_c, ctypes.c_uint(3), ctypes.c_void_p(outs[t][0].data_ptr()), ctypes.c_long(cap),
                                    ctypes.byref(dlen), ctypes.c_int(0), ctypes.c_void_p(streams[t].cuda_stream))
                        if rc:
                            raise RuntimeError("%s returned %d" % (fn_name, rc))
                except Exception as e:   # pragma: no cover
                    errs.append(e)
            best = None
            for _ in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ths = [threading.Thread(target=th, args=(t,)) for t in range(n_inflight)]
                [x.start() for x in ths]; [x.join() for x in ths]
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                if errs:
                    raise errs[0]
                best = dt if best is None or dt < best else best
            return {"value": round(nbytes * args.steps / best / 1e9, 1), "unit": "GB/s", "ms_per_step": round(best / args.steps * 1e3, 4),
                    "calls_in_flight": n_inflight, "entry_point": fn_name, "timing": "best of 6 blocks of %d steps" % args.steps}
        try:
            with sqeazy_amd.option(*CHAIN):
                alt["3_in_flight_DeviceAt"] = alt_point(min(3, inflight), "SQYAMD_PipelineEncode_UI16_DeviceAt")
                alt["4_in_flight_Device_blob_at_offset_0"] = alt_point(min(4, inflight), "SQYAMD_PipelineEncode_UI16_Device")
            alt["note"] = ("round 3's headline was 4 in flight / DeviceAt / GPU_MAX_HW_QUEUES=8 as well; round 2's was 3 in flight / _Device; the hardware-queue "
                           "count is read once by the HIP runtime and cannot be varied inside one process")
        except Exception as e:   # reported, never required
            alt["error"] = repr(e)
    # one call at a time (nothing else in flight), for the record: latency of the call and the kernels' undisturbed durations
    fence()
    sqeazy_amd.profile_reset()
    sqeazy_amd.profile_enable(True)
    single = []
    for _ in range(5):
        tl = time.perf_counter()
        rc, single_off, _n = sqeazy_amd.encode_device_at(PIPELINE, vol.data_ptr(), shape, np.uint16, outs[0][0].data_ptr(), cap, nthreads=0,
                                                         stream=streams[0].cuda_stream)
        torch.cuda.synchronize()
        single.append((time.perf_counter() - tl) * 1e3)
    sqeazy_amd.profile_enable(False)
    prof_alone = sqeazy_amd.profile_get()
    fence()
    # the layout every default caller of the reference asks for (nthreads = 1: one block-linked frame), one call at a time; its blob is
    # hashed against the digest the reference pieces give for that layout (tests/golden/headline.json "serial")
    serial_layout = None
    if rank == 0:
        try:
            sqeazy_amd.profile_reset()
            sqeazy_amd.profile_enable(True)
            sl = []
            for _ in range(4):
                tl = time.perf_counter()
                rc, sn = sqeazy_amd.encode_device(PIPELINE, vol.data_ptr(), shape, np.uint16, outs[0][0].data_ptr(), cap, nthreads=1,
                                                  stream=streams[0].cuda_stream)
                torch.cuda.synchronize()
                sl.append((time.perf_counter() - tl) * 1e3)
                if rc != 0:
                    raise RuntimeError("nthreads = 1 call failed")
            sqeazy_amd.profile_enable(False)
            sdig = hashlib.sha256(outs[0][0][:sn].cpu().numpy().tobytes()).hexdigest()
            serial_layout = {"ms": round(min(sl[1:]), 4), "value": round(nbytes / (min(sl[1:]) / 1e3) / 1e9, 1), "unit": "GB/s", "blob_bytes": int(sn),
                             "blob_sha256": sdig, "verified": None,
                             "what": "SQYAMD_PipelineEncode_UI16_Device, nthreads = 1 (ONE block-linked LZ4 frame, lz4_utils.hpp:99-173), one call at a time; "
                                     "blocks parsed block-parallel from verified table guesses (DESIGN.md section 3)",
                             "kernels_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in sqeazy_amd.profile_get().items()}}
            # .. and back: the decode of that one frame (every block at once with the history as an unknown, DESIGN.md section 5 "decode")
            back = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            dfn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device
            dl = []
            for _ in range(3):
                tl = time.perf_counter()
                drc = dfn(ctypes.c_void_p(outs[0][0].data_ptr()), ctypes.c_long(sn), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nbytes), None)
                torch.cuda.synchronize()
                dl.append((time.perf_counter() - tl) * 1e3)
            serial_layout["decode"] = {"ms": round(min(dl[1:]), 4), "value": round(nbytes / (min(dl[1:]) / 1e3) / 1e9, 1), "unit": "GB/s", "rc": int(drc),
                                       "round_trip_equal": bool((back.view(torch.uint16).reshape(shape) == vol).all().item()),
                                       "what": "SQYAMD_Decode_UI16_Device on that blob"}
            del back
            with open(os.path.join(ROOT, "tests", "golden", "headline.json")) as f:
                for g in json.load(f)["stacks"]:
                    if tuple(g["shape_zyx"]) == tuple(shape) and g["z_offset"] == 0 and g["z_total"] == world * shape[0] and "serial" in g:
                        serial_layout["verified"] = sdig == g["serial"]["blob_sha256"]
                        serial_layout["against"] = "tests/golden/headline.json (%s, serial): reference SSE bitswap + liblz4 1.9.3 encode_serial" % g["name"]
        except Exception as e:   # reported, never required
            serial_layout = {"error": repr(e)}
    fence()
    runner.close()                       # the caller threads are done

    if rank == 0:
        dt = statistics.median(times)
        if timed_blob_bytes is None:
            raise SystemExit("bench.py: the timed region produced no blob")
        payload = timed_blob_bytes
        payload_bytes = timed_blob_bytes - timed_header_bytes     # LZ4 frames only: what the path's algorithmic bytes count as written
        # dominant kernel by device time; per-launch average over every launch of the timed blocks
        dom, (dom_ms, dom_n) = max(prof.items(), key=lambda kv: kv[1][0]) if prof else ("none", (0.0, 0))
        avg_ms = dom_ms / max(dom_n, 1)
        algo_bytes = nbytes + payload_bytes                  # 2 B/voxel read once + payload written once
        achieved = (algo_bytes / 1e9) / (avg_ms / 1e3) if avg_ms else 0.0
        ident = lib_identity()
        alone_ms = prof_alone[dom][0] / max(prof_alone[dom][1], 1) if dom in prof_alone else None
        single_ms = min(single)
        mode = "%d calls in flight" % inflight
        line = {
            "metric": "encode GB/s (input voxels) for bitswap1->lz4 uint16 volume",
            "value": round(world * nbytes * args.steps / dt / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "timing": {"blocks": nblocks, "steps_per_block": args.steps, "seconds_timed": round(sum(times), 3), "reported": "median block",
                       "ms_per_step_min": round(min(times) / args.steps * 1e3, 4), "ms_per_step_max": round(max(times) / args.steps * 1e3, 4)},
            "config": {"workload": "%dx%dx%d uint16 synthetic microscopy stack per GPU, pipeline '%s', one C-ABI call per step, %s%s" % (
                shape[2], shape[1], shape[0], PIPELINE, mode,
                ", slab blobs stay sharded on their GPUs, sizes all_gathered over RCCL" if world > 1 else ""),
                "input_bytes_per_gpu": nbytes, "payload_bytes": payload_bytes, "blob_bytes": payload, "calls_in_flight_per_gpu": inflight,
                "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
                "options": options_timed, "options_note": "transpose_chain_caller_streams is OFF by default (include/sqeazy_amd.h): `value` is measured with the "
                                                         "caller opting in; `default_options` is the same leg without",
                "default_options": default_leg,
                "single_call_latency_ms": round(single_ms, 4),
                "one_call_at_a_time": {"value": round(nbytes / (single_ms / 1e3) / 1e9, 1), "unit": "GB/s",
                                       "roofline_frac_whole_call": round(algo_bytes / (single_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         # every kernel of the call together: algorithmic bytes of one step / wall time of one step
                         "whole_step_frac": round((algo_bytes / 1e9) / (dt / args.steps) / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(avg_ms, 4), "launches_timed": dom_n,
                         # the same kernel with the GPU to itself (one call at a time, measured right after the timed region)
                         "alone_launch_ms": round(alone_ms, 4) if alone_ms else None,
                         "alone_frac": round((algo_bytes / 1e9) / (alone_ms / 1e3) / HBM_PEAK_GBS, 5) if alone_ms else None,
                         "kernels_ms_per_step": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}},
            # one call at a time (what the sqy tool, the HDF5 filter and the Java binding do): latency of the call, its rate, and
            # the whole call's algorithmic bytes against the roofline
            "single_call": {"ms": round(single_ms, 4), "value": round(nbytes / (single_ms / 1e3) / 1e9, 1), "unit": "GB/s",
                            "roofline_frac": round(algo_bytes / (single_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                            "kernels_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof_alone.items()}},
            "verified": verify["verified"], "payload_sha256": verify.get("payload_sha256"), "verification": verify,
            "serial_layout": serial_layout,
            "other_operating_points": alt,
            "entry_point": "SQYAMD_PipelineEncode_UI16_DeviceAt (device pointers; the blob may start anywhere in the destination: frames in place)",
            "build": ident,
        }
        # launches of every kernel per encode call inside the timed region (1 each on this stack; a dense second pass would count too)
        ncalls = max(nblocks * args.steps, 1)
        tr = measured_traffic(ident["sha256"], dom, {k: v[1] / ncalls for k, v in prof.items()})
        if tr:
            line["roofline"]["traffic"] = tr["bytes_per_launch"]
            line["roofline"]["traffic_all_kernels_per_call"] = tr["per_call_all_kernels"]     # the kernels of ONE timed encode call
            line["roofline"]["traffic_ratio"] = round(tr["per_call_all_kernels"] / algo_bytes, 4)   # HBM bytes moved / algorithmic bytes
            if tr["kernels_missing"]:
                line["roofline"]["traffic_kernels_missing"] = tr["kernels_missing"]
            line["roofline"]["traffic_source"] = tr["source"]
        else:
            line["roofline"]["traffic_source"] = "no rocprofv3 PMC pass of this build under profiles/ (tools/profile_run.sh)"
        if gather_times:
            gdt = statistics.median(gather_times)
            line["with_gather"] = {"value": round(world * nbytes * args.steps / gdt / 1e9, 3), "unit": "GB/s",
                                   "ms_per_step": round(gdt / args.steps * 1e3, 4), "blocks": len(gather_times),
                                   "what": "the same steps with the compressed slab of every rank gathered to rank 0 over RCCL (sizes all_gather + "
                                           "ncclSend/ncclRecv) inside the step, posted from a gather thread and overlapped with the next encodes"}
            # the gather by itself (rank 0's gather thread: size exchange + transfers + the wait for them) and every rank's own clock
            line["with_gather"].update(gather_stats or {})
            line["with_gather"]["per_rank"] = per_rank_gather
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["world_size_rccl"] = per_rank["world_size_rccl"]
        if world == 1 and not args.quick:
            vol_host = vol.cpu().numpy()
            if not args.no_cpu_baseline:
                try:
                    line["cpu_baseline"] = cpu_baseline(vol_host)
                except Exception as e:   # the baseline is reported, never required
                    line["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
            try:
                line["host_abi"] = host_abi(vol_host, shape)
            except Exception as e:
                line["host_abi"] = {"value": None, "error": repr(e)}
            del vol
            vols.clear()
            outs.clear()
            torch.cuda.empty_cache()
            sqeazy_amd.lib().SQYAMD_Release_Workspace()
            try:
                line["config"]["secondary"] = secondary_configs(dev)
            except Exception as e:
                line["config"]["secondary"] = {"error": repr(e)}
            # north_star's own target (2048^3 uint16, bitswap1->lz4, one GPU) as a top-level key and as scalars of `config`, so that it
            # survives a reader that keeps only the scalar fields of the line
            sec = line["config"]["secondary"]
            ns = {}
            for key, short in (("ONE Slabs call from a plain C program", "one_slabs_call_plain_c"), ("ONE Slabs call (8 slabs", "one_slabs_call"),
                               ("8 sequential", "eight_sequential_calls")):
                for k, v in sec.items():
                    if k.startswith("north_star") and key in k and isinstance(v, dict) and "ms_total" in v:
                        ns[short] = {"ms_total": v["ms_total"], "input_GBps": v["input_GBps"], "roofline_frac": v["roofline_frac"],
                                     "entry_point": v.get("entry_point", "SQYAMD_PipelineEncode_UI16_DeviceAt, one call after the other"
                                                          if short == "eight_sequential_calls" else "SQYAMD_PipelineEncode_Slabs_UI16_Device (tools/slabs_c_test.c)")}
            if ns:
                best_key = max(ns, key=lambda k: ns[k]["roofline_frac"])
                line["north_star"] = {"workload": "2048x2048x2048 uint16, bitswap1->lz4, 8 z-slabs of 2048x2048x256 on ONE MI355X, inputs resident in HBM",
                                      "target_roofline_frac": 0.5, "best": best_key, **ns}
                line["config"]["north_star_2048cube_ms_total"] = ns[best_key]["ms_total"]
                line["config"]["north_star_2048cube_roofline_frac"] = ns[best_key]["roofline_frac"]
                line["config"]["north_star_2048cube_entry_point"] = best_key
            if isinstance(line.get("host_abi"), dict) and "error" not in line["host_abi"]:
                line["host_abi"]["two_calls_in_flight"] = host_abi_two_calls(vol_host, shape)      # (last: see there)
            del vol_host
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
