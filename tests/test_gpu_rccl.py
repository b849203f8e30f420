"""Section C of the C-ABI on one GPU: the RCCL gather of slab blobs (a communicator of ONE rank: the code path of the N-GPU run --
dlopen of RCCL, unique id, communicator, size all-gather, grouped send / receive, the root's own copy -- rehearsed where only one
GPU is at hand), and the frame-offset table that replaces walking a blob's LZ4 frames on the host."""
import ctypes

import numpy as np
import pytest

from sqeazy_amd import synth, multi

pytestmark = pytest.mark.gpu


def test_gather_blobs_world_of_one(sqy):
    import torch
    dev = torch.device("cuda", 0)
    L = sqy.lib()
    ident = ctypes.create_string_buffer(128)
    assert L.SQYAMD_Comm_UniqueId(ident) == 0
    comm = ctypes.c_void_p()
    assert L.SQYAMD_Comm_Init(ctypes.byref(comm), 1, 0, ident) == 0
    try:
        vol = synth.stack_torch((32, 128, 128), np.uint16, dev)
        cap = sqy.max_compressed_length("bitswap1->lz4", (32, 128, 128), np.uint16)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        rc, off, n = sqy.encode_device_at("bitswap1->lz4", vol.data_ptr(), (32, 128, 128), np.uint16, out.data_ptr(), cap)
        assert rc == 0
        recv = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        sizes = (ctypes.c_long * 1)()
        stream = torch.cuda.Stream(device=dev)
        rc = L.SQYAMD_Gather_Blobs(comm, 0, ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(n), ctypes.c_void_p(recv.data_ptr()),
                                   ctypes.c_long(recv.numel()), sizes, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0 and sizes[0] == n
        assert torch.equal(recv[:n], out[off:off + n]) and int(recv[n:].sum().item()) == 0
        # a root buffer that is too small: refused (by every rank, before anybody sends)
        rc = L.SQYAMD_Gather_Blobs(comm, 0, ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(n), ctypes.c_void_p(recv.data_ptr()),
                                   ctypes.c_long(n - 1), sizes, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 1
        # an empty blob is a legal member of the container
        rc = L.SQYAMD_Gather_Blobs(comm, 0, None, ctypes.c_long(0), ctypes.c_void_p(recv.data_ptr()), ctypes.c_long(recv.numel()), sizes,
                                   ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0 and sizes[0] == 0
    finally:
        assert L.SQYAMD_Comm_Destroy(comm) == 0


@pytest.mark.parametrize("dtype,shape", [(np.uint16, (64, 256, 256)), (np.uint8, (64, 256, 512))])
def test_frame_offsets_are_the_plane_ranges(sqy, dtype, shape):
    """SQYAMD_PipelineEncode_*_DeviceAt_Frames with every = chunks per bit plane against the host walk over the blob's frames"""
    import torch
    dev = torch.device("cuda", 0)
    vol = synth.stack_torch(shape, dtype, dev)
    W = np.dtype(dtype).itemsize * 8
    plane_bytes = int(np.prod(shape)) // 8
    every = plane_bytes // multi.LZ4_CHUNK_BYTES
    assert every >= 1 and plane_bytes % multi.LZ4_CHUNK_BYTES == 0
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, dtype)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, off, n, fo = sqy.encode_device_at_frames("bitswap1->lz4", vol.data_ptr(), shape, dtype, out.data_ptr(), cap, every)
    assert rc == 0 and len(fo) == W + 1 and fo[-1] == n
    blob = bytes(out[off:off + n].cpu().numpy().tobytes())
    hdr, ranges = multi.plane_ranges(blob, shape, dtype)
    assert multi.plane_ranges_from_frame_offsets(fo, W) == (hdr, ranges)


def test_single_blob_from_slabs_without_a_host_walk(sqy):
    """two slab blobs + their frame-offset tables -> the blob of the whole volume (multi.assemble_single_blob), byte for byte what
    one call on the whole volume yields"""
    import torch
    dev = torch.device("cuda", 0)
    shape, dtype, world = (64, 256, 256), np.uint16, 2
    vol = synth.stack_torch(shape, dtype, dev)
    cap = sqy.max_compressed_length("bitswap1->lz4", shape, dtype)
    whole = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, woff, wn = sqy.encode_device_at("bitswap1->lz4", vol.data_ptr(), shape, dtype, whole.data_ptr(), cap)
    assert rc == 0
    assert multi.single_blob_possible(shape, dtype, world)
    blobs, ranges = [], []
    for r in range(world):
        z0, nz = multi.slab_range(shape[0], r, world)
        sshape = (nz, shape[1], shape[2])
        every = (nz * shape[1] * shape[2] // 8) // multi.LZ4_CHUNK_BYTES
        buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        rc, off, n, fo = sqy.encode_device_at_frames("bitswap1->lz4", vol[z0:z0 + nz].data_ptr(), sshape, dtype, buf.data_ptr(), cap, every)
        assert rc == 0
        blobs.append(buf[off:off + n])
        ranges.append(multi.plane_ranges_from_frame_offsets(fo, 16)[1])
    one = multi.assemble_single_blob(shape, dtype, blobs, ranges)
    assert one.numel() == wn and torch.equal(one, whole[woff:woff + wn])
    # volumes of 2^31 voxels and more cannot be one blob (no single call could have produced or can decode it)
    assert not multi.single_blob_possible((2048, 2048, 2048), np.uint16, 8)


def test_bench_self_spawn_and_distributed_path_on_one_gpu():
    """bench.py's N > 1 plumbing rehearsed on the one-GPU box as a fresh child process: the script starts
    its rank itself through torch.distributed.run (SQY_BENCH_FORCE_SPAWN), initialises RCCL with a group of one (SQY_BENCH_FORCE_DIST),
    exchanges the blob sizes per step, runs the second measurement with the overlapped gather to rank 0, and prints ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update({"SQY_BENCH_FORCE_SPAWN": "1", "SQY_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"})
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--quick", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["unit"] == "GB/s" and line["value"] > 0
    assert "with_gather" in line and line["with_gather"]["value"] > 0          # the RCCL gather path ran
    assert line["verified"] is True and line["verification"]["threads_agree"]  # and what was timed equals the reference digest
    assert "sharded" in line["config"]["workload"] or line["n_gpus"] == 1
    # what makes a first real N > 1 run readable: every rank's own clock, the world RCCL reports, what the
    # gather moved per step and what it cost by itself
    pr = line["per_rank"]
    assert line["world_size_rccl"] == 1 and pr["world_size_rccl"] == 1 and pr["backend"] == "nccl"
    assert len(pr["ms_per_step_own"]) == 1 and pr["ms_per_step_own"][0] > 0 and pr["ms_per_step_fenced"][0] >= pr["ms_per_step_own"][0]
    wg = line["with_gather"]
    assert wg["gathers_timed"] >= 3 and wg["gather_ms_per_step"] > 0
    assert wg["bytes_gathered_per_step"] == line["config"]["blob_bytes"]           # a world of one: its own blob per step
    assert len(wg["per_rank"]["ms_per_step_own"]) == 1
    assert line["config"]["payload_bytes"] < line["config"]["blob_bytes"]          # (header excluded, from the timed call's own blob)
