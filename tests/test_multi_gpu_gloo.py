"""CPU, 2 processes over gloo: the z-slab sharding and the variable-length gather used by bench.py at N > 1
(sqeazy_amd/multi.py runs unchanged on RCCL with device tensors)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import sqy_oracle as o
    from sqeazy_amd import multi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Z, Y, X = 13, 24, 32
        full = synth.stack((Z, Y, X))
        z0, nz = multi.slab_range(Z, rank, world)
        slab = full[z0:z0 + nz]
        # every rank "encodes" its slab (the CPU oracle stands in for the GPU call here) ...
        blob = o.pipeline_encode("bitswap1->lz4", slab)
        t = torch.zeros(len(blob) + 100, dtype=torch.uint8)
        t[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        # ... and the compressed slabs are gathered to rank 0
        # the step bench.py times at N > 1: only the sizes (the sharded container's index) are exchanged
        idx = multi.exchange_sizes(len(blob), t.device)
        assert len(idx) == world and idx[rank] == len(blob)
        for _ in range(2):                                   # twice: the buffers are reused across steps in bench.py
            sizes, flat = multi.gather_blobs(t, len(blob))
        assert sizes == idx
        if rank == 0:
            blobs = multi.unpack_container(multi.pack_container(sizes, flat))
            assert len(blobs) == world
            parts = [o.pipeline_decode(b) for b in blobs]
            assert np.array_equal(np.concatenate(parts, axis=0), full)
            # each contained blob is exactly what one C-ABI call on that slab yields
            for r, b in enumerate(blobs):
                a0, an = multi.slab_range(Z, r, world)
                assert b == o.pipeline_encode("bitswap1->lz4", full[a0:a0 + an])
        else:
            assert flat is None
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_slab_range_partitions():
    from sqeazy_amd import multi
    for Z in (1, 7, 8, 13, 512):
        for world in (1, 2, 3, 8):
            got = [multi.slab_range(Z, r, world) for r in range(world)]
            assert got[0][0] == 0 and sum(n for _, n in got) == Z
            for (a, n), (b, _) in zip(got, got[1:]):
                assert a + n == b


def test_variable_length_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
