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


def _worker_gatherer(rank, world, port, q):
    """the overlapped gather bench.py uses at N > 1 (multi.SlabGatherer), several steps in flight, uneven z split, and the
    pipeline whose slab geometry is part of the contract (diff3x3x1: every slab is encoded as its own volume)"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import sqy_oracle as o
    from sqeazy_amd import multi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Z, Y, X = 17, 12, 16
        pipe = "diff3x3x1->bitswap1->lz4"
        steps = 4
        fulls = [synth.stack((Z, Y, X), seed=100 + s) for s in range(steps)]
        z0, nz = multi.slab_range(Z, rank, world)
        cap = 1 << 16
        g = multi.SlabGatherer(world * cap, torch.device("cpu"))
        released = []
        seen = []
        bufs = []
        for s in range(steps):
            blob = o.pipeline_encode(pipe, fulls[s][z0:z0 + nz])
            t = torch.zeros(cap, dtype=torch.uint8)
            t[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            bufs.append(t)
            g.post(t, len(blob), on_done=lambda s=s: released.append(s))
            if s % 2 == 1:                                   # look at the newest container every other step, like a consumer would
                g.drain()
                if rank == 0:
                    sizes, flat = g.last
                    seen.append((s, multi.unpack_container(multi.pack_container(sizes, flat))))
        g.drain()
        g.close()
        assert released == list(range(steps)) and g.done == steps
        if rank == 0:
            for s, blobs in seen:
                assert len(blobs) == world
                for r, b in enumerate(blobs):
                    a0, an = multi.slab_range(Z, r, world)
                    assert b == o.pipeline_encode(pipe, fulls[s][a0:a0 + an]), (s, r)      # = one reference call on that slab
                    assert np.array_equal(o.pipeline_decode(b), fulls[s][a0:a0 + an])
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


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


def _worker_single_blob(rank, world, port, q):
    """single-blob mode: every rank encodes its slab (the CPU oracle stands in for the GPU call), rank 0 ends up with ONE blob
    that is byte for byte what one call on the whole volume yields"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import sqy_oracle as o
    from sqeazy_amd import multi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for dtype, shape in ((np.uint16, (8 * world, 256, 1024)), (np.uint8, (4 * world, 512, 1024))):
            full = synth.stack(shape, dtype)
            assert multi.single_blob_possible(shape, dtype, world)
            z0, nz = multi.slab_range(shape[0], rank, world)
            blob = o.pipeline_encode("bitswap1->lz4", full[z0:z0 + nz])
            t = torch.zeros(len(blob) + 64, dtype=torch.uint8)
            t[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            one = multi.gather_single_blob(t, len(blob), shape, dtype)
            if rank == 0:
                got = one.numpy().tobytes()
                assert got == o.pipeline_encode("bitswap1->lz4", full), (np.dtype(dtype).name, len(got))
                assert np.array_equal(o.pipeline_decode(got), full)
            else:
                assert one is None
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_single_blob_assembly_from_slab_blobs():
    """no process group: the re-ordering alone, 1 / 2 / 4 slabs, both voxel types; and the shapes it has to refuse"""
    sys.path.insert(0, ROOT)
    from oracle import sqy_oracle as o
    from sqeazy_amd import multi, synth
    for dtype in (np.uint16, np.uint8):
        shape = (16, 512, 1024)
        vol = synth.stack(shape, dtype)
        want = o.pipeline_encode("bitswap1->lz4", vol)
        for world in (1, 2, 4):
            blobs, ranges = [], []
            for r in range(world):
                z0, nz = multi.slab_range(shape[0], r, world)
                b = o.pipeline_encode("bitswap1->lz4", vol[z0:z0 + nz])
                blobs.append(b)
                ranges.append(multi.plane_ranges(b, (nz,) + shape[1:], dtype)[1])
            assert multi.assemble_single_blob(shape, dtype, blobs, ranges) == want, (np.dtype(dtype).name, world)
    assert not multi.single_blob_possible((13, 24, 32), np.uint16, 2)          # slabs that are no whole chunks per plane
    assert not multi.single_blob_possible((16, 512, 1024), np.uint16, 3)       # uneven split: 6 + 5 + 5 frames
    assert not multi.single_blob_possible((16, 512, 1024), np.uint16, 8)       # 2 frames = half a chunk per plane
    with pytest.raises(ValueError):
        multi.walk_frames(b"\x04\x22\x4d\x18\x40\x60\x51" + (100).to_bytes(4, "little") + b"short")


@pytest.mark.parametrize("world", [2, 4])
def test_single_blob_gather(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_single_blob, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


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


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_gather_diff_pipeline(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gatherer, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _worker_gatherer_failure(rank, world, port, q):
    """one rank's encode fails between two steps: its next post still takes part in the size exchange (poison value), so that
    EVERY rank raises in that gather instead of waiting for sends that never come"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sqeazy_amd import multi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = multi.SlabGatherer(world * 64, torch.device("cpu"))
        t = torch.arange(64, dtype=torch.uint8)
        g.post(t, 10 + rank)
        g.drain()
        if rank == 0:
            assert g.last[0] == [10 + r for r in range(world)]
        raised = []
        if rank == world - 1:
            g.error = RuntimeError("encode of step 1 failed on rank %d" % rank)        # what a failed C-ABI call leaves behind
        try:
            g.post(t, 20)
        except RuntimeError as e:
            raised.append("post: %s" % e)
        try:
            g.drain()
        except RuntimeError as e:
            raised.append("drain: %s" % e)
        g.close()
        assert raised, "rank %d did not learn of the failure" % rank
        if rank != world - 1:
            assert "rank %d reported a failed" % (world - 1) in raised[0], raised
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_gatherer_failure_reaches_every_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gatherer_failure, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(3)], res


def _worker_bench_step_loop(rank, world, port, q):
    """bench.py's OWN step loop (bench.StepRunner: caller threads, two buffers each, steps taken in order, buffer hand-back) under two
    real processes, both ways it is timed at N > 1 -- sizes exchanged per step (`value`) and the gather to rank 0 inside every step
    (`with_gather`) -- with the CPU oracle standing in for the encoder, and the two reports rank 0 prints (`per_rank`, `with_gather`)."""
    sys.path.insert(0, ROOT)
    import threading
    import torch
    import torch.distributed as dist
    import bench
    from oracle import sqy_oracle as o
    from sqeazy_amd import multi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        inflight, cap, pipe = 3, 1 << 15, "bitswap1->lz4"
        Z, Y, X = 6, 16, 32
        # rank r holds frames [r*Z, (r+1)*Z) of an (N*Z, Y, X) stack (bench.py's weak-scaling layout); every step another seed
        vols = [synth.stack((world * Z, Y, X), seed=500 + s)[rank * Z:(rank + 1) * Z] for s in range(7)]
        blobs = [o.pipeline_encode(pipe, v) for v in vols]
        outs = [[torch.zeros(cap, dtype=torch.uint8) for _ in range(2)] for _ in range(inflight)]
        busy = [[False, False] for _ in range(inflight)]           # a buffer handed to the gatherer must not be encoded into before it comes back
        count = [0] * inflight
        lock = threading.Lock()
        step_of = {}

        def encode(t, b):
            with lock:
                assert not busy[t][b], "thread %d encodes into buffer %d while the gather still reads it" % (t, b)
                s = count[t] * inflight + t                       # the step this call is (thread t takes t, t + inflight, ..)
                count[t] += 1
            blob = blobs[s % len(blobs)]
            off = 11 + 3 * t + b                                  # frames in place: the blob starts somewhere inside its buffer
            outs[t][b][off:off + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            with lock:
                step_of[(t, b)] = s
                busy[t][b] = gathering[0]
            return off, len(blob)

        gathering = [False]
        index_rows = [torch.zeros(world, dtype=torch.int64) for _ in range(4)]
        gatherer = multi.SlabGatherer(world * cap, dev)
        posted = []

        class Watch:                                              # the gatherer, with the hand-back observed
            stats = gatherer.stats

            def post(self, view, n, on_done=None):
                posted.append(n)

                def done(od=on_done, key=len(posted) - 1):
                    with lock:
                        for t in range(inflight):
                            for b in range(2):
                                if outs[t][b].data_ptr() <= view.data_ptr() < outs[t][b].data_ptr() + cap:
                                    busy[t][b] = False
                    od()
                gatherer.post(view, n, on_done=done)

            def drain(self):
                gatherer.drain()

        runner = bench.StepRunner(inflight, encode, dist_on=True, blob_view=lambda t, b, off: outs[t][b][off:],
                                  exchange=lambda n, slot: multi.exchange_sizes(n, dev, out=index_rows[slot % len(index_rows)], sync=False),
                                  gatherer=Watch())
        # ---- `value`: blobs stay where they are, the sizes are all_gathered per step ----
        k = 7
        last = runner.run_steps(k, gather=False)
        assert last == len(blobs[(k - 1) % len(blobs)])
        dist.barrier()
        sizes_all = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes_all, torch.tensor([len(blobs[(k - 1) % len(blobs)])], dtype=torch.int64))
        assert index_rows[(k - 1) % len(index_rows)].tolist() == [int(x.item()) for x in sizes_all]      # the last step's index row
        t, b, off = runner.last_at
        assert bytes(outs[t][b][off:off + last].numpy().tobytes()) == blobs[(k - 1) % len(blobs)]
        # ---- `with_gather`: the gather to rank 0 inside every step, overlapped; the buffer comes back when its gather is done ----
        count[:] = [0] * inflight
        gathering[0] = True
        g0 = dict(gatherer.stats)
        local = []
        import time
        for _ in range(2):                                        # two timed blocks, like timed_blocks()
            t0 = time.perf_counter()
            base = [c for c in count]
            runner.run_steps(k, gather=True)
            t_own = time.perf_counter() - t0
            dist.barrier()
            local.append((t_own, time.perf_counter() - t0))
            assert all(not x for tb in busy for x in tb), "a buffer was not handed back"
            if rank == 0:
                szs, flat = gatherer.last
                got = multi.unpack_container(multi.pack_container(szs, flat))
                assert len(got) == world and got[0] == blobs[(k - 1) % len(blobs)]
                assert np.array_equal(o.pipeline_decode(got[1]), synth.stack((world * Z, Y, X), seed=500 + (k - 1) % len(blobs))[Z:2 * Z])
            count[:] = [0] * inflight
        gs = bench.gather_stats_delta(g0, dict(gatherer.stats))
        assert gs["gathers_timed"] == 2 * k and len(posted) == 2 * k
        # bytes of ALL ranks per step, averaged over the steps of the blocks
        mine = torch.tensor([sum(len(blobs[s % len(blobs)]) for s in range(k))], dtype=torch.int64)
        dist.all_reduce(mine)
        assert gs["bytes_gathered_per_step"] == int(mine.item() * 2 / (2 * k))
        assert gs["gather_ms_per_step"] > 0
        rep = bench.per_rank_rows(local, k, world, dev, rank)
        assert rep["world_size_rccl"] == world and rep["world_size_env"] == world and rep["backend"] == "gloo"
        assert rep["device_index"] == list(range(world)) and len(rep["ms_per_step_own"]) == world
        assert all(a > 0 and f >= a - 1e-9 for a, f in zip(rep["ms_per_step_own"], rep["ms_per_step_fenced"]))
        runner.close()
        gatherer.close()
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-800:]))
    finally:
        dist.destroy_process_group()


def test_bench_step_loop_world2():
    """the N > 1 path of bench.py -- its own step loop, not only multi.*'s pieces -- under two processes"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_step_loop, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
