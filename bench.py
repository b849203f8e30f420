#!/usr/bin/env python3
"""bench.py -- encode GB/s (input voxels) for the 'bitswap1->lz4' uint16 pipeline on N x MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run,
one rank per GPU).  A "step" is one pass of the hot path over one z-slab: every rank encodes a
1024x1024x512 uint16 synthetic stack (BASELINE.json configs[1]; 1 GiB, already resident in HBM) with ONE
C-ABI call; the slab blobs are independent sqeazy blobs and stay on their GPUs, only their sizes (the index of
the sharded container) are all_gathered over RCCL (N>1 only; --gather-to-root also moves the blobs to rank 0).
Weak scaling: per-GPU work is fixed.  value = (N * input bytes * K) / max-over-ranks wall time, GB = 1e9 bytes.

The JSON line also carries
  roofline      the dominant kernel (largest share of device time, timed with HIP events on the launch
                stream inside the timed region) priced at the path's ALGORITHMIC bytes per call
                (2 B per voxel read once + payload bytes written once: SURVEY.md 8(d)) against 8 TB/s.
  cpu_baseline  the same pipeline on the host cores: the reference's own SSE bit-plane gather + liblz4
                1.9.3 frames (oracle/_ref, "reference") when that library loads, else our C restatement
                ("port"); bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PIPELINE = "bitswap1->lz4"
SHAPE = (512, 1024, 1024)          # {z,y,x}: 1024x1024x512 voxels, uint16 -> 1 GiB
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(sample_frames=96, reps=3):
    """bitswap1->lz4 on the host cores over the first `sample_frames` frames of the same synthetic stack"""
    from sqeazy_amd import synth
    from oracle import ref, sqy_oracle
    cores = os.cpu_count() or 1
    vol = synth.stack((sample_frames, SHAPE[1], SHAPE[2]), np.uint16)
    nbytes = vol.nbytes
    use_ref = ref.available()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        if use_ref:
            planes = ref.bitswap1_encode_u16(vol, nthreads=min(cores, 16))
            enc = ref.lz4_encode_parallel(planes.view(np.uint8), nthreads=cores)
        else:
            planes = sqy_oracle.bitswap1_encode_planes(vol, nthreads=min(cores, 16))
            enc = sqy_oracle.lz4_encode_chunked(planes.view(np.uint8))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(nbytes / best / 1e9, 4), "unit": "GB/s", "cores": cores if use_ref else min(cores, 16),
            "kind": "reference" if use_ref else "port",
            "sample": "%dx%dx%d uint16 (first frames of the bench stack, %.0f MiB), best of %d, %s" % (
                SHAPE[2], SHAPE[1], sample_frames, nbytes / 2**20, reps,
                "reference SSE bitswap + liblz4 1.9.3 frames, OpenMP all cores" if use_ref
                else "C restatement: 16-plane bitswap OpenMP + serial LZ4 frames"),
            "payload_ratio": round(nbytes / enc.size, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=SHAPE[0], help="z extent per GPU (default: the BASELINE config)")
    ap.add_argument("--gather-to-root", action="store_true",
                    help="N>1: also move every rank's compressed blob to rank 0 inside the step (default: blobs stay sharded, only "
                         "their sizes -- the container index -- are exchanged)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="C-ABI calls in flight per GPU (host threads, one stream + workspace each; the C-ABI is re-entrant like "
                         "the reference's).  1 = strictly one call after the other.  Two already keep the GPU busy (the LZ4 parse "
                         "of one call overlaps the HBM-bound kernels of the other); more only stretch each kernel's duration")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sqeazy_amd
    from sqeazy_amd import multi, synth

    sqeazy_amd.lib()   # fails loudly when the HIP library has not been built
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path (RCCL init, size exchange, barrier, max-reduce) on a one-GPU box: a group of one
    dist_on = world > 1 or os.environ.get("SQY_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    shape = (args.frames, SHAPE[1], SHAPE[2])
    # rank r holds frames [r*Z, (r+1)*Z) of an (N*Z, Y, X) synthetic stack
    vol = synth.stack_torch(shape, np.uint16, dev, z_offset=rank * shape[0], z_total=world * shape[0])
    nbytes = vol.numel() * 2
    cap = sqeazy_amd.max_compressed_length(PIPELINE, shape, np.uint16)
    gather_buf = torch.empty(world * cap, dtype=torch.uint8, device=dev) if (dist_on and rank == 0 and args.gather_to_root) else None
    index_rows = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(8)] if dist_on else []   # container index of the last steps
    import queue
    import threading
    sys.setswitchinterval(1e-4)      # caller threads hand the GIL over promptly (default 5 ms would show up as whole milliseconds per step)
    inflight = max(1, args.inflight)
    # every caller thread owns a stream and two output buffers (the gather of step s may still read one while s+inflight encodes)
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
    outs = [[torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(inflight)]
    out = outs[0][0]
    torch.cuda.synchronize()

    def run_steps(k):
        """k steps: thread t encodes steps t, t+inflight, ...; the main thread gathers the blobs in step order"""
        done_q = queue.Queue()
        free_q = [queue.Queue() for _ in range(inflight)]
        for fq in free_q:
            fq.put(0)
            fq.put(1)
        errors = []

        def worker(t):
            try:
                torch.cuda.set_device(local_rank)
                for s_ in range(t, k, inflight):
                    b = free_q[t].get()
                    rc, n = sqeazy_amd.encode_device(PIPELINE, vol.data_ptr(), shape, np.uint16, outs[t][b].data_ptr(), cap, nthreads=0,
                                                     stream=streams[t].cuda_stream)
                    if rc:
                        raise RuntimeError("SQYAMD_PipelineEncode_UI16_Device returned %d" % rc)
                    done_q.put((s_, t, b, n))
            except Exception as e:   # pragma: no cover
                errors.append(e)
                done_q.put((-1, t, 0, 0))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(min(inflight, max(k, 1)))]
        for th in threads:
            th.start()
        pending, nxt, last_n = {}, 0, 0
        t_start = time.perf_counter()
        while nxt < k:
            s_, t, b, n = done_q.get()
            if s_ < 0:
                break
            pending[s_] = (t, b, n)
            while nxt in pending:
                t2, b2, n2 = pending.pop(nxt)
                if dist_on:
                    if args.gather_to_root:
                        multi.gather_blobs(outs[t2][b2], n2, dst_buffer=gather_buf)
                        torch.cuda.current_stream().synchronize()
                    else:
                        multi.exchange_sizes(n2, dev, out=index_rows[nxt % len(index_rows)], sync=False)   # enqueued; the closing fence waits for it
                free_q[t2].put(b2)
                last_n = n2
                nxt += 1
                if os.environ.get("SQY_BENCH_TRACE"):
                    print("step %d done at %.2f ms (thread %d)" % (nxt - 1, (time.perf_counter() - t_start) * 1e3, t2), file=sys.stderr)
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        return last_n

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(inflight)          # untimed priming: every caller thread's context allocates its HBM workspace once
    payload = run_steps(args.warmup) if args.warmup else 0
    sqeazy_amd.profile_reset()
    sqeazy_amd.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    payload = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    sqeazy_amd.profile_enable(False)
    prof = sqeazy_amd.profile_get()

    # one call at a time (nothing else in flight), for the record: latency of the call and the kernels' undisturbed durations
    fence()
    sqeazy_amd.profile_reset()
    sqeazy_amd.profile_enable(True)
    single_call_ms = None
    for _ in range(3):
        tl = time.perf_counter()
        rc, _n = sqeazy_amd.encode_device(PIPELINE, vol.data_ptr(), shape, np.uint16, outs[0][0].data_ptr(), cap, nthreads=0,
                                          stream=streams[0].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tl) * 1e3
        single_call_ms = ms if single_call_ms is None else min(single_call_ms, ms)
    sqeazy_amd.profile_enable(False)
    prof_alone = sqeazy_amd.profile_get()

    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_in = world * nbytes * args.steps
        hdr = sqeazy_amd.header_size(bytes(out[:4096].cpu().numpy().tobytes()))
        payload_bytes = payload - hdr
        # dominant kernel by device time
        dom, (dom_ms, dom_n) = max(prof.items(), key=lambda kv: kv[1][0]) if prof else ("none", (0.0, 0))
        avg_ms = dom_ms / max(dom_n, 1)
        algo_bytes = nbytes + payload_bytes                  # 2 B/voxel read once + payload written once
        achieved = (algo_bytes / 1e9) / (avg_ms / 1e3) if avg_ms else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(dom)
            except Exception:
                traffic = None
        line = {
            "metric": "encode GB/s (input voxels) for bitswap1->lz4 uint16 volume",
            "value": round(total_in / dt / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": "%dx%dx%d uint16 synthetic microscopy stack per GPU, pipeline '%s', one C-ABI call per step, %d calls in flight%s" % (
                shape[2], shape[1], shape[0], PIPELINE, inflight, (", RCCL gather of the compressed slabs to rank 0" if args.gather_to_root else ", slab blobs stay sharded, sizes all_gathered over RCCL") if world > 1 else ""),
                "input_bytes_per_gpu": nbytes, "payload_bytes": payload_bytes, "blob_bytes": payload, "calls_in_flight_per_gpu": inflight,
                "single_call_latency_ms": round(single_call_ms, 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(avg_ms, 4),
                         # the same kernel with the GPU to itself (one call at a time, measured right after the timed region)
                         "alone_launch_ms": round(prof_alone[dom][0] / max(prof_alone[dom][1], 1), 4) if dom in prof_alone else None,
                         "alone_frac": round((algo_bytes / 1e9) / (prof_alone[dom][0] / max(prof_alone[dom][1], 1) / 1e3) / HBM_PEAK_GBS, 5)
                         if dom in prof_alone and prof_alone[dom][0] else None,
                         "kernels_ms_per_step": {k: round(v[0] / max(args.steps, 1), 4) for k, v in prof.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # the baseline is reported, never required
                line["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
