"""z-slab sharding across the GPUs of one node.

A volume {Z,Y,X} is cut into `world_size` contiguous z-slabs; every rank encodes its slab on its own GPU
with one C-ABI call (each slab blob is a bit-exact sqeazy blob: this is exactly what the CPU reference
produces for the same slab call).  The path itself has no exchange step; what a sharded container needs is its
index: `exchange_sizes` (one 8-byte all_gather).  `gather_blobs` additionally moves the compressed blobs to rank 0
over RCCL/xGMI (backend "nccl" on ROCm) for callers that want one contiguous container there -- root ingress then
bounds the job (compressed bytes of all slabs per step).  Written against torch.distributed so that the same code
runs on gloo for the CPU tests.

Container produced on rank 0 (OUR framing, not sqeazy's):  u64 count | u64 size[count] | blob_0 | blob_1 ...
"""
import numpy as np


def slab_range(Z, rank, world):
    """contiguous z-range of `rank`: the first Z % world ranks get one frame more"""
    base, rem = divmod(int(Z), int(world))
    z0 = rank * base + min(rank, rem)
    return z0, base + (1 if rank < rem else 0)


def exchange_sizes(nbytes, device, group=None, out=None, sync=True):
    """all_gather of the per-rank blob sizes (8 bytes per rank): the container's index.  Every rank learns where its
    blob sits in the container (offset = sum of the sizes of the ranks before it); the blobs themselves stay on the
    GPUs that produced them -- slabs are independent sqeazy blobs, the path has no other exchange step.
    sync=False leaves the index on the device (tensor `out`, world int64) and does not block the host."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    size_t = torch.empty(1, dtype=torch.int64, device=device).fill_(int(nbytes))     # no host synchronisation
    sizes_t = out if out is not None else torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes_t, size_t, group=group)
    if not sync:
        return sizes_t                           # stays on the device; valid after the stream / the next fence
    return [int(v) for v in sizes_t.tolist()]


def gather_blobs(blob, nbytes, dst_buffer=None, group=None, root=0):
    """Variable-length gather of `blob[:nbytes]` (1-D uint8 torch tensor on this rank's device) to `root`.

    returns (sizes list, flat uint8 tensor with all blobs back to back) on root, (sizes, None) elsewhere.
    Sizes travel with one all_gather of an int64; payloads with batched point-to-point sends (ncclSend /
    ncclRecv on RCCL): every non-root rank sends exactly its compressed bytes, so the traffic on the
    root's xGMI links is the compressed payload only."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    size_t = torch.tensor([int(nbytes)], dtype=torch.int64, device=blob.device)
    sizes_t = [torch.zeros(1, dtype=torch.int64, device=blob.device) for _ in range(world)]
    dist.all_gather(sizes_t, size_t, group=group)
    sizes = [int(s.item()) for s in sizes_t]
    if world == 1:
        return sizes, blob[:nbytes]
    if rank == root:
        total = sum(sizes)
        if dst_buffer is None or dst_buffer.numel() < total:
            dst_buffer = torch.empty(total, dtype=torch.uint8, device=blob.device)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ops = []
        for r in range(world):
            view = dst_buffer[int(offs[r]):int(offs[r + 1])]
            if r == root:
                view.copy_(blob[:nbytes])
            elif sizes[r]:
                ops.append(dist.P2POp(dist.irecv, view, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return sizes, dst_buffer[:total]
    if nbytes:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, blob[:nbytes], root, group)]):
            req.wait()
    return sizes, None


class SlabGatherer:
    """north_star's "final RCCL gather", overlapped: every rank posts its finished slab blob, a gather thread moves it to
    rank 0 (`gather_blobs`: sizes all_gather + ncclSend / ncclRecv on RCCL) while the caller threads already encode the next
    slabs.  Root ingress is double-buffered (`depth` buffers of `capacity_bytes`), so the container of step s stays readable
    while step s+1 arrives.  Every rank must post in the same order (the gather thread works first in, first out); no other
    collective may be issued while posts are outstanding -- `drain()` first."""

    def __init__(self, capacity_bytes, device, group=None, root=0, depth=2):
        import queue
        import threading
        import torch
        import torch.distributed as dist
        self.group, self.root, self.device = group, root, device
        self.is_root = dist.get_rank(group) == root
        self.bufs = [torch.empty(int(capacity_bytes), dtype=torch.uint8, device=device) for _ in range(depth)] if self.is_root else []
        self.done = 0                    # gathers completed
        self.last = None                 # (sizes, flat view into one of the ingress buffers) of the newest completed gather, on root
        self.error = None
        self._q = queue.Queue()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        import torch
        if getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                blob, nbytes, on_done = item
                if self.error is None:
                    try:
                        dst = self.bufs[self.done % len(self.bufs)] if self.is_root else None
                        self.last = gather_blobs(blob, nbytes, dst_buffer=dst, group=self.group, root=self.root)
                    except Exception as e:      # pragma: no cover
                        self.error = e
                self.done += 1
                if on_done is not None:
                    on_done()
            finally:
                self._q.task_done()

    def post(self, blob, nbytes, on_done=None):
        """queue `blob[:nbytes]` (must stay untouched until on_done runs) for the gather; returns at once"""
        self._q.put((blob, int(nbytes), on_done))

    def drain(self):
        """wait until every posted gather is complete (raises what the gather thread hit)"""
        self._q.join()
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def close(self):
        self._q.put(None)
        self._t.join()


def pack_container(sizes, flat):
    """u64 count | u64 sizes | blobs, as bytes (host side convenience for tests / file output)"""
    head = np.array([len(sizes)] + list(sizes), dtype=np.uint64).tobytes()
    return head + bytes(flat.cpu().numpy().tobytes())


def unpack_container(buf):
    buf = bytes(buf)
    count = int(np.frombuffer(buf[:8], dtype=np.uint64)[0])
    sizes = np.frombuffer(buf[8:8 + 8 * count], dtype=np.uint64).astype(np.int64)
    out, off = [], 8 + 8 * count
    for s in sizes:
        out.append(buf[off:off + int(s)])
        off += int(s)
    return out
