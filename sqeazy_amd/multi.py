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
