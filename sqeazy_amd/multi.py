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
    # one all_gather, one read-back (nbytes < 0 is the poison value of a rank that failed earlier: every rank raises together)
    size_t = torch.tensor([int(nbytes)], dtype=torch.int64, device=blob.device)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=blob.device)
    dist.all_gather_into_tensor(sizes_t, size_t, group=group)
    sizes = [int(v) for v in sizes_t.tolist()]
    if min(sizes) < 0:
        raise RuntimeError("rank %d reported a failed encode / gather: no blobs are exchanged" % sizes.index(min(sizes)))
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
        # what the gather itself costs, for the record (bench.py's `with_gather`): wall time of gather_blobs on the gather thread
        # incl. the wait for its transfers, and the compressed bytes of all ranks it moved
        self.stats = {"gathers": 0, "bytes": 0, "seconds": 0.0}
        self.last = None                 # (sizes, flat view into one of the ingress buffers) of the newest completed gather, on root
        self.error = None
        self._poisoned = False
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
                # a rank that failed keeps taking part in the size exchange with the poison value -1: its peers then raise in the
                # same gather instead of waiting for sends that never come (until the RCCL time-out)
                try:
                    import time
                    dst = self.bufs[self.done % len(self.bufs)] if self.is_root else None
                    t0 = time.perf_counter()
                    res = gather_blobs(blob, nbytes if self.error is None else -1, dst_buffer=dst, group=self.group, root=self.root)
                    if getattr(self.device, "type", "cpu") == "cuda":
                        # RCCL's wait() only orders the transfers on this thread's stream: the blob goes back to its encoder thread
                        # (on_done) when they are DONE, not when they are queued
                        torch.cuda.current_stream().synchronize()
                    self.stats["gathers"] += 1
                    self.stats["bytes"] += int(sum(res[0]))
                    self.stats["seconds"] += time.perf_counter() - t0
                    if self.error is None:
                        self.last = res
                except Exception as e:
                    if self.error is None:
                        self.error = e
                self.done += 1
                if on_done is not None:
                    on_done()
            finally:
                self._q.task_done()

    def post(self, blob, nbytes, on_done=None):
        """queue `blob[:nbytes]` (must stay untouched until on_done runs) for the gather; returns at once.
        Raises what an earlier gather hit (on every rank: see _run), after which nothing more is queued."""
        if self.error is not None:
            # the peers post in step with this rank, and some of them may have queued several gathers before they notice: EVERY
            # later post of this rank still takes place, with the poison size, so that each of their queued gathers finds its
            # partner and raises instead of waiting in an all_gather nobody joins (round-3 advice: one poison gather was not enough)
            self._poisoned = True
            self._q.put((blob, -1, on_done))
            raise self.error
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


# ---------------------------------------------------------------------------------------------------------------------
# Single-blob mode ("optional"): N ranks produce ONE blob, byte-identical to what one call on the whole
# volume yields, for `bitswap1->lz4` in the chunked layout.  The whole volume's payload is the 16 (8) bit planes one after
# the other, each cut into LZ4 chunks that are compressed independently; a z-slab's bits are a contiguous piece of every
# plane.  When that piece is a whole number of chunks, the frames a rank gets for its slab ARE the frames of the whole
# volume for that piece -- the single blob is  header(whole shape) | plane 15: rank 0's frames, rank 1's, ... | plane 14: ...
# No halo, no re-encoding: every rank makes its ordinary slab call, the root re-orders byte ranges.
# ---------------------------------------------------------------------------------------------------------------------
LZ4_CHUNK_BYTES = 256 << 10          # sqeazy's default framestep (lz4.hpp:91-101)
SINGLE_BLOB_PIPELINE = "bitswap1->lz4"


def single_blob_possible(shape, dtype, world, chunk_bytes=LZ4_CHUNK_BYTES):
    """every rank's slab (slab_range) must cover a whole number of LZ4 chunks of every bit plane: voxels_r / 8 bytes per plane"""
    shape = [int(s) for s in shape]
    if len(shape) != 3 or np.dtype(dtype) not in (np.dtype(np.uint16), np.dtype(np.uint8)):
        return False
    per_frame = shape[1] * shape[2]
    # one blob = what ONE call on the whole volume yields, and one call is < 2^31 voxels (the reference counts them in an
    # int, dynamic_pipeline.hpp:565; SQY_Decode refuses headers that claim more): larger volumes stay sharded containers
    if shape[0] * per_frame >= 1 << 31:
        return False
    for r in range(world):
        _, nz = slab_range(shape[0], r, world)
        if nz == 0 or (nz * per_frame) % (8 * chunk_bytes):
            return False
    return True


def walk_frames(payload):
    """(offset, length) of every LZ4 frame in `payload` (bytes-like): magic, FLG/BD/HC, blocks (4-byte size field, bit 31 =
    stored), end mark.  Only what sqeazy writes (no content size, checksums or dictionary id)."""
    buf = memoryview(payload)
    out, off, n = [], 0, len(buf)
    while off < n:
        if n - off < 11 or bytes(buf[off:off + 4]) != b"\x04\x22\x4d\x18":
            raise ValueError("no LZ4 frame at payload offset %d" % off)
        p = off + 7
        while True:
            field = int.from_bytes(buf[p:p + 4], "little")
            p += 4
            if field == 0:
                break
            p += field & 0x7fffffff
            if p + 4 > n:
                raise ValueError("LZ4 frame runs past the payload")
        out.append((off, p - off))
        off = p
    return out


def plane_ranges(slab_blob, slab_shape, dtype, chunk_bytes=LZ4_CHUNK_BYTES):
    """header size and the byte range (offset into the blob, length) of every bit plane's frames inside one slab blob"""
    import sqeazy_amd
    blob = bytes(slab_blob)
    hdr = sqeazy_amd.header_size(blob[:65536])
    W = np.dtype(dtype).itemsize * 8
    plane_bytes = int(np.prod(slab_shape)) // 8
    if plane_bytes % chunk_bytes:
        raise ValueError("the slab is not a whole number of LZ4 chunks per bit plane")
    per_plane = plane_bytes // chunk_bytes
    frames = walk_frames(blob[hdr:])
    if len(frames) != W * per_plane:
        raise ValueError("%d frames where %d x %d were expected" % (len(frames), W, per_plane))
    ranges = []
    for p in range(W):
        first, last = frames[p * per_plane], frames[(p + 1) * per_plane - 1]
        ranges.append((hdr + first[0], last[0] + last[1] - first[0]))
    return hdr, ranges


def build_header(pipeline, dtype, shape, payload_bytes):
    """the header sqeazy puts in front of `payload_bytes` of payload (SQYAMD_Header_Build, host only)"""
    import ctypes
    import sqeazy_amd
    L = sqeazy_amd.lib()
    dims = (ctypes.c_long * len(shape))(*[int(s) for s in shape])
    n = ctypes.c_long(0)
    esz = np.dtype(dtype).itemsize
    if L.SQYAMD_Header_Build(pipeline.encode(), esz, dims, len(shape), ctypes.c_long(int(payload_bytes)), None, ctypes.byref(n)):
        raise ValueError("SQYAMD_Header_Build refused %r" % (pipeline,))
    buf = ctypes.create_string_buffer(n.value)
    cap = ctypes.c_long(n.value)
    if L.SQYAMD_Header_Build(pipeline.encode(), esz, dims, len(shape), ctypes.c_long(int(payload_bytes)), buf, ctypes.byref(cap)):
        raise ValueError("SQYAMD_Header_Build failed")
    return buf.raw[:cap.value]


def assemble_single_blob(shape, dtype, slab_blobs, ranges, pipeline=SINGLE_BLOB_PIPELINE):
    """slab_blobs[r]: rank r's slab blob (1-D uint8 torch tensor, any device, or bytes); ranges[r]: plane_ranges of it.
    Returns the whole volume's blob as a uint8 torch tensor on the device of the first slab blob (bytes in, bytes out)."""
    import torch
    W = np.dtype(dtype).itemsize * 8
    payload = sum(ln for rr in ranges for (_, ln) in rr)
    head = build_header(pipeline, dtype, shape, payload)
    as_bytes = not hasattr(slab_blobs[0], "device")
    if as_bytes:
        parts = [head]
        for p in range(W):
            for r, b in enumerate(slab_blobs):
                o, ln = ranges[r][p]
                parts.append(bytes(b[o:o + ln]))
        return b"".join(parts)
    dev = slab_blobs[0].device
    out = torch.empty(len(head) + payload, dtype=torch.uint8, device=dev)
    out[:len(head)] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(dev)
    pos = len(head)
    for p in range(W):
        for r, b in enumerate(slab_blobs):
            o, ln = ranges[r][p]
            out[pos:pos + ln] = b[o:o + ln]
            pos += ln
    return out


def plane_ranges_from_frame_offsets(frame_offsets, planes):
    """(header bytes, [(offset, length) per bit plane]) from the table SQYAMD_PipelineEncode_*_DeviceAt_Frames returns with
    every = chunks per bit plane: frame_offsets[p] = start of plane p's first frame, the last entry = blob length"""
    fo = [int(v) for v in frame_offsets]
    if len(fo) != planes + 1:
        raise ValueError("%d frame offsets where %d planes + 1 were expected" % (len(fo), planes))
    return fo[0], [(fo[p], fo[p + 1] - fo[p]) for p in range(planes)]


def gather_single_blob(blob, nbytes, shape, dtype, group=None, root=0, pipeline=SINGLE_BLOB_PIPELINE, frame_offsets=None):
    """Every rank passes the blob of ITS slab (slab_range of shape[0]); `root` gets the single blob of the whole volume
    (uint8 tensor on its device), everybody else None.  Exchange: one all_gather of 1 + W plane sizes, then the
    variable-length gather of the slab blobs (compressed bytes only).
    frame_offsets: what sqeazy_amd.encode_device_at_frames(..., every = chunks per bit plane) returned for this slab -- the plane
    ranges then come from the encoder's own frame table in HBM (W + 1 integers).  Without it the blob is copied to the host and
    its LZ4 frames are walked there (CPU tests, blobs from elsewhere)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if not single_blob_possible(shape, dtype, world):
        raise ValueError("slabs of %r over %d ranks are not whole LZ4 chunks per bit plane (or the volume has 2^31 or more voxels)" % (list(shape), world))
    _, nz = slab_range(shape[0], rank, world)
    W = np.dtype(dtype).itemsize * 8
    if frame_offsets is not None:
        hdr, rr = plane_ranges_from_frame_offsets(frame_offsets, W)
    else:
        hdr, rr = plane_ranges(blob[:nbytes].cpu().numpy().tobytes(), (nz, shape[1], shape[2]), dtype)
    mine = torch.tensor([hdr] + [v for pr in rr for v in pr], dtype=torch.int64, device=blob.device)
    table_t = torch.zeros(world * mine.numel(), dtype=torch.int64, device=blob.device)
    dist.all_gather_into_tensor(table_t, mine, group=group)
    sizes, flat = gather_blobs(blob, nbytes, group=group, root=root)
    if rank != root:
        return None
    table = np.asarray(table_t.tolist(), dtype=np.int64).reshape(world, -1)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    blobs, ranges = [], []
    for r in range(world):
        t = [int(v) for v in table[r]]
        blobs.append(flat[int(offs[r]):int(offs[r + 1])])
        ranges.append([(t[1 + 2 * p], t[2 + 2 * p]) for p in range(W)])
    return assemble_single_blob(shape, dtype, blobs, ranges, pipeline)
