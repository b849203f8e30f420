"""north_star's volume (2048^3 uint16, 'bitswap1->lz4') as ONE SQYAMD_PipelineEncode_Slabs_UI16_Device call: ms and roofline fraction against
the number of slab calls the library keeps in flight.   python tools/ns_volume.py [pipeline]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
pipe = sys.argv[1] if len(sys.argv) > 1 else "bitswap1->lz4"
dev = torch.device("cuda", 0)
Z = 2048
vol = torch.empty((Z, 2048, 2048), dtype=torch.uint16, device=dev)
for s in range(Z // 256):
    vol[256 * s:256 * (s + 1)] = synth.stack_torch((256, 2048, 2048), np.uint16, dev, z_offset=256 * s, z_total=Z)
cap = sqeazy_amd.max_compressed_length(pipe, (256, 2048, 2048), np.uint16)
out = torch.empty(cap * 8, dtype=torch.uint8, device=dev)
nbytes = vol.numel() * 2
for k in (3, 4, 5, 6, 8, 2, 3, 4):
    best, lens = 1e9, None
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, offs, lens = sqeazy_amd.encode_slabs_device(pipe, vol.data_ptr(), (Z, 2048, 2048), np.uint16, 8, out.data_ptr(), cap, inflight=k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert rc == 0
        best = min(best, dt)
    algo = nbytes + sum(lens)
    print("%s: %d slab calls in flight: %.2f ms = %.0f GB/s, roofline fraction %.3f (payload %.2f GB)" % (pipe, k, best * 1e3, nbytes / best / 1e9, algo / best / 8e12, sum(lens) / 1e9), flush=True)
