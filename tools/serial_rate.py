"""rate of the serial (nthreads = 1) block-linked LZ4 layout on the bench stack (GPU box) next to the chunked layout; every case three times
(the first call allocates the workspace); SQY_NO_BLOCK_PARALLEL=1 gives the frame walk of rounds 2-3 (one wavefront)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
dev = torch.device("cuda", 0)
for shape in ((64, 1024, 1024), (512, 1024, 1024)):
    vol = synth.stack_torch(shape, np.uint16, dev)
    cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    for nt in (1, 1, 1, 0, 0):
        sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, n = sqeazy_amd.encode_device("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap, nthreads=nt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        sqeazy_amd.profile_enable(False)
        print("%s nthreads=%d: rc %d, %.1f ms = %.2f GB/s, %d bytes | %s" % (shape, nt, rc, dt * 1e3, vol.numel() * 2 / dt / 1e9, n,
              "  ".join("%s %.2f" % (k, v[0] / v[1]) for k, v in sqeazy_amd.profile_get().items())), flush=True)
