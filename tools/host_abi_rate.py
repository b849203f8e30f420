"""PCIe-inclusive rate of the reference-protocol (host pointer) C-ABI: SQY_PipelineEncode_UI16 on pageable host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
shape = (512, 1024, 1024)
vol = synth.stack_torch(shape, np.uint16, torch.device("cuda", 0)).cpu().numpy()
for i in range(3):
    t = time.perf_counter(); rc, blob = sqeazy_amd.encode("bitswap1->lz4", vol, nthreads=0); dt = time.perf_counter() - t
    print("host ABI 1024x1024x512 u16 bitswap1->lz4: rc %d, %.1f ms, %.2f GB/s (H2D + kernels + D2H, pageable memory)" % (rc, dt * 1e3, vol.nbytes / dt / 1e9))
