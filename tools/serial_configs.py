"""the serial layout (nthreads = 1) next to the chunked one on the C3 / C5 slabs: encode and decode, per kernel (GPU box; the C5 encode takes
13.6 s: the sequence-heavy plane defeats the block-parallel parse).  SQY_BLOCK_PARALLEL_STATS=1 names the failed blocks."""
import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sqeazy_amd
from sqeazy_amd import synth
dev = torch.device("cuda", 0)
fn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device
for pipeline in ("diff3x3x1->bitswap1->lz4", "quantiser->bitswap1->lz4"):
    shape = (256, 2048, 2048)
    vol = synth.stack_torch(shape, np.uint16, dev)
    nb = vol.numel() * 2
    cap = sqeazy_amd.max_compressed_length(pipeline, shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    back = torch.empty(nb, dtype=torch.uint8, device=dev)
    for nt in (1, 0):
        for rep in range(2):
            sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc, n = sqeazy_amd.encode_device(pipeline, vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap, nthreads=nt)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            sqeazy_amd.profile_enable(False)
        pe = sqeazy_amd.profile_get()
        for rep in range(2):
            sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            drc = fn(ctypes.c_void_p(out.data_ptr()), ctypes.c_long(n), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
            torch.cuda.synchronize(); dd = time.perf_counter() - t0
            sqeazy_amd.profile_enable(False)
        pd = sqeazy_amd.profile_get()
        print("%s nthreads=%d: encode rc %d %.2f ms (%s) | decode rc %d %.2f ms (%s)" % (pipeline, nt, rc, dt * 1e3,
              " ".join("%s %.2f" % (k, v[0] / v[1]) for k, v in pe.items()), drc, dd * 1e3, " ".join("%s %.2f" % (k, v[0] / v[1]) for k, v in pd.items())), flush=True)
