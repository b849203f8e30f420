"""per-kernel device times for the BASELINE configs (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
sqeazy_amd.lib()
dev = torch.device("cuda", 0)
def run(pipeline, shape, dtype, reps=3, extra=0):
    vol = synth.stack_torch(shape, dtype, dev)
    cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype) + extra
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap); assert rc == 0
    sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    sqeazy_amd.profile_enable(False)
    p = sqeazy_amd.profile_get()
    nb = vol.numel() * vol.element_size()
    print("%-28s %s %s: %.2f ms/call = %.1f GB/s, out %.1f MiB | " % (pipeline, shape, np.dtype(dtype).name, dt * 1e3, nb / dt / 1e9, m / 2**20) +
          "  ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in p.items()), flush=True)
    # decode timing
    back = torch.empty(nb, dtype=torch.uint8, device=dev)
    L = sqeazy_amd.lib(); import ctypes
    fn = L.SQYAMD_Decode_UI16_Device if np.dtype(dtype) == np.uint16 else L.SQYAMD_Decode_UI8_Device
    rc = fn(ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
    sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = fn(ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
    torch.cuda.synchronize(); dd = time.perf_counter() - t0
    sqeazy_amd.profile_enable(False)
    pd = sqeazy_amd.profile_get()
    ok = bool((back.view(torch.uint16 if np.dtype(dtype) == np.uint16 else torch.uint8).reshape(shape) == vol).all().item()) if "quantiser" not in pipeline and "frame_shuffle" not in pipeline else None
    print("    decode rc %d: %.2f ms = %.1f GB/s, round trip equal: %s | " % (rc, dd * 1e3, nb / dd / 1e9, ok) +
          "  ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in pd.items()), flush=True)
    del vol, out, back; torch.cuda.empty_cache()
run("bitswap1->lz4", (512, 1024, 1024), np.uint16)
run("diff3x3x1->bitswap1->lz4", (256, 2048, 2048), np.uint16)
run("frame_shuffle->lz4", (1024, 1024, 1024), np.uint8, extra=1 << 16)
run("quantiser->bitswap1->lz4", (256, 2048, 2048), np.uint16)
