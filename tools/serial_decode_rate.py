"""decode rate of the serial (nthreads = 1, ONE block-linked frame) layout next to the chunked one, device to device (GPU box)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
dev = torch.device("cuda", 0)
fn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device
for shape in ((64, 1024, 1024), (512, 1024, 1024)):
    vol = synth.stack_torch(shape, np.uint16, dev)
    nb = vol.numel() * 2
    cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    back = torch.empty(nb, dtype=torch.uint8, device=dev)
    for nt in (1, 0):
        rc, n = sqeazy_amd.encode_device("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap, nthreads=nt)
        assert rc == 0
        for rep in range(2):
            sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = fn(ctypes.c_void_p(out.data_ptr()), ctypes.c_long(n), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            sqeazy_amd.profile_enable(False)
        ok = bool((back.view(torch.uint16).reshape(shape) == vol).all().item())
        print("%s nthreads=%d decode: rc %d, %.1f ms = %.2f GB/s, equal %s | %s" % (shape, nt, rc, dt * 1e3, nb / dt / 1e9, ok,
              "  ".join("%s %.2f" % (k, v[0] / v[1]) for k, v in sqeazy_amd.profile_get().items())), flush=True)
