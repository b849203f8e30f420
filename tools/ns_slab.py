import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
sqeazy_amd.lib()
dev = torch.device("cuda", 0)
tag = sys.argv[1]
for name, pipe, shape, zoff, ztot in (("ns_slab3", "bitswap1->lz4", (256, 2048, 2048), 768, 2048), ("ns_slab0", "bitswap1->lz4", (256, 2048, 2048), 0, 2048),
                                      ("C3_slab", "diff3x3x1->bitswap1->lz4", (256, 2048, 2048), 0, 256)):
    v = synth.stack_torch(shape, np.uint16, dev, z_offset=zoff, z_total=ztot)
    cap = sqeazy_amd.max_compressed_length(pipe, shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(4):
        sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, off, m = sqeazy_amd.encode_device_at(pipe, v.data_ptr(), shape, np.uint16, out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        sqeazy_amd.profile_enable(False)
        if dt < best: best, prof = dt, sqeazy_amd.profile_get()
    print(tag, name, "%.3f ms" % (best * 1e3), m, {k: round(a / max(c, 1), 3) for k, (a, c) in prof.items() if "lz4_chunks" in k}, flush=True)
    del v, out; torch.cuda.empty_cache()
