"""times the lz4_chunks kernel on different byte streams (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
sqeazy_amd.lib()
dev = torch.device("cuda", 0)
n = 256 << 20

def run(name, t, pipeline="lz4", shape=None, dtype=np.uint8, reps=3):
    shape = shape or (1, 1, t.numel())
    cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    for _ in range(1):
        rc, m = sqeazy_amd.encode_device(pipeline, t.data_ptr(), shape, dtype, out.data_ptr(), cap)
        assert rc == 0
    sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
    for _ in range(reps):
        rc, m = sqeazy_amd.encode_device(pipeline, t.data_ptr(), shape, dtype, out.data_ptr(), cap)
    sqeazy_amd.profile_enable(False)
    p = sqeazy_amd.profile_get()
    nb = t.numel() * t.element_size()
    print("%-22s in %5d MiB out %5.1f MiB  " % (name, nb >> 20, m / 2**20) +
          "  ".join("%s %.3f ms" % (k, v[0] / v[1]) for k, v in p.items()), flush=True)

g = torch.Generator(device=dev); g.manual_seed(1)
run("zeros", torch.zeros(n, dtype=torch.uint8, device=dev))
run("random", torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g))
run("2level", torch.randint(0, 2, (n,), dtype=torch.uint8, device=dev, generator=g))
x = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
x[torch.rand(n, device=dev, generator=g) < 0.98] = 0
run("sparse2pct", x)
vol = synth.stack_torch((512, 1024, 1024), np.uint16, dev)
run("synth bitswap1->lz4", vol, "bitswap1->lz4", (512, 1024, 1024), np.uint16)
# per-plane cost: bitswap on device via the library (pipeline 'bitswap1'), then lz4 on each plane segment
cap = sqeazy_amd.max_compressed_length("bitswap1", (512, 1024, 1024), np.uint16)
planes = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, m = sqeazy_amd.encode_device("bitswap1", vol.data_ptr(), (512, 1024, 1024), np.uint16, planes.data_ptr(), cap)
hdr = m - vol.numel() * 2
seg = vol.numel() * 2 // 16
body = planes[hdr:hdr + vol.numel() * 2].clone()
for p in range(16):
    run("plane bit %d" % (15 - p), body[p * seg:(p + 1) * seg])
