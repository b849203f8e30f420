"""encode of one config, three times, per-kernel device times (GPU box):  python3 tools/enc_time.py c2|c3|c4|c5 [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
label = sys.argv[2] if len(sys.argv) > 2 else ""
pipeline, shape, dtype = {"c2": ("bitswap1->lz4", (512, 1024, 1024), np.uint16), "c3": ("diff3x3x1->bitswap1->lz4", (256, 2048, 2048), np.uint16),
                          "c4": ("frame_shuffle->lz4", (1024, 1024, 1024), np.uint8), "c5": ("quantiser->bitswap1->lz4", (256, 2048, 2048), np.uint16)}[which]
dev = torch.device("cuda", 0)
vol = synth.stack_torch(shape, dtype, dev)
cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype) + (1 << 16)
out = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap)
sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
for _ in range(3):
    rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap)
    torch.cuda.synchronize()
sqeazy_amd.profile_enable(False)
p = sqeazy_amd.profile_get()
print(label, which, "encode rc", rc, "bytes", m, " ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in p.items()), flush=True)
