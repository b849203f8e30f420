"""decode of one config, three times, per-kernel device times (GPU box):  python3 tools/dec_time.py c2|c3|c5 [label]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
label = sys.argv[2] if len(sys.argv) > 2 else ""
pipeline, shape = {"c2": ("bitswap1->lz4", (512, 1024, 1024)), "c3": ("diff3x3x1->bitswap1->lz4", (256, 2048, 2048)),
                   "c5": ("quantiser->bitswap1->lz4", (256, 2048, 2048))}[which]
dev = torch.device("cuda", 0)
vol = synth.stack_torch(shape, np.uint16, dev)
cap = sqeazy_amd.max_compressed_length(pipeline, shape, np.uint16)
out = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap); assert rc == 0
nb = vol.numel() * 2
back = torch.empty(nb, dtype=torch.uint8, device=dev)
fn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device
rc = fn(ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
for _ in range(3):
    rc = fn(ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
    torch.cuda.synchronize()
sqeazy_amd.profile_enable(False)
p = sqeazy_amd.profile_get()
print(label, which, "decode rc", rc, " ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in p.items()), flush=True)
