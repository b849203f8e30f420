"""one encode + two decodes of a config (GPU box), for rocprofv3 passes over the decode kernels:  python3 tools/dec_one.py c2|c3|c4|c5"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
pipeline, shape, dtype = {"c2": ("bitswap1->lz4", (512, 1024, 1024), np.uint16), "c3": ("diff3x3x1->bitswap1->lz4", (256, 2048, 2048), np.uint16),
                          "c4": ("frame_shuffle->lz4", (1024, 1024, 1024), np.uint8), "c5": ("quantiser->bitswap1->lz4", (256, 2048, 2048), np.uint16)}[which]
dev = torch.device("cuda", 0)
vol = synth.stack_torch(shape, dtype, dev)
cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype) + (1 << 16)
out = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, off, m = sqeazy_amd.encode_device_at(pipeline, vol.data_ptr(), shape, dtype, out.data_ptr(), cap); assert rc == 0
nb = vol.numel() * vol.element_size()
back = torch.empty(nb, dtype=torch.uint8, device=dev)
fn = sqeazy_amd.lib().SQYAMD_Decode_UI16_Device if dtype == np.uint16 else sqeazy_amd.lib().SQYAMD_Decode_UI8_Device
for _ in range(2):
    rc = fn(ctypes.c_void_p(out.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
    torch.cuda.synchronize()
print("decode rc", rc)
