"""PCIe-inclusive rate of the reference-protocol (host pointer) C-ABI on pageable host memory: SQY_PipelineEncode_UI16 and
SQY_Decode_UI16 called through ctypes with caller buffers that are allocated and touched beforehand (what a C caller has)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
shape = (512, 1024, 1024)
vol = synth.stack_torch(shape, np.uint16, torch.device("cuda", 0)).cpu().numpy()
L = sqeazy_amd.lib()
cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
dst = np.zeros(cap, np.uint8)                      # touched
shp = (ctypes.c_long * 3)(*shape)
n = ctypes.c_long(0)
L.SQY_PipelineEncode_UI16.restype = ctypes.c_int
for i in range(4):
    t = time.perf_counter()
    rc = L.SQY_PipelineEncode_UI16(b"bitswap1->lz4", ctypes.c_void_p(vol.ctypes.data), shp, 3, ctypes.c_void_p(dst.ctypes.data), ctypes.byref(n), 0)
    dt = time.perf_counter() - t
    print("SQY_PipelineEncode_UI16 1024x1024x512 (host pointers): rc %d, %.1f ms, %.2f GB/s of voxels (H2D %.2f GB + kernels + D2H %.2f GB)" % (
        rc, dt * 1e3, vol.nbytes / dt / 1e9, vol.nbytes / 1e9, n.value / 1e9))
back = np.zeros(vol.size, np.uint16)
for i in range(4):
    t = time.perf_counter()
    rc = L.SQY_Decode_UI16(ctypes.c_void_p(dst.ctypes.data), ctypes.c_long(n.value), ctypes.c_void_p(back.ctypes.data), 0)
    dt = time.perf_counter() - t
    print("SQY_Decode_UI16: rc %d, %.1f ms, %.2f GB/s of voxels, equal %s" % (rc, dt * 1e3, vol.nbytes / dt / 1e9, bool(np.array_equal(back.reshape(shape), vol))))
