import os, sys, time, threading, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
sqeazy_amd.lib()
dev = torch.device("cuda", 0)
shape = (512, 1024, 1024)
vol = synth.stack_torch(shape, np.uint16, dev)
cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
streams = [torch.cuda.Stream(device=dev) for _ in range(M)]
outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(M)]
torch.cuda.synchronize()
log = []
def worker(t, k):
    torch.cuda.set_device(0)
    for i in range(k):
        t0 = time.perf_counter()
        rc, n = sqeazy_amd.encode_device("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, outs[t].data_ptr(), cap, nthreads=0, stream=streams[t].cuda_stream)
        log.append((t, i, (time.perf_counter() - t0) * 1e3))
T0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=(t, 6)) for t in range(M)]
[th.start() for th in ths]; [th.join() for th in ths]
print("total %.1f ms for %d calls" % ((time.perf_counter() - T0) * 1e3, 6 * M))
for t in range(M):
    print("thread", t, " ".join("%.1f" % ms for (tt, i, ms) in log if tt == t))
