"""chunk_split on / off on the bench stack and a slab, one call at a time: kernel times, how many chunks were nominated (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sqeazy_amd
from sqeazy_amd import synth
dev = torch.device("cuda", 0)
sqeazy_amd.set_option("block_parallel_stats", 1)
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sqeazy_amd.set_option("chunk_split_segments", nseg)
if len(sys.argv) > 2:
    sqeazy_amd.set_option("chunk_split_slots", int(sys.argv[2]))
print("segments", nseg, "slots", sqeazy_amd.get_option("chunk_split_slots"))
for shape in ((256, 256, 256), (128, 512, 512), (256, 512, 512), (256, 1024, 1024), (512, 1024, 1024)):
    vol = synth.stack_torch(shape, np.uint16, dev)
    cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    for mode in (0, 1, 0, 1):
        sqeazy_amd.set_option("chunk_split", mode)
        best = None
        for rep in range(4):
            sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc, off, n = sqeazy_amd.encode_device_at("bitswap1->lz4", vol.data_ptr(), shape, np.uint16, out.data_ptr(), cap)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            sqeazy_amd.profile_enable(False)
            if rep == 1:
                sqeazy_amd.set_option("block_parallel_stats", 0)
            if best is None or dt < best[0]:
                best = (dt, sqeazy_amd.profile_get())
        sqeazy_amd.set_option("block_parallel_stats", 1)
        print(shape, "chunk_split", mode, "rc", rc, "%.3f ms" % (best[0] * 1e3), " ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in best[1].items()), flush=True)
