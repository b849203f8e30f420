import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sqeazy_amd
v = np.random.default_rng(5).integers(0, 256, (40000, 4, 4), dtype=np.uint8)
print("cap", sqeazy_amd.max_compressed_length("frame_shuffle->lz4", v.shape, np.uint8))
rc, blob = sqeazy_amd.encode("frame_shuffle->lz4", v, nthreads=2)
print(rc, None if blob is None else len(blob))
