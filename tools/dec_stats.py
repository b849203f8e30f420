"""what the two wavefronts of lz4_frames_decode2_kernel counted, per frame (a library built by tools/dec_stats.sh; GPU box).  The LZ4 stage is
run on its own -- the bit planes (c2) or the quantised planes (c5) of the bench data as a one-byte volume, pipeline "lz4" -- so that the
frames decode straight into the caller's buffer, where the diagnostic build leaves its counts."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda", 0)
front, shape = {"c2": ("bitswap1", (512, 1024, 1024)), "c5": ("quantiser->bitswap1", (256, 2048, 2048))}[which]
vol = synth.stack_torch(shape, np.uint16, dev)
cap = sqeazy_amd.max_compressed_length(front, shape, np.uint16)
o1 = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, off, m = sqeazy_amd.encode_device_at(front, vol.data_ptr(), shape, np.uint16, o1.data_ptr(), cap); assert rc == 0
nb = vol.numel() * (2 if which == "c2" else 1)
planes = o1[off + m - nb: off + m].clone()
del vol, o1
CH = 262144
shape8 = (nb // CH, 512, 512)
cap = sqeazy_amd.max_compressed_length("lz4", shape8, np.uint8)
o2 = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, off, m = sqeazy_amd.encode_device_at("lz4", planes.data_ptr(), shape8, np.uint8, o2.data_ptr(), cap); assert rc == 0
back = torch.zeros(nb, dtype=torch.uint8, device=dev)
fn = sqeazy_amd.lib().SQYAMD_Decode_UI8_Device
sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
rc = fn(ctypes.c_void_p(o2.data_ptr() + off), ctypes.c_long(m), ctypes.c_void_p(back.data_ptr()), ctypes.c_long(nb), None)
torch.cuda.synchronize()
sqeazy_amd.profile_enable(False)
print("rc", rc, {k: round(v[0] / v[1], 3) for k, v in sqeazy_amd.profile_get().items()})
st = back.view(-1, CH)[:, :512].contiguous().cpu().numpy().view(np.uint64)
ok = (st[:, 8] < 300000) & (st[:, 8] > 0) & (st[:, 0] < 10**10) & (st[:, 1] < 10**6)
print("frames with counts:", int(ok.sum()))
key = np.where(ok, np.maximum(st[:, 0], st[:, 16]).astype(np.int64), -1)
print("frame | wave 0: cycles batches sequences-in-batches singles cyc-batches cyc-singles publishes cyc-publish block-bytes cyc-stage-waits ext-runs |"
      " wave 1: cycles units cyc-waiting round-units rounds cyc-rounds in-order-seqs cyc-in-order cyc-fills flushes cyc-flush short-matches cyc-short long-matches cyc-long"
      " far-matches cyc-far")
for f in np.argsort(-key)[:10]:
    print(f, "|", " ".join(str(int(x)) for x in st[f, :11]), "|", " ".join(str(int(x)) for x in st[f, 16:33]))
