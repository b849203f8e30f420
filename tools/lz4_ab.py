"""A/B timing of the lz4_chunks kernel of the INSTALLED library (GPU box): sample plane files, the bench stack and its
shell planes, the C3 / C5 slabs.  Prints one line per case; swap sqeazy_amd/lib/libsqeazy_amd.so between runs.
    python tools/lz4_ab.py [tag] [--full]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth
sqeazy_amd.lib()
dev = torch.device("cuda", 0)
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "lib"
full = "--full" in sys.argv


def run(name, t, pipeline="lz4", shape=None, dtype=np.uint8, reps=5):
    shape = shape or (1, 1, t.numel())
    cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, m = sqeazy_amd.encode_device(pipeline, t.data_ptr(), shape, dtype, out.data_ptr(), cap)
    assert rc == 0
    digest = hashlib.sha256(out[:m].cpu().numpy().tobytes()).hexdigest()[:12] if m < (64 << 20) else "-"
    best = {}
    for _ in range(reps):
        sqeazy_amd.profile_reset(); sqeazy_amd.profile_enable(True)
        rc, m = sqeazy_amd.encode_device(pipeline, t.data_ptr(), shape, dtype, out.data_ptr(), cap)
        sqeazy_amd.profile_enable(False)
        for k, v in sqeazy_amd.profile_get().items():
            best[k] = min(best.get(k, 1e9), v[0] / v[1])
    print("%-6s %-26s out %9d %s | " % (tag, name, m, digest) + "  ".join("%s %.3f" % (k, v) for k, v in best.items()), flush=True)


here = os.path.dirname(os.path.abspath(__file__))
for f in ("_plane11", "_c3plane", "_c5plane"):
    p = os.path.join(here, f + ".bin")
    if os.path.exists(p):
        d = torch.from_numpy(np.fromfile(p, np.uint8)).to(dev)
        run(f, d)
vol = synth.stack_torch((512, 1024, 1024), np.uint16, dev)
run("C2 bitswap1->lz4", vol, "bitswap1->lz4", (512, 1024, 1024), np.uint16)
cap = sqeazy_amd.max_compressed_length("bitswap1", (512, 1024, 1024), np.uint16)
planes = torch.empty(cap, dtype=torch.uint8, device=dev)
rc, m = sqeazy_amd.encode_device("bitswap1", vol.data_ptr(), (512, 1024, 1024), np.uint16, planes.data_ptr(), cap)
hdr = m - vol.numel() * 2
seg = vol.numel() * 2 // 16
body = planes[hdr:hdr + vol.numel() * 2].clone()
for p in ((3, 4, 5, 6, 7, 15) if not full else range(16)):
    run("C2 plane bit %d" % (15 - p), body[p * seg:(p + 1) * seg])
del planes, body, vol
def planes_of(pipeline, v, shape, dtype, elem_out):
    cap = sqeazy_amd.max_compressed_length(pipeline, shape, dtype)
    buf = torch.empty(cap, dtype=torch.uint8, device=dev)
    rc, m = sqeazy_amd.encode_device(pipeline, v.data_ptr(), shape, dtype, buf.data_ptr(), cap)
    assert rc == 0
    nbytes = int(np.prod(shape)) * elem_out
    return buf[m - nbytes:m].clone(), nbytes


if "--c3planes" in sys.argv or "--c5planes" in sys.argv:
    v = synth.stack_torch((256, 2048, 2048), np.uint16, dev)
    if "--c3planes" in sys.argv:
        body, nb = planes_of("diff3x3x1->bitswap1", v, (256, 2048, 2048), np.uint16, 2)
        for p in range(16):
            run("C3 plane bit %d" % (15 - p), body[p * (nb // 16):(p + 1) * (nb // 16)], reps=2)
    if "--c5planes" in sys.argv:
        body, nb = planes_of("quantiser->bitswap1", v, (256, 2048, 2048), np.uint16, 1)
        for p in range(8):
            run("C5 plane bit %d" % (7 - p), body[p * (nb // 8):(p + 1) * (nb // 8)], reps=2)
    del v
if full or "--slabs" in sys.argv:
    v = synth.stack_torch((256, 2048, 2048), np.uint16, dev)
    run("C3 slab", v, "diff3x3x1->bitswap1->lz4", (256, 2048, 2048), np.uint16, reps=2)
    run("C5 slab", v, "quantiser->bitswap1->lz4", (256, 2048, 2048), np.uint16, reps=2)
    del v
    v8 = synth.stack_torch((1024, 1024, 1024), np.uint8, dev)
    run("C4", v8, "frame_shuffle->lz4", (1024, 1024, 1024), np.uint8, reps=2)
