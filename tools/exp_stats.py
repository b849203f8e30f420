"""Experiment builds only (SQY_EXTRA_FLAGS=-DSQY_EXP_STATS python -m sqeazy_amd.build --force): what the parse waves of the bench stack
wait for, one call at a time and with four calls in flight.  Every first-pass parse wave logs start / end (s_memrealtime, 100 MHz), its
own clock ticks (s_memtime), the time and number of its waits for the ring's LDS-DMA and for far candidates, sequences, generic batches.
    python tools/exp_stats.py [inflight=4] [steps=6]
"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import sqeazy_amd
from sqeazy_amd import synth

L = sqeazy_amd.lib()
if not hasattr(L, "SQYAMD_Exp_Set_Buffer"):
    raise SystemExit("library built without -DSQY_EXP_STATS")
L.SQYAMD_Exp_Set_Buffer.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shape = (512, 1024, 1024)
CAP = 4096 * (M * K + 8)
buf = torch.zeros(8 + CAP * 8, dtype=torch.int64, device=dev)
assert L.SQYAMD_Exp_Set_Buffer(ctypes.c_void_p(buf.data_ptr()), CAP) == 0
sqeazy_amd.set_option("transpose_chain_caller_streams", 1)
vol = synth.stack_torch(shape, np.uint16, dev)
vols = [vol] + [vol.clone() for _ in range(M - 1)]
cap = sqeazy_amd.max_compressed_length("bitswap1->lz4", shape, np.uint16)
streams = [torch.cuda.Stream(device=dev) for _ in range(M)]
outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(M)]
entry = L.SQYAMD_PipelineEncode_UI16_DeviceAt
shape_c = (ctypes.c_long * 3)(*shape)


def prepared(t):
    doff, dlen = ctypes.c_long(0), ctypes.c_long(0)
    return (b"bitswap1->lz4", ctypes.c_void_p(vols[t].data_ptr()), shape_c, ctypes.c_uint(3), ctypes.c_void_p(outs[t].data_ptr()), ctypes.c_long(cap),
            ctypes.byref(doff), ctypes.byref(dlen), ctypes.c_int(0), ctypes.c_void_p(streams[t].cuda_stream)), doff, dlen


ARGS = [prepared(t) for t in range(M)]
sys.setswitchinterval(1e-4)


def call(t):
    assert entry(*ARGS[t][0]) == 0


def report(tag, wall_ms, ncalls):
    torch.cuda.synchronize()
    h = buf.cpu().numpy().view(np.uint64)
    n = int(min(h[0], CAP))
    r = h[8:8 + n * 8].reshape(n, 8)
    start, end, ticks = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2].astype(np.float64)
    dur = (end - start) * 0.01                                   # us
    wait_t, wait_n = (r[:, 3] & ((1 << 40) - 1)).astype(np.float64), (r[:, 3] >> 40).astype(np.int64)
    far_t, far_n = (r[:, 4] & ((1 << 40) - 1)).astype(np.float64), (r[:, 4] >> 40).astype(np.int64)
    chunk, nseq = (r[:, 5] & 0xffffffff).astype(np.int64), (r[:, 5] >> 32).astype(np.int64)
    ngen = (r[:, 7] & 0xffff).astype(np.int64)
    tu = 1.0 / 2400.0                                              # s_memtime tick (measured: 0.0004 us)
    acc = np.stack([(r[:, 2] & 0x1fffff), ((r[:, 2] >> 21) & 0x1fffff), ((r[:, 2] >> 42) & 0x1fffff), ((r[:, 6] >> 32) & 0xffff), ((r[:, 6] >> 48) & 0xffff),
                    ((r[:, 7] >> 16) & 0xffff)], axis=1).astype(np.float64) * 256 * tu     # us: other, lean, no-hit, generic search, match, emission
    print("== %s: %d records, %d calls, wall %.3f ms/call; s_memtime tick = %.4f us" % (tag, n, ncalls, wall_ms, tu))
    plane = 15 - chunk // 256
    span = (end.max() - start.min()) * 0.01
    print("   sum of wave durations %.1f chunk*ms over a span of %.3f ms -> %.0f waves resident on average" % (dur.sum() / 1e3, span / 1e3, dur.sum() / max(span, 1)))
    for b in sorted(set(plane.tolist()), reverse=True):
        m = plane == b
        d = dur[m]
        if d.sum() < 1:
            continue
        print("   bit %2d: %5d waves | us mean %7.1f p50 %7.1f p90 %7.1f max %7.1f | chunk*ms/call %6.1f | ring waits %5.1f/wave, %6.1f us (%4.1f %%) | far %5.1f/wave, %6.1f us (%4.1f %%) | seqs %6.1f generic %5.1f"
              % (b, m.sum(), d.mean(), np.median(d), np.percentile(d, 90), d.max(), d.sum() / 1e3 / ncalls, wait_n[m].mean(), (wait_t[m] * tu).mean(),
                 100 * (wait_t[m] * tu).sum() / d.sum(), far_n[m].mean(), (far_t[m] * tu).mean(), 100 * (far_t[m] * tu).sum() / d.sum(), nseq[m].mean(), ngen[m].mean())
              + (" | us per far fetch %.2f" % ((far_t[m] * tu).sum() / max(1, far_n[m].sum()))))
    for b in (8, 11):
        m = np.nonzero(plane == b)[0]
        if len(m) == 0:
            continue
        top = m[np.argsort(-dur[m])[:6]]
        print("   bit %d, where the time goes (us per wave: other %.0f | lean %.0f | proved-empty batches %.0f | generic search %.0f | generic match %.0f | generic emission %.0f) of %.0f" % ((b,) + tuple(acc[m].mean(axis=0)) + (dur[m].mean(),)))
        i0 = m[np.argmax(dur[m])]
        print("      slowest: chunk %d %.0f us: other %.0f | lean %.0f | proved-empty %.0f | generic search %.0f | match %.0f | emission %.0f" % ((chunk[i0], dur[i0]) + tuple(acc[i0])))
        print("   slowest waves of bit %d: " % b + "; ".join("chunk %d %.0f us, %d lean seqs, %d generic batches, %d far (%.0f us), out %d" % (chunk[i], dur[i], nseq[i], ngen[i], far_n[i], far_t[i] * tu, int(r[i, 7] >> 32)) for i in top))
        # the same chunk numbers' distribution
        cs = sorted(set(chunk[top].tolist()))
        for c in cs[:3]:
            mm = chunk == c
            print("      chunk %d over all calls: us min %.0f mean %.0f max %.0f (n %d)" % (c, dur[mm].min(), dur[mm].mean(), dur[mm].max(), mm.sum()))
    # per kernel (calls of one caller thread share the scratch buffer and follow each other in time): when its waves started
    tagv = (r[:, 6] & 0xffffffff).astype(np.int64)
    rows = []
    for tg in sorted(set(tagv.tolist())):
        idx = np.nonzero(tagv == tg)[0]
        idx = idx[np.argsort(start[idx], kind="stable")]
        # split into kernels: a chunk number seen again starts a new kernel
        seen, cur = set(), []
        kernels = []
        for i in idx:
            c = int(chunk[i])
            if c in seen:
                kernels.append(cur); cur = []; seen = set()
            seen.add(c); cur.append(i)
        if cur:
            kernels.append(cur)
        for k in kernels:
            k = np.array(k)
            if len(k) < 1000:
                continue
            t0 = start[k].min()
            hv = k[(plane[k] >= 11) & (plane[k] <= 12) & (dur[k] > 100)]
            ns = k[plane[k] <= 7]
            rows.append(((end[k].max() - t0) * 0.01, (start[hv].max() - t0) * 0.01 if len(hv) else 0, (end[hv].max() - t0) * 0.01 if len(hv) else 0,
                         (start[ns].min() - t0) * 0.01 if len(ns) else 0, (start[ns].max() - t0) * 0.01 if len(ns) else 0, (end[ns].max() - t0) * 0.01 if len(ns) else 0,
                         np.median(start[hv] - t0) * 0.01 if len(hv) else 0))
    if rows:
        a = np.array(rows)
        print("   per kernel (n %d, us after its first wave): last wave ends %.0f | heavy (bits 12, 11) starts: median %.0f last %.0f, last end %.0f | noise (bits 7..0) starts %.0f .. %.0f, last end %.0f"
              % (len(a), a[:, 0].mean(), a[:, 6].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean(), a[:, 5].mean()))
    buf[0] = 0
    torch.cuda.synchronize()


for _ in range(2):
    call(0)
torch.cuda.synchronize()
buf[0] = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call(0)
torch.cuda.synchronize()
report("one call at a time", (time.perf_counter() - t0) * 1e3 / 3, 3)


def worker(t):
    torch.cuda.set_device(0)
    for _ in range(K):
        call(t)


t0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(M)]
[th.start() for th in ths]
[th.join() for th in ths]
torch.cuda.synchronize()
report("%d calls in flight" % M, (time.perf_counter() - t0) * 1e3 / (M * K), M * K)
