"""Timeline of a rocprofv3 --kernel-trace CSV of the bench (several calls in flight): for every instant of the steady state, which
kernels run side by side.  Prints per kernel the mean duration, the share of wall time covered by k concurrent instances of the
transpose / the LZ4 parse, the idle share, and the time a call's chain spends between its kernels.
    python tools/timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict


def short(name):
    for k in ("bitswap1_u16", "lz4_dedupe_key", "lz4_dedupe_verify", "lz4_chunks", "lz4_frame_scan", "lz4_tail_marks", "lz4_frame_gather",
              "lz4_stash_raw", "lz4_fused"):
        if k in name:
            if k == "lz4_chunks" and "ILb0ELb1E" in name:
                return "lz4_chunks_dense"
            return k
    return name[:40]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Thread_Id", "?"))))
    rows.sort()
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
    t_hi = rows[0][0] + (rows[-1][1] - rows[0][0]) * 0.95
    rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    span = rows[-1][1] - rows[0][0]
    print("steady-state window: %.2f ms, %d dispatches" % (span / 1e6, len(rows)))
    dur = defaultdict(list)
    for s, e, k, q, st in rows:
        dur[k].append(e - s)
    ncalls = len(dur.get("lz4_chunks", [])) or 1
    print("calls in window: %d -> %.3f ms per call" % (ncalls, span / 1e6 / ncalls))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("  %-22s n %5d  mean %.3f ms  sum/call %.3f ms" % (k, len(v), sum(v) / len(v) / 1e6, sum(v) / ncalls / 1e6))
    # concurrency histogram
    ev = []
    for s, e, k, q, st in rows:
        ev.append((s, 1, k)); ev.append((e, -1, k))
    ev.sort()
    cur = defaultdict(int)
    hist = defaultdict(float)
    last = ev[0][0]
    for t, d, k in ev:
        key = (cur["bitswap1_u16"], cur["lz4_chunks"] + cur["lz4_chunks_dense"], sum(cur.values()) - cur["bitswap1_u16"] - cur["lz4_chunks"] - cur["lz4_chunks_dense"])
        hist[key] += t - last
        last = t
        cur[k] += d
    tot = sum(hist.values())
    print("share of wall time by (transposes, parses, others) running side by side:")
    for key, v in sorted(hist.items(), key=lambda kv: -kv[1])[:16]:
        print("   T=%d L=%d other=%d : %5.1f %%" % (key[0], key[1], key[2], 100 * v / tot))
    for name, idx in (("transposes", 0), ("parses", 1)):
        agg = defaultdict(float)
        for key, v in hist.items():
            agg[key[idx]] += v
        print("  %s side by side: " % name + "  ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(agg.items())))
    # per-queue chains: gap between consecutive kernels of one queue
    byq = defaultdict(list)
    for s, e, k, q, st in rows:
        byq[q].append((s, e, k))
    print("queues: " + "  ".join("%s: %d" % (q, len(v)) for q, v in byq.items()))
    gaps = defaultdict(list)
    for q, v in byq.items():
        v.sort()
        for a, b in zip(v, v[1:]):
            gaps[(a[2], b[2])].append(b[0] - a[1])
    print("gap between consecutive kernels of one queue (mean us):")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("   %-20s -> %-20s n %5d  mean %8.1f us   sum/call %.3f ms" % (a, b, len(v), sum(v) / len(v) / 1e3, sum(v) / ncalls / 1e6))


if __name__ == "__main__":
    main()
