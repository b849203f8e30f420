#!/usr/bin/env python3
"""Condenses the rocprofv3 output of a GPU run (gpurun_out/prof/...) into profiles/ (tracked).

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace     -- python3 bench.py ...
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py ...

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

SHORT = {"lz4_chunks_kernel<false, false>": "lz4_chunks", "lz4_chunks_kernel<false, true>": "lz4_chunks_dense",
         "lz4_chunks_kernel<true, false>": "lz4_linked", "bitswap1_u16_regs": "bitswap1_u16",
         "bitswap1_u16_regs<true>": "bitswap1_u16", "bitswap1_u16_regs<false>": "bitswap1_u16",
         "lz4_tail_marks_kernel": "lz4_tail_marks", "lz4_stash_raw_kernel": "lz4_stash_raw",
         "lz4_frame_gather_kernel": "lz4_frame_gather", "lz4_frame_scan_kernel": "lz4_frame_scan",
         "bitswap1_u16_generic": "bitswap1_u16_generic", "lz4_dedupe_key_kernel": "lz4_dedupe_key",
         "lz4_dedupe_verify_kernel": "lz4_dedupe_verify"}


def newest(pattern):
    """the files of the most recent run only: gpurun merges a run's output into gpurun_out/ without removing what an earlier run left"""
    files = glob.glob(pattern)
    if not files:
        return []
    files.sort(key=os.path.getmtime)
    return [files[-1]]


def is_ours(name):
    return name.startswith("sqy::") or name.startswith("void sqy::")


def short(name):
    base = name.split("(")[0]
    if base.startswith("void "):
        base = base[5:]
    if base.startswith("sqy::"):
        base = base[5:]
    # round 4: the parse kernel has a third template parameter (liblz4 acceleration above 1)
    if base.startswith("lz4_chunks_kernel<"):
        args = [a.strip() for a in base[len("lz4_chunks_kernel<"):-1].split(",")]
        if args[0] == "true":
            return "lz4_linked"
        return "lz4_chunks_dense" if args[1] == "true" else ("lz4_chunks_accel" if len(args) > 2 and args[2] == "true" else "lz4_chunks")
    for k, v in (("lz4_dedupe_clear_kernel", "lz4_dedupe_clear"), ("lz4_inplace_finish_kernel", "lz4_inplace_finish"),
                 ("lz4_inplace_tail_fused_kernel", "lz4_inplace_tail_fused")):
        if base.startswith(k):
            return v
    return SHORT.get(base, base)


rows = []
for f in newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if is_ours(r["Name"]):
            rows.append(r)
with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --quick  (defaults: 30 steps, 5 warm-up, 4 calls in flight, GPU_MAX_HW_QUEUES=8)   (sqy:: kernels only)\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("pmc_fetch", "pmc_write"):
    for f in newest(os.path.join(src, name, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
with open(os.path.join(dst, "%s_pmc_hbm.csv" % tag), "w") as f:
    f.write("# separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over bench.py; KiB per launch, mean over launches\n")
    f.write("# hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE counts 128-B requests at 64 B)\n")
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_per_launch"])
    for k, c in sorted(pmc.items()):
        fe = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
        wr = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
        hbm = int((2 * fe + wr) * 1024)
        traffic[k] = hbm
        w.writerow([k, len(c["FETCH_SIZE"]), "%.1f" % fe, "%.1f" % wr, hbm])
sha_file = os.path.join(src, "library_sha256.txt")
sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
# bench.py reports `roofline.traffic` from this file only when it was collected on the very library build it is running
json.dump({"library_sha256": sha, "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --quick",
           "hbm_bytes": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, mean over launches", "traffic": traffic},
          open(os.path.join(dst, "%s_pmc_hbm.json" % tag), "w"), indent=1, sort_keys=True)
# wave-level counters (what the waves wait for), summed over the XCDs' instances of a launch, mean over launches
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    for f in newest(os.path.join(src, name, "*", "*_counter_collection.csv")):
        per_launch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]):
                per_launch[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _, c), v in per_launch.items():
            sq[k][c].append(v)
if sq:
    names = sorted({c for k in sq for c in sq[k]})
    with open(os.path.join(dst, "%s_pmc_sq.csv" % tag), "w") as f:
        f.write("# rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 bench.py --quick (three passes); per launch, mean over launches\n")
        w = csv.writer(f)
        w.writerow(["kernel"] + names)
        for k in sorted(sq):
            w.writerow([k] + ["%.0f" % (sum(sq[k][c]) / len(sq[k][c])) if sq[k][c] else "" for c in names])
    print(open(os.path.join(dst, "%s_pmc_sq.csv" % tag)).read())
for n in ("bench_plain.log", "bench_trace.log"):
    p = os.path.join(src, n)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, "%s_%s" % (tag, n.replace(".log", ".json"))), "w").write(lines[-1])
print(open(os.path.join(dst, "%s_kernel_stats.csv" % tag)).read())
print(open(os.path.join(dst, "%s_pmc_hbm.csv" % tag)).read())
