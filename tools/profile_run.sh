#!/bin/bash
# GPU box: plain bench + rocprofv3 kernel trace + the two HBM counter passes, all over the default bench command.
# Outputs under gpurun_out/prof/; condense afterwards (here) with tools/summarize_profile.py <tag>.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P=gpurun_out/prof
rm -rf $P; mkdir -p $P      # (delete your local gpurun_out/prof before fetching: gpurun merges, it does not remove stale files)
B="bench.py"
timeout -k 10 500 python3 $B > $P/bench_plain.log 2>&1 || exit 1
tail -1 $P/bench_plain.log | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $B --quick > $P/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_fetch -- python3 $B --quick > $P/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_write -- python3 $B --quick > $P/bench_write.log 2>&1 || exit 1
# wave-level counters of the same command (what the waves wait for): three passes, a handful of counters each
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $P/pmc_sq1 -- python3 $B --quick > $P/bench_sq1.log 2>&1 || echo "pmc_sq1 pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $P/pmc_sq2 -- python3 $B --quick > $P/bench_sq2.log 2>&1 || echo "pmc_sq2 pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $P/pmc_sq3 -- python3 $B --quick > $P/bench_sq3.log 2>&1 || echo "pmc_sq3 pass failed"
sha256sum sqeazy_amd/lib/libsqeazy_amd.so | cut -d" " -f1 > $P/library_sha256.txt
echo profile passes done
