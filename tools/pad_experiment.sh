#!/bin/bash
# LDS occupancy sensitivity of the LZ4 kernel: the same build with fewer chunk waves per CU.
#   here (no GPU):  for p in 6144 14336; do SQY_EXTRA_HIPCC_FLAGS=-DSQY_LZ4_PAD=$p python3 -m sqeazy_amd.build --force &&
#                   cp sqeazy_amd/lib/libsqeazy_amd.so sqeazy_amd/lib/libsqeazy_amd_pad$p.so; done; python3 -m sqeazy_amd.build --force
#   GPU box:        tools/pad_experiment.sh          (26 KiB -> 6 waves/CU; +6 KiB -> 5; +14 KiB -> 4)
for lib in "" _pad6144 _pad14336; do
  L=$PWD/sqeazy_amd/lib/libsqeazy_amd$lib.so
  [ -f "$L" ] || continue
  echo "== lib$lib"
  SQEAZY_AMD_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['roofline']['kernels_ms_per_step'])
"
  SQEAZY_AMD_LIB=$L timeout -k 10 200 python tools/config_times.py 2>/dev/null | grep -v decode | cut -c1-200
done
