#!/bin/bash
# LZ4 source window experiment: 8 KiB ring (6 chunk waves per CU) against a 4 KiB ring (7 waves per CU).
#   here (no GPU):  SQY_EXTRA_HIPCC_FLAGS="-DSQY_LZ4_WIN=4096 -DSQY_LZ4_AHEAD=1280" python3 -m sqeazy_amd.build --force &&
#                   cp sqeazy_amd/lib/libsqeazy_amd.so sqeazy_amd/lib/libsqeazy_amd_win4k.so; python3 -m sqeazy_amd.build --force
#   GPU box:        tools/win_experiment.sh
for lib in "" _win4k; do
  L=$PWD/sqeazy_amd/lib/libsqeazy_amd$lib.so
  [ -f "$L" ] || continue
  echo "== lib$lib"
  for k in 2 4; do SQEAZY_AMD_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --inflight $k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['roofline']['kernels_ms_per_step'])
"; done
  SQEAZY_AMD_LIB=$L timeout -k 10 200 python tools/config_times.py 2>/dev/null | grep -v decode | cut -c1-200
done
