#!/bin/bash
# LDS occupancy sensitivity of the LZ4 kernel: the same build with fewer chunk waves per CU (padding experiment;
# build the variants with SQY_EXTRA_HIPCC_FLAGS=-DSQY_LZ4_PAD=<bytes> and copy the library to libsqeazy_amd_pad<bytes>.so)
for lib in "" _pad1536 _pad4096 _pad8192; do
  echo "== lib$lib"
  L=$PWD/sqeazy_amd/lib/libsqeazy_amd$lib.so
  SQEAZY_AMD_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['roofline']['kernels_ms_per_step'])
"
  SQEAZY_AMD_LIB=$L timeout -k 10 200 python tools/config_times.py 2>/dev/null | grep -v decode | cut -c1-200
done
