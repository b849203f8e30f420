#!/bin/bash
# bench the bitswap kernel variants (SQY_BSW_MODE: 0 = LDS tiles, n>=1 = register tiles with n (1 -> 8) blocks per CU)
for m in 0 1 4 16; do
  echo "== SQY_BSW_MODE=$m"
  SQY_BSW_MODE=$m timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step single', d['config'].get('single_call_latency_ms'), d['roofline']['kernels_ms_per_step'])
"
done
