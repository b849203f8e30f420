#!/bin/bash
# headline bench at several numbers of calls in flight
for k in 1 2 3 4 6 8; do
  echo "== inflight $k"
  SQY_BSW_MODE=${SQY_BSW_MODE:-16} timeout -k 10 200 python bench.py --steps 16 --warmup 4 --inflight $k --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'GB/s', d['ms_per_step'], 'ms/step', d['roofline']['kernels_ms_per_step'])
"
done
