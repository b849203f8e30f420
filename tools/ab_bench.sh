#!/bin/bash
# A/B of the in-flight bench on one GPU box: tools/ab_bench.sh "<bench flags>" base v1 base v1   (libraries ab_libs/<tag>.so)
flags="$1"; shift
. tools/ab_common.sh
for tag in "$@"; do
    ab_install $tag
    timeout -k 10 200 python bench.py --quick --steps 30 $flags 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$tag@$AB_SHA', d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], 'single', d['single_call']['ms'], 'verified', d.get('verified'), {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" | tee -a gpurun_out/abb.txt
done
