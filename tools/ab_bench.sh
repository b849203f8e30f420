#!/bin/bash
# A/B of the in-flight bench on one GPU box: tools/ab_bench.sh "<bench flags>" base v1 base v1   (libraries under tools/_ab/)
flags="$1"; shift
mkdir -p gpurun_out
cp sqeazy_amd/lib/libsqeazy_amd.so /tmp/_installed.so
for tag in "$@"; do
    cp tools/_ab/$tag.so sqeazy_amd/lib/libsqeazy_amd.so
    timeout -k 10 200 python bench.py --quick --steps 30 $flags 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$tag', d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], 'single', d['single_call']['ms'], 'verified', d.get('verified'), {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" | tee -a gpurun_out/abb.txt
done
cp /tmp/_installed.so sqeazy_amd/lib/libsqeazy_amd.so
