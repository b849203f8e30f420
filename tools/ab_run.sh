#!/bin/bash
# A/B on one GPU box: every library ab_libs/<tag>.so named on the command line is put in place of the installed one in turn and
# tools/lz4_ab.py is run on it.   tools/ab_run.sh "<lz4_ab flags>" base v1 base v1
set -e
flags="$1"; shift
. tools/ab_common.sh
for tag in "$@"; do
    ab_install $tag
    python tools/lz4_ab.py "$tag@$AB_SHA" $flags 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab.txt
done
