#!/bin/bash
# A/B on one GPU box: every library under tools/_ab/ named on the command line is put in place of the installed one in turn and
# tools/lz4_ab.py is run on it (the box's copy of the repo is scratch).   tools/ab_run.sh "<lz4_ab flags>" base v1 base v1
set -e
flags="$1"; shift
mkdir -p gpurun_out
cp sqeazy_amd/lib/libsqeazy_amd.so /tmp/_installed.so
for tag in "$@"; do
    cp tools/_ab/$tag.so sqeazy_amd/lib/libsqeazy_amd.so
    python tools/lz4_ab.py $tag $flags 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab.txt
done
cp /tmp/_installed.so sqeazy_amd/lib/libsqeazy_amd.so
