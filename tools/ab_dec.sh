#!/bin/bash
# A/B of the DECODE kernels on one GPU box: every library under tools/_ab/ named on the command line is put in place of the installed
# one in turn and tools/config_times.py is run on it (its decode lines are kept).   tools/ab_dec.sh base v1 base v1
set -e
mkdir -p gpurun_out
cp sqeazy_amd/lib/libsqeazy_amd.so /tmp/_installed.so
for tag in "$@"; do
    cp tools/_ab/$tag.so sqeazy_amd/lib/libsqeazy_amd.so
    python tools/config_times.py 2>&1 | grep "decode rc" | sed "s/^/$tag /" | tee -a gpurun_out/ab_dec.txt
done
cp /tmp/_installed.so sqeazy_amd/lib/libsqeazy_amd.so
