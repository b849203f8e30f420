#!/bin/bash
# A/B of the DECODE kernels on one GPU box: every library ab_libs/<tag>.so named on the command line is put in place of the installed
# one in turn and tools/config_times.py is run on it (its decode lines are kept).   tools/ab_dec.sh base v1 base v1
set -e
. tools/ab_common.sh
for tag in "$@"; do
    ab_install $tag
    python tools/config_times.py 2>&1 | grep "decode rc" | sed "s/^/$tag@$AB_SHA /" | tee -a gpurun_out/ab_dec.txt
done
