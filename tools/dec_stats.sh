#!/bin/bash
# Diagnostic build of the two-wavefront LZ4 decode (sqy_kernels.hip: SQY_DEC_STATS / SQY_DEC_INERT) and its run on a GPU box.
#   here:     tools/dec_stats.sh build            -> ab_libs/dec_stats.so, ab_libs/dec_stats_inert.so  (remove ab_libs/ afterwards: it travels with every lease)
#   GPU box:  tools/dec_stats.sh run [c2|c5]      -> per frame what both waves counted (tools/dec_stats.py), the slowest frames first
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value -DSQY_PRODUCT_BUILD"
if [ "$1" = build ]; then
  mkdir -p ab_libs
  for v in "stats:-DSQY_DEC_STATS" "stats_inert:-DSQY_DEC_STATS -DSQY_DEC_INERT"; do
    tag=${v%%:*}; defs=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS $defs -c sqeazy_amd/csrc/sqy_kernels.hip -o /tmp/dec_$tag.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/dec_$tag.so /tmp/dec_$tag.o sqeazy_amd/lib/sqy_pipeline.o sqeazy_amd/lib/sqy_capi.o sqeazy_amd/lib/sqy_rccl.o -ldl
  done
  ls -la ab_libs
else
  . tools/ab_common.sh
  for tag in dec_stats dec_stats_inert; do
    ab_install $tag
    echo "== $tag"
    timeout -k 10 200 python tools/dec_stats.py ${2:-c2}
  done
fi
