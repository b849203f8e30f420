#!/bin/bash
# builds tools/lz4_diag_<A>_<B> for the usual regions (here, no GPU needed) or runs them all on a chunk file (GPU box)
#   tools/lz4_diag_all.sh build        |   tools/lz4_diag_all.sh run tools/_plane11.bin
REGIONS=${REGIONS:-"0_1 1_2 2_3 3_4 4_5 5_6 6_7 7_8 8_9 9_1 1_9"}
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for r in $REGIONS; do
    a=${r%_*}; b=${r#*_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DSQY_DIAG_A=$a -DSQY_DIAG_B=$b -I sqeazy_amd/csrc tools/lz4_diag.hip -o tools/lz4_diag_$r &
    if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
  done
  wait
else
  for r in $REGIONS; do timeout -k 5 60 tools/lz4_diag_$r "$2" || exit 1; done
fi
