#!/bin/bash
# quick A/B harness for the LZ4 parse loop (GPU box): whole-loop region on the three sample plane files
#   here:     tools/lz4_quick.sh build      (builds tools/lz4_diag_1_9 and tools/lz4_diag_9_1)
#   GPU box:  tools/lz4_quick.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  REGIONS="${REGIONS:-99_99}" tools/lz4_diag_all.sh build
else
  for f in _plane11 _c3plane _c5plane; do
    echo "== $f"
    for r in ${REGIONS:-99_99}; do timeout -k 5 60 tools/lz4_diag_$r tools/$f.bin || exit 1; done
  done
fi
