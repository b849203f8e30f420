#!/bin/bash
# A/B of the north_star slab run: tools/ab_ns.sh base v1   (libraries ab_libs/<tag>.so)
. tools/ab_common.sh
for tag in "$@"; do ab_install $tag; python tools/ns_slab.py "$tag@$AB_SHA" 2>&1 | grep -v amdgpu.ids; done
