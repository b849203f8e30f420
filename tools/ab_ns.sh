#!/bin/bash
cp sqeazy_amd/lib/libsqeazy_amd.so /tmp/_installed.so
for tag in "$@"; do cp tools/_ab/$tag.so sqeazy_amd/lib/libsqeazy_amd.so; python tools/ns_slab.py $tag 2>&1 | grep -v amdgpu.ids; done
cp /tmp/_installed.so sqeazy_amd/lib/libsqeazy_amd.so
