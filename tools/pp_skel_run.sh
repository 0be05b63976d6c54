#!/bin/bash
# run convbench with each measurement library of tools/pp_variants.sh; args: "tag1 tag2 ..." then convbench args
tags="$1"; shift
for t in $tags; do
  echo "== $t"
  UNET_ZOO_AMD_LIB=$PWD/unet_zoo_amd/libuz_pp_$t.so timeout -k 10 200 python tools/convbench.py --tune=0 --reps=2 "$@" 2>&1 | grep -v amdgpu.ids
done
