#!/bin/bash
# window-attention kernel durations under UZ_TUNE variants of the ablation build: tools/attn_ablate.sh "0 65536 131072 262144"
for t in $1; do
  echo "== UZ_TUNE=$t"
  UZ_TUNE=$t UNET_ZOO_AMD_LIB=$PWD/unet_zoo_amd/libunetzoo_hip_ablate.so bash tools/attn_trace.sh abl_$t | grep winattn
done
