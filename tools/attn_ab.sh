#!/bin/bash
# same-box A/B of the window-attention kernels: tools/attn_ab.sh LIB_A LIB_B  (kernel durations by rocprofv3, twice each)
for i in 1 2; do
  for lib in "$@"; do
    echo "== $lib (pass $i)"
    UNET_ZOO_AMD_LIB=$PWD/$lib bash tools/attn_trace.sh ab_$(basename $lib .so)_$i | grep winattn
  done
done
