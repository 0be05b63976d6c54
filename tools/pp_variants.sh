#!/bin/bash
# Measurement builds of the ping-pong convolution: one ablation library per -D variant (objects of the other files are
# shared with the ablation build), named unet_zoo_amd/libuz_pp_<tag>.so; tools/convbench.py loads one through
# UNET_ZOO_AMD_LIB.   tools/pp_variants.sh tag1:"-DUZ_PP_SKEL=8" tag2:"-DX=1 -DY=2" ...
set -e
cd "$(dirname "$0")/../unet_zoo_amd/csrc"
make -s -j8 ABLATE=1 >/dev/null
OBJS=$(ls obj_ablate/*.o | grep -v uz_conv3x3_pp.o)
for spec in "$@"; do
  tag="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DUZ_ABLATE $defs -c uz_conv3x3_pp.hip -o /tmp/pp_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libuz_pp_$tag.so $OBJS /tmp/pp_$tag.o
  echo "built libuz_pp_$tag.so ($defs)"
done
