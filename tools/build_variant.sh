#!/bin/bash
# usage: tools/build_variant.sh NAME file.hip -DFLAG[=V] ...   -> tmp_ab/lib_NAME.so: the shipped objects with ONE source
# recompiled under extra defines (debugging / measurement variants; tmp_ab/ is git-ignored and travels to the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p $ROOT/tmp_ab /tmp/t
cd $ROOT/unet_zoo_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $SRC -o /tmp/t/variant_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$" | sed 's#^#./#')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tmp_ab/lib_$NAME.so $OBJS /tmp/t/variant_$NAME.o
echo built $ROOT/tmp_ab/lib_$NAME.so
