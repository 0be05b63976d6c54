#!/bin/bash
# Two builds of the kernel library on ONE box, alternating processes (boxes differ by several per cent on the MFMA kernels):
#   bash tools/ab_lib.sh OLD.so NEW.so [model ...]      (paths relative to the repo root; default models: unet swin_unet_v2)
# prints ms_per_step of `bench.py --model M` (graph replays, no baselines) for OLD, NEW, OLD, NEW.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OLD=$1; NEW=$2; shift 2
MODELS=${@:-unet swin_unet_v2}
for m in $MODELS; do
  for rep in 1 2; do
    for lib in $OLD $NEW; do
      UNET_ZOO_AMD_LIB=$ROOT/$lib python3 $ROOT/bench.py --model $m --steps 40 --warmup 8 --profile-steps 0 --no-cpu-baseline \
        --second-steps 0 --fp32-steps 0 --no-torch-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m', '$lib', 'ms_per_step', l['ms_per_step'])"
    done
  done
done
