#!/bin/bash
# bench.py lines of the other model configurations (one JSON line each) -> gpurun_out/bench_TAG_<model>.json
# usage (on the GPU box): bash tools/bench_models.sh TAG
TAG=${1:-r02}
run() { name=$1; shift; python bench.py --no-cpu-baseline --fp32-steps 0 --second-steps 0 "$@" > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err; python3 - gpurun_out/bench_${TAG}_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:24s} {d['ms_per_step']:8.3f} ms/step {d['value']:9.1f} {d['unit']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run u2net_512 --model u2net --size 512 --batch 8
run attention_unet_512 --model attention_unet --size 512
run nested_unet_256 --model nested_unet
run resunet_256 --model resunet
run swin_unet_v2_256 --model swin_unet_v2 --size 256
run swin_unet_v2_224 --model swin_unet_v2 --size 224 --batch 32
run missformer_512 --model missformer --size 512 --batch 8
run transatt_unet_256 --model transatt_unet
run unet_transformer_256 --model unet_transformer
run multiresunet_256 --model multiresunet
run uctransnet_256 --model uctransnet
