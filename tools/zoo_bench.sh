#!/bin/bash
# usage (GPU box, repo root): tools/zoo_bench.sh "model:size:batch[:nobase] ..."   -> gpurun_out/r05/zoo_<model>_<size>.json
# one bench.py line per model with --torch-baseline (the same step through stock PyTorch-ROCm ops on the same GPU);
# :nobase skips that (the 512 x 512 configurations: MIOpen's search + fp32 steps of stock torch take > 7 minutes there)
mkdir -p gpurun_out/r05
for spec in $1; do
  IFS=: read m s b nb <<< "$spec"
  tb="--torch-baseline"; [ -n "$nb" ] && tb=""
  timeout -k 10 420 python3 bench.py --model $m --size $s --batch $b --steps 20 --warmup 5 --no-cpu-baseline --second-steps 0 \
    --fp32-steps 0 $tb > gpurun_out/r05/zoo_${m}_${s}.json 2> gpurun_out/r05/zoo_${m}_${s}.err
  python3 - gpurun_out/r05/zoo_${m}_${s}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d.get("torch_rocm_baseline", {})
    print("%-18s %8.1f img/s %8.3f ms | stock torch fp32 %s  bf16+cl %s" % (
        d["config"]["workload"].split()[0], d["value"], d["ms_per_step"],
        t.get("as_reference_fp32_nchw", {}).get("ms_per_step"), t.get("bf16_autocast_channels_last", {}).get("ms_per_step")), flush=True)
except Exception as e:      # noqa
    print(sys.argv[1], "failed:", repr(e)[:200], flush=True)
PY
done
