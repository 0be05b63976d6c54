#!/bin/bash
# One box, one call: bench.py's line (ms/step, sensors, MFMA clock probe) and the steady-state per-kernel table of the
# same graphed step (rocprofv3 --kernel-trace) -> gpurun_out/box_<tag>_{bench.json,steady.txt}
TAG=${1:-$(date +%H%M%S)}
MODEL=${2:-unet}      # bench.py --model
EXTRA=${3:-}          # further bench.py arguments, e.g. "--size 224 --batch 32"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --model $MODEL $EXTRA --no-cpu-baseline --fp32-steps 0 --second-steps 0 --profile-steps 0 > gpurun_out/box_${TAG}_bench.json 2>/dev/null
OUT=gpurun_out/prof_box_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace -d "$OUT" -o x -- python3 bench.py --model $MODEL $EXTRA --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --fp32-steps 0 --second-steps 0 > "$OUT/bench.json" 2> "$OUT/err.log"
DB=$(ls "$OUT"/*.db "$OUT"/*/*.db 2>/dev/null | head -1)
python3 tools/prof_steady.py "$DB" 5 30 > gpurun_out/box_${TAG}_steady.txt
rm -rf "$OUT"
python3 - "$TAG" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/box_{t}_bench.json").read().strip().splitlines()[-1])
print("box", t, "ms_per_step", d["ms_per_step"], "mfma_clock", d["device"]["mfma_clock"], "power_after", d["device"]["after_timed_steps"].get("power_w"), "pci", d["device"]["after_timed_steps"].get("pci"))
PY
grep "^#" gpurun_out/box_${TAG}_steady.txt | head -12
