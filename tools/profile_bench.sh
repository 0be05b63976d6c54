#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration, summarised into the files profiles/ keeps.
# usage (on the GPU box): bash tools/profile_bench.sh TAG [bench.py arguments ...]
#   -> gpurun_out/TAG_kernel_stats.csv, gpurun_out/TAG_kernels_by_grid.txt, gpurun_out/TAG_bench.json
set -e
TAG=$1; shift
STEPS=20; WARM=3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps $STEPS --warmup $WARM --profile-steps 0 --no-cpu-baseline --fp32-steps 0 --second-steps 0 $*"
rocprofv3 --kernel-trace --stats -d "$OUT" -o x -- $CMD > gpurun_out/${TAG}_bench.json 2> "$OUT/err.log"
DB=$(ls "$OUT"/*.db "$OUT"/*/*.db 2>/dev/null | head -1)
python3 tools/prof_csv.py "$DB" "rocprofv3 --kernel-trace --stats -- $CMD" > gpurun_out/${TAG}_kernel_stats.csv
python3 tools/prof_top.py "$DB" $((STEPS + WARM)) 8 > gpurun_out/${TAG}_kernels_by_grid.txt
head -5 gpurun_out/${TAG}_kernel_stats.csv
