#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_step.sh MODEL TAG [extra bench.py flags]
#   -> gpurun_out/r05/TAG_bench.json (the line of the traced run), TAG_steady.txt (tools/prof_steady.py: per (kernel, grid)
#      of the last 5 replays), TAG_sequence.txt (tools/step_sequence.py: one step in launch order)
set -e
MODEL=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$TAG
rocprofv3 --kernel-trace -d /tmp/pf_$TAG -o run -- python3 $ROOT/bench.py --model $MODEL --steps 20 --warmup 5 --profile-steps 0 \
  --no-cpu-baseline --second-steps 0 --fp32-steps 0 "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
DB=$(find /tmp/pf_$TAG -name "*.db" | head -1)
python3 $ROOT/tools/prof_steady.py $DB 5 > $OUT/${TAG}_steady.txt
python3 $ROOT/tools/step_sequence.py $DB > $OUT/${TAG}_sequence.txt
head -1 $OUT/${TAG}_steady.txt
