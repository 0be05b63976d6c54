#!/bin/bash
# HBM bytes per kernel launch of the bench step from the PMC counters: two SEPARATE passes (FETCH_SIZE, WRITE_SIZE),
# kernel-trace only beside them, aggregated by tools/pmc_traffic.py.
# usage (on the GPU box): bash tools/pmc_traffic.sh OUT.json [bench.py arguments ...]
set -e
OUTJ=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
ARGS="bench.py --graph off --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --fp32-steps 0 $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 $ARGS > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 $ARGS > gpurun_out/pmc_w.log 2>&1
F=$(ls gpurun_out/pmc_f/*counter_collection.csv gpurun_out/pmc_f/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/pmc_w/*counter_collection.csv gpurun_out/pmc_w/*/*counter_collection.csv 2>/dev/null | head -1)
python3 tools/pmc_traffic.py "$F" "$W" "$OUTJ"
