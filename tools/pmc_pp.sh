#!/bin/bash
# SQ counters + clock of the direct convolution kernels on unet layers (tools/convbench.py, one UZ_TUNE setting).
# usage (on the GPU box): bash tools/pmc_pp.sh "e2b,d3a" OUTDIR [tune]
set -e
LAYERS=${1:-d3a}; OUT=${2:-gpurun_out/pmc_pp}; TUNE=${3:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/a" -- python3 tools/convbench.py --tune=$TUNE --reps=1 --only=$LAYERS > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/b" -- python3 tools/convbench.py --tune=$TUNE --reps=1 --only=$LAYERS > "$OUT/b.log" 2>&1
python3 tools/pmc_pp_report.py "$OUT"
