#!/bin/bash
# SQ counters of the direct convolution on one UNet layer (tools/kbench.py): where do the wave cycles go?
# usage (on the GPU box): bash tools/pmc_conv.sh LAYER OUTDIR
set -e
LAYER=${1:-e3b}; OUT=${2:-gpurun_out/pmc_conv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
KB_ONLY=$LAYER rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/a" -- python3 tools/kbench.py ${WHAT:-conv} > "$OUT/a.log" 2>&1
KB_ONLY=$LAYER rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/b" -- python3 tools/kbench.py ${WHAT:-conv} > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    files = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE"): n[k] += 1
    for k, c in acc.items():
        if any(t in k for t in ("conv3x3", "wgrad3x3", "gemm_dma")):
            print(sub, k, "launches", n[k], {a: round(b / max(n[k], 1)) for a, b in c.items()})
PY
