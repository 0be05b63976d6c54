#!/bin/bash
# PMC counters of the window-attention kernels (tools/attn_bench.py); one rocprofv3 pass per counter group
# usage: tools/attn_pmc.sh TAG -> gpurun_out/r04/attn_pmc_TAG.txt
TAG=${1:-a}
OUT=gpurun_out/r04/attn_pmc_$TAG
mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -o t -- python3 tools/attn_bench.py > "$OUT/g$i.log" 2>&1 || echo "group $i failed" 
done
python3 - "$OUT" <<PY > gpurun_out/r04/attn_pmc_$TAG.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][27:50], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in d.items():
        if "winattn" not in k[0]: continue
        print(k, {n: round(sorted(v)[len(v) // 2]) for n, v in c.items()})
PY
cat gpurun_out/r04/attn_pmc_$TAG.txt
