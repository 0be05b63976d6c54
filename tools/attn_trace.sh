#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of the window-attention kernels at the swin_unet_v2 stage sizes (tools/attn_bench.py)
# usage: tools/attn_trace.sh TAG   -> gpurun_out/r04/attn_TAG.txt
TAG=${1:-a}
OUT=gpurun_out/r04/attn_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 tools/attn_bench.py > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<PY > gpurun_out/r04/attn_$TAG.txt
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    g = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    d[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    if len(v) >= 10:
        print("%-50s grid %-26s n=%3d median %6.1f us  min %6.1f" % (k[0], "x".join(k[1]), len(v), v[len(v) // 2], v[0]))
PY
cat gpurun_out/r04/attn_$TAG.txt
