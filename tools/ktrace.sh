#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of one tools/kbench.py layer under UZ_TUNE variants
# usage (on the GPU box): bash tools/ktrace.sh WHAT LAYER "TUNE1 TUNE2 ..." OUTDIR
WHAT=${1:-conv}; LAYER=${2:-e3b}; TUNES=${3:-0}; OUT=${4:-gpurun_out/ktrace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for t in $TUNES; do
  rm -rf "$OUT/t$t"; KB_ONLY=$LAYER rocprofv3 --kernel-trace --output-format csv -d "$OUT/t$t" -- python3 tools/kbench.py $WHAT --tune=$t > "$OUT/t$t.log" 2>&1
  python3 - "$OUT/t$t" "$t" "$LAYER" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if any(s in k for s in ("conv3x3", "wgrad", "gemm_dma")):
        v = sorted(v)
        print(f"{sys.argv[3]} tune {sys.argv[2]:>6s}  {k:64s} n={len(v):3d}  min {v[0]:7.1f}  median {v[len(v)//2]:7.1f} us")
PY
done
