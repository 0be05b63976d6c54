#!/bin/bash
# SQ counters of the plain and the input-transform (XF) forms of the 64 -> 64 @ 256 x 256 convolution and weight gradient
# (tools/xfbench.py --only=64 --reps=1 under rocprofv3, two counter passes; kernel-trace only beside them).
# usage (on the GPU box): bash tools/pmc_xf.sh OUTDIR
set -e
OUT=${1:-gpurun_out/pmc_xf}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/a" -- python3 tools/xfbench.py --only=64 --reps=1 > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/b" -- python3 tools/xfbench.py --only=64 --reps=1 > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(dict)
for sub in ("a", "b"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(s in n for s in ("conv3x3_pp_kernel", "wgrad9_kernel", "bn_relu_apply")):
                continue
            n = n.replace("(anonymous namespace)::", "").replace("void ", "")
            i = min([n.find(s) for s in ("conv3x3_pp_kernel", "wgrad9_kernel", "bn_relu_apply") if n.find(s) >= 0])
            key = n[i:].split(">(")[0][:96]
            rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, sub)].add(r["Dispatch_Id"])
            dur[key][(sub, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# per launch (averages over the launches of tools/xfbench.py --only=64 --reps=1); cycles are summed over the chip's SIMDs / CUs as the counters report them")
for key, c in sorted(rows.items()):
    na, nb = max(len(cnt[(key, "a")]), 1), max(len(cnt[(key, "b")]), 1)
    us = sum(dur[key].values()) / max(len(dur[key]), 1) / 1e3
    g = lambda k, n: c.get(k, 0.0) / n
    wave = g("SQ_WAVE_CYCLES", na)
    print(f"{key}\n   avg {us:6.1f} us  clock {g('GRBM_GUI_ACTIVE', nb) / 8 / us / 1e3:4.2f} GHz | MFMA busy / busy cycles {g('SQ_VALU_MFMA_BUSY_CYCLES', na) / max(g('SQ_BUSY_CYCLES', nb), 1) :5.2f}"
          f" | VALU insts {g('SQ_INSTS_VALU', na) / 1e6:7.2f} M  MFMA insts {g('SQ_INSTS_MFMA', nb) / 1e6:6.2f} M  LDS insts {g('SQ_INSTS_LDS', nb) / 1e6:6.2f} M  SALU {g('SQ_INSTS_SALU', nb) / 1e6:6.2f} M"
          f" | LDS active {g('SQ_LDS_IDX_ACTIVE', nb) / 1e6:7.2f} Mcyc  bank conflicts {g('SQ_LDS_BANK_CONFLICT', na) / 1e6:6.2f} Mcyc"
          f" | wait any / wave cycles {g('SQ_WAIT_ANY', na) / max(wave, 1):4.2f}  waiting on LDS {g('SQ_WAIT_INST_LDS', na) / max(wave, 1):4.2f}")
PY
