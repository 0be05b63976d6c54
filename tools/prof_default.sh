#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (graph replays + the eager, event-bracketed profile steps)
# and the dominant kernel's average duration in each part of the trace beside the bench line's own figure.
# usage (on the GPU box): bash tools/prof_default.sh  -> gpurun_out/prof_default_bench.json, gpurun_out/prof_default_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_default; mkdir -p gpurun_out/prof_default
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_default -o x -- python3 bench.py --no-cpu-baseline --fp32-steps 0 > gpurun_out/prof_default_bench.json 2> gpurun_out/prof_default/err.log
DB=$(ls gpurun_out/prof_default/*.db gpurun_out/prof_default/*/*.db 2>/dev/null | head -1)
python3 tools/prof_csv.py "$DB" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --fp32-steps 0" > gpurun_out/prof_default_kernel_stats.csv
python3 - <<'PY'
import sqlite3, glob, json
db = glob.glob("gpurun_out/prof_default/*.db") + glob.glob("gpurun_out/prof_default/*/*.db")
c = sqlite3.connect(db[0])
rows = [r for r in c.execute("select start, duration/1000.0 from kernels where name like '%conv3x3_direct_kernelIDF16bLi32ELi128ELb0ELb0%' order by start")]
n = len(rows)
last = rows[-85:]      # 5 eager profile steps x 17 launches
first = rows[:-85]
print("launches", n, "all avg", sum(r[1] for r in rows) / n)
print("last 85 (eager profile steps) avg", sum(r[1] for r in last) / len(last))
print("before (graph replays + capture) avg", sum(r[1] for r in first) / len(first))
d = json.loads(open("gpurun_out/prof_default_bench.json").read().strip().splitlines()[-1])
print("bench roofline avg_launch_us", d["roofline"]["avg_launch_us"], "launches/step", d["roofline"]["launches_per_step"], "ms", d["ms_per_step"])
PY
