#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (graph replays + the eager, event-bracketed profile steps)
# and the dominant kernel family's average duration in each part of the trace beside the bench line's own figure.
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
d = json.loads(open("gpurun_out/prof_default_bench.json").read().strip().splitlines()[-1])
fam = d["roofline"]["kernel"]
# kernel symbol of the family the bench line names (uz_conv_igemm_kernel_name() -> template arguments)
pat = {"conv3x3_pp512_bf16": "%PpCfg<16, 32, 4, 2, 1>, false>%", "conv3x3_pp512x64_bf16": "%PpCfg<16, 32, 8, 1, 3>, false>%",
       "conv3x3_pp256_bf16": "%PpCfg<8, 32, 4, 2, 3>, false>%"}.get(fam)
if pat is None:
    print("no symbol pattern for family", fam)
    raise SystemExit
rows = [r for r in c.execute("select start, duration/1000.0 from kernels where name like ? order by start", (pat,))]
per_step = d["roofline"]["launches_per_step"]
n_eager = 5 * per_step                      # --profile-steps 5, after the timed graph replays
eager, before = rows[-n_eager:], rows[:-n_eager]
replays = before[-5 * per_step:]            # the last five graph replays of the timed region
print("family", fam, "launches in the trace", len(rows), "per step", per_step)
print("last 5 graph replays (the timed region): avg %.2f us" % (sum(r[1] for r in replays) / len(replays)))
print("eager profile steps (what bench.py's events bracket): avg %.2f us" % (sum(r[1] for r in eager) / len(eager)))
print("bench.py roofline.avg_launch_us %.2f (HIP events, same run)  ms_per_step %.3f" % (d["roofline"]["avg_launch_us"], d["ms_per_step"]))
a, b = sum(r[1] for r in replays) / len(replays), d["roofline"]["avg_launch_us"]
print("eager-event figure vs graph replay: %+.1f %%" % ((b / a - 1) * 100))
PY
