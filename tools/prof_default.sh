#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (graph replays + the eager, event-bracketed profile steps)
# and the dominant kernel family's average duration in each part of the trace beside the bench line's own figure.
# usage (on the GPU box): bash tools/prof_default.sh  -> gpurun_out/prof_default_bench.json, gpurun_out/prof_default_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_default; mkdir -p gpurun_out/prof_default
# the same command WITHOUT the profiler first: its per-launch figure is the one the bench line carries
python3 bench.py --no-cpu-baseline --fp32-steps 0 --second-steps 0 > gpurun_out/prof_default_bench_plain.json 2> gpurun_out/prof_default/plain_err.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_default -o x -- python3 bench.py --no-cpu-baseline --fp32-steps 0 --second-steps 0 > gpurun_out/prof_default_bench.json 2> gpurun_out/prof_default/err.log
DB=$(ls gpurun_out/prof_default/*.db gpurun_out/prof_default/*/*.db 2>/dev/null | head -1)
python3 tools/prof_csv.py "$DB" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --fp32-steps 0 --second-steps 0" > gpurun_out/prof_default_kernel_stats.csv
python3 - <<'PY'
import sqlite3, glob, json
db = glob.glob("gpurun_out/prof_default/*.db") + glob.glob("gpurun_out/prof_default/*/*.db")
c = sqlite3.connect(db[0])
d = json.loads(open("gpurun_out/prof_default_bench.json").read().strip().splitlines()[-1])
# the kernel the un-profiled bench line names (under the profiler the two-kernel operations -- weight gradient + slab
# reduction inside one event bracket -- read longer and can take over the top place; the comparison below is about the
# convolution either way).  A kernel = one ping-pong configuration with both of its epilogue variants, as bench.py groups them.
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py")
bm = importlib.util.module_from_spec(spec); spec.loader.exec_module(bm)
plain = json.loads(open("gpurun_out/prof_default_bench_plain.json").read().strip().splitlines()[-1])
fam = plain["roofline"]["kernel"]             # the dominant kernel family of the un-profiled line, by the library's own name
pats = bm.PMC_KEYS[fam]                       # ... and the kernel symbols that belong to it (bench.py's table)
rows = []
for alt in pats:                              # a substring, or a tuple of substrings that must all occur
    subs = (alt,) if isinstance(alt, str) else tuple(alt)
    rows += [(r[0], r[1]) for r in c.execute("select start, duration/1000.0, name from kernels where name like ? order by start",
                                             ("%" + subs[0] + "%",)) if all(x in r[2] for x in subs)]
rows.sort()
marks = [r[0] for r in c.execute("select start from kernels where name like '%adamw_apply_kernel%' order by start")]
per_step = len([r for r in rows if marks[-2] < r[0] < marks[-1]])      # launches inside one graph replay
k = d["kernel_ms_per_step"]
ev_us = (k[fam] + k.get(fam + "_bnred", 0.0)) * 1e3 / per_step            # bench.py's HIP-event figure, same process
n_eager = 5 * per_step                      # --profile-steps 5, after the timed graph replays
eager = rows[-n_eager:]
replays = [r for r in rows if marks[-6] < r[0] < marks[-1]]   # the last five graph replays of the timed region
print("kernel", fam, "(with its epilogue variants) launches in the trace", len(rows), "per step", per_step)
print("last 5 graph replays (the timed region): avg %.2f us" % (sum(r[1] for r in replays) / len(replays)))
print("eager profile steps in the trace: avg %.2f us" % (sum(r[1] for r in eager) / len(eager)))
print("bench.py per-launch figure from its own HIP events, same run: %.2f us   ms_per_step %.3f (under the profiler)" % (ev_us, d["ms_per_step"]))
a = sum(r[1] for r in replays) / len(replays)
print("eager-event figure vs graph replay, both under the profiler: %+.1f %%" % ((ev_us / a - 1) * 100))
pl = json.loads(open("gpurun_out/prof_default_bench_plain.json").read().strip().splitlines()[-1])
r = pl["roofline"]
print("the same command without the profiler, same box: roofline.kernel %s avg_launch_us %.2f frac %.4f ms_per_step %.3f" % (r["kernel"], r["avg_launch_us"], r["frac"], pl["ms_per_step"]))
print("un-profiled eager-event figure vs the graph replays of the trace: %+.1f %%" % ((r["avg_launch_us"] / a - 1) * 100))
PY
