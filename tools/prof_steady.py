"""Steady state of a bench.py kernel trace: the last N graph replays of a rocprofv3 results .db, delimited by the AdamW
launches (one per step).  Per (kernel, grid): us per step, launches per step, average duration.
usage: python tools/prof_steady.py x_results.db [N=5] [min_us=0]"""
import sqlite3
import sys
import collections


def short(name):
    if "conv3x3_pp_kernel" in name:
        import re
        m = re.search(r"PpCfg<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>, (true|false), (true|false), (true|false)", name)
        if m:
            return "conv3x3_pp<%sx%s,%sx%s,%s>%s%s%s" % (m.group(1), m.group(2), m.group(3), m.group(4), m.group(5),
                                                         "_bnred" if m.group(6) == "true" else "",
                                                         "_splitk" if m.group(7) == "true" else "", "_xf" if m.group(8) == "true" else "")
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("at::native::"):      # torch glue: keep enough of the functor's name to tell which operation it is
        import re
        m = re.search(r"at::native::(\w+)<.*?at::native::(\w+)", n)
        return ("torch:" + m.group(1) + ":" + m.group(2)) if m else n[:96]
    return n[:64]


def main():
    db = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, grid_x, grid_y, workgroup_x, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "adamw_apply_kernel" in r[0]]
    if len(marks) < n + 1:
        raise SystemExit("not enough steps in the trace")
    lo, hi = marks[-n - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    wall = (rows[marks[-1]][5] - rows[marks[-n - 1]][5]) / 1e3 / n
    acc = collections.OrderedDict()
    for name, gx, gy, wx, s, e in sel:
        k = (short(name), gx // wx, gy, wx)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in acc.values()) / n
    print(f"# steady state: the last {n} graph replays; sum of kernel time {tot:.1f} us/step; wall clock between the AdamW launches {wall:.1f} us/step")
    print("# us per step | launches per step | average us | grid x wg | kernel")
    fam = collections.Counter()
    for k, (cnt, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        fam[k[0].split("<")[0].split("(")[0]] += us / n
        if us / n >= min_us:
            print(f"{us / n:9.1f} {cnt / n:5.1f} x {us / cnt:8.1f} us  grid {k[1]}x{k[2]} wg {k[3]}  {k[0]}")
    print("# by family:")
    for f, us in fam.most_common(16):
        print(f"#   {us:8.1f} us/step  {f}")


if __name__ == "__main__":
    main()
