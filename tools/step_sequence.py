"""The kernels of ONE steady-state step of a bench.py kernel trace, in launch order: start offset, duration, idle gap in
front, grid, name -- to find the small launches and the gaps between the big ones.
usage: python tools/step_sequence.py x_results.db [which=-2] [min_us=0]   (step = from one AdamW launch to the next)"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_steady import short


def main():
    db = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, grid_x, grid_y, workgroup_x, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "adamw_apply_kernel" in r[0]]
    lo, hi = marks[which - 1] + 1, marks[which] + 1
    t0 = rows[lo][4]
    prev_end = rows[lo - 1][5]
    busy = gaps = 0.0
    small = 0.0
    for name, gx, gy, wx, s, e in rows[lo:hi]:
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        busy += dur
        gaps += max(gap, 0.0)
        if dur < 8.0:
            small += dur + max(gap, 0.0)
        print(f"{(s - t0) / 1e3:9.1f} us  +{dur:7.1f}  gap {gap:6.1f}  grid {gx // wx:5d}x{gy:<3d} wg {wx:4d}  {short(name)}")
        prev_end = max(prev_end, e)
    print(f"# {hi - lo} launches, busy {busy:.1f} us, gaps {gaps:.1f} us, launches under 8 us incl. their gaps {small:.1f} us, "
          f"step {(rows[hi - 1][5] - rows[lo - 1][5]) / 1e3:.1f} us")


main()
