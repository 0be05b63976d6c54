"""Per-step kernel times from a rocprofv3 results .db (kernel-trace): grouped by kernel and grid.

usage: python tools/prof_top.py gpurun_out/prof/x_results.db STEPS_IN_TRACE [min_us]
"""
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    c = sqlite3.connect(db)
    q = ("select substr(name,1,70), grid_x, grid_y, workgroup_x, count(*), avg(duration)/1000.0, "
         "sum(duration)/1000.0 from kernels group by name, grid_x, grid_y, workgroup_x order by 7 desc")
    total = 0.0
    for name, gx, gy, wx, n, avg, tot in c.execute(q):
        total += tot / steps
        if tot / steps >= min_us:
            print(f"{tot / steps:9.1f} us/step  {n / steps:6.1f} x {avg:8.1f} us  grid {gx // wx}x{gy} wg {wx}  {name}")
    print(f"total {total:.1f} us/step")


if __name__ == "__main__":
    main()
