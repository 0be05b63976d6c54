"""Kernel summary (name, calls, total_ms, avg_us, pct) of a rocprofv3 --kernel-trace results .db as CSV.

usage: python tools/prof_csv.py gpurun_out/prof/x_results.db "<command line that was profiled>" > profiles/xx.csv
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3 from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# {sys.argv[2] if len(sys.argv) > 2 else 'rocprofv3 --kernel-trace --stats'}")
    print("# name, calls, total_ms, avg_us, pct")
    for name, n, tot, avg in rows:
        name = name.replace('"', "'")
        print(f'"{name}",{n},{tot:.3f},{avg:.2f},{100 * tot / total:.2f}')


if __name__ == "__main__":
    main()
