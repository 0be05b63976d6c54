"""Summarise the counter CSVs written by tools/pmc_pp.sh: per (kernel, grid) averages, clock from GRBM_GUI_ACTIVE."""
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, c in acc.items():
        cnt = max(len(n[k]), 1)
        avg = sum(dur[k].values()) / cnt / 1e3
        line = {a: round(b / cnt) for a, b in c.items()}
        extra = ""
        if "GRBM_GUI_ACTIVE" in line and avg > 0:
            extra = f" clock {line['GRBM_GUI_ACTIVE'] / 8 / avg / 1e3:.2f} GHz"
        print(sub, k, "launches", cnt, f"avg {avg:.1f} us", line, extra)
