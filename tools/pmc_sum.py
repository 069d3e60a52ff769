#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small CSVs for profiles/.

    pmc_sum.py counters OUT.csv DIR [DIR...]     mean counter value per dispatch per kernel (from *counter_collection.csv)
    pmc_sum.py stats OUT.csv DIR                 per-kernel calls / total / average duration (from *kernel_stats.csv)
Kernel names are shortened to the template head so the files stay readable."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def counters(out, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("kernel,counter,mean_per_launch,launches\n")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                f.write(f'"{k}",{c},{sum(v) / len(v):.0f},{len(v)}\n')


def stats(out, d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    rows.sort(key=lambda r: -r[2])
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for n, c, t, a, p in rows:
            f.write(f'"{n}",{c},{t / 1e6:.3f},{a / 1e3:.2f},{p:.2f}\n')


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3:])
    else:
        stats(sys.argv[2], sys.argv[3])
