#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv: mean per dispatch per kernel."""
import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "halo" not in k and "igemm" not in k and "flash" not in k: continue
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
