#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/gridtrace_5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:70]
        key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
with open("$REPO/gpurun_out/grid_5.csv", "w") as f:
    f.write("kernel,gx,gy,gz,wg,launches,avg_us,total_ms,pct\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k])):
        f.write(",".join(['"%s"' % k[0]] + list(k[1:])) + f",{len(acc[k])},{sum(acc[k]) / len(acc[k]) / 1e3:.2f},{sum(acc[k]) / 1e6:.3f},{100*sum(acc[k])/tot:.2f}\n")
PY
rm -rf $OUT/t
head -40 $REPO/gpurun_out/grid_5.csv
