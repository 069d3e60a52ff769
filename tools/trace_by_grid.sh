#!/bin/bash
# Per-(kernel, grid, block) launch statistics of a short bench run: tells WHICH shapes a generic kernel's time goes to.
#   tools/trace_by_grid.sh TAG  ->  gpurun_out/grid_TAG.csv
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/gridtrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --no-extras > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:70]
        key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""))
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$REPO/gpurun_out/grid_$TAG.csv", "w") as f:
    f.write("kernel,gx,gy,gz,wg,lds,launches,avg_us,total_ms\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k])):
        f.write(",".join(['"%s"' % k[0]] + list(k[1:])) + f",{len(acc[k])},{sum(acc[k]) / len(acc[k]) / 1e3:.2f},{sum(acc[k]) / 1e6:.3f}\n")
PY
rm -rf $OUT/t
head -70 $REPO/gpurun_out/grid_$TAG.csv
