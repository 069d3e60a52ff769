#!/bin/bash
# round 6, call 9: where does the remainder split lose?  kernel trace of a short guided sampler run, per-launch durations of the sub-range conv launches
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
REPO=$(pwd); O=$REPO/gpurun_out/r06_9; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/tools/op_trace.py --batch 64 --k 4 --steps 2 --out $O/op.csv > $O/trace.log 2>&1 </dev/null )
python3 - <<PY
import csv, glob, collections, re
rows = []
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv3x3_halo4" in n or "splitk_finish" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), re.sub(r"\(.*$", "", n)[:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
rows.sort()
with open("$O/conv_launches.txt", "w") as f:
    prev_end = None
    for st, dur, n, g, w in rows[-160:]:
        gap = "" if prev_end is None else f"gap {(st - prev_end) / 1e3:8.1f} us"
        f.write(f"{dur / 1e3:9.1f} us  grid {g:>8s}  {n}  {gap}\n")
        prev_end = st + dur
PY
rm -rf $O/trace
echo done
