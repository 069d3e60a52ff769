#!/bin/bash
# Per-(kernel, grid, block) launch statistics of a short bench run: tells WHICH shapes a generic kernel's time goes to.
#   tools/trace_by_grid.sh TAG [bench.py arguments ...]  ->  gpurun_out/grid_TAG.csv
# default arguments: the headline config with 4 DDIM steps; e.g.  tools/trace_by_grid.sh 5 --config 5 --steps 1 --warmup 0  for the RARM decode
set -euo pipefail
TAG=${1:-x}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT="$REPO/gpurun_out/gridtrace_$TAG"
mkdir -p "$OUT"
if [ $# -eq 0 ]; then set -- --steps 1 --warmup 0 --ddim-steps 4; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 "$REPO/bench.py" "$@" --no-cpu-baseline --no-extras > "$OUT/log" 2>&1
python3 - "$OUT" "$REPO/gpurun_out/grid_$TAG.csv" <<'PY'
import csv, glob, collections, re, sys
out, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:70]
        key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""))
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values()) or 1
with open(dst, "w") as f:
    f.write("kernel,gx,gy,gz,wg,lds,launches,avg_us,total_ms,pct\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k])):
        f.write(",".join(['"%s"' % k[0]] + list(k[1:])) + f",{len(acc[k])},{sum(acc[k]) / len(acc[k]) / 1e3:.2f},{sum(acc[k]) / 1e6:.3f},{100 * sum(acc[k]) / tot:.2f}\n")
PY
rm -rf "${OUT:?}/t"
head -70 "$REPO/gpurun_out/grid_$TAG.csv"
