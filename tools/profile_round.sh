#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh TAG      ->  gpurun_out/prof_TAG/{stats.csv, sq.csv, fetch.csv, write.csv, bench.json}
# kernel-trace + stats on the default bench command; counters in their own passes (--pmc with --kernel-trace only) on a
# shortened run (4 DDIM steps, full-size database so the kNN scan is the real one).
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
python3 $REPO/tools/pmc_sum.py stats $OUT/stats.csv $OUT/stats
SHORT="--steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/sq1 -- python3 $REPO/bench.py $SHORT > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/sq2 -- python3 $REPO/bench.py $SHORT > $OUT/sq2.log 2>&1
python3 $REPO/tools/pmc_sum.py counters $OUT/sq.csv $OUT/sq1 $OUT/sq2
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- python3 $REPO/bench.py $SHORT > $OUT/fetch.log 2>&1
python3 $REPO/tools/pmc_sum.py counters $OUT/fetch.csv $OUT/fetch
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- python3 $REPO/bench.py $SHORT > $OUT/write.log 2>&1
python3 $REPO/tools/pmc_sum.py counters $OUT/write.csv $OUT/write
# durations of the counter passes' kernels (kernel-trace of the SQ pass) for per-launch cycle/time cross-checks
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:90]
        acc[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$OUT/sq_durations.csv", "w") as f:
    f.write("kernel,launches,avg_us\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k])):
        f.write(f'"{k}",{len(acc[k])},{sum(acc[k]) / len(acc[k]) / 1e3:.2f}\n')
PY
python3 $REPO/tools/pmc_sum.py json $OUT/pmc.json $OUT/sq.csv $OUT/fetch.csv $OUT/write.csv "FETCH_SIZE x 2 (gfx950 wide-read correction) + WRITE_SIZE, KB -> bytes; separate rocprofv3 --pmc passes of bench.py --ddim-steps 4 on this tree (tools/profile_round.sh $TAG); means over the launches of the passes"
rm -rf $OUT/stats $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write
# RARM decode (config #5): bench line + kernel stats at 64 / 256 / 512 / 1024 / 2048 sequences, one SQ pass at the default batch (2048) on a short run (32 tokens)
for b in ${RARM_BATCHES:-64 256 512 1024 2048}; do
  python3 $REPO/bench.py --config 5 --batch $b --db-rows 2000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rarm_bench_b$b.json 2> $OUT/rarm_bench_b$b.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rstats_b$b -- python3 $REPO/bench.py --config 5 --batch $b --db-rows 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/rstats_b$b.log 2>&1
  python3 $REPO/tools/pmc_sum.py stats $OUT/rarm_b${b}_kernel_stats.csv $OUT/rstats_b$b
  rm -rf $OUT/rstats_b$b
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/rsq -- python3 $REPO/bench.py --config 5 --db-rows 2000000 --ddim-steps 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/rsq.log 2>&1
python3 $REPO/tools/pmc_sum.py counters $OUT/rarm_b2048_sq.csv $OUT/rsq
rm -rf $OUT/rsq
# the other single-GPU configurations on the same box
python3 $REPO/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_config2.json 2> $OUT/bench_config2.err
python3 $REPO/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err
# per-op table of one guided UNet forward (tools/op_trace.py)
python3 $REPO/tools/op_trace.py --out $OUT/op_trace.csv > $OUT/op_trace.log 2>&1
ls -la $OUT
