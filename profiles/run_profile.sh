#!/bin/bash
# Collects the rocprofv3 evidence for one round.  Run ON THE GPU BOX from the repo root:
#   bash profiles/run_profile.sh r01                        (the metric: BASELINE config 3)
#   bash profiles/run_profile.sh r01_p3 "--workload p3"    (any extra bench.py arguments)
# Pass 1: kernel trace + stats (per-kernel durations).  Passes 2..: PMC counters, each in its own run
# (never combined with other trace domains).  Output lands in gpurun_out/prof_<tag>/ and the summaries
# are condensed into profiles/<tag>_*.{csv,json} by profiles/summarize.py.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu ${2:-}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o pmc -- $BENCH > $OUT/pmc_wr.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc -- $BENCH > $OUT/pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc_mem -o pmc -- $BENCH > $OUT/pmc_mem.log 2>&1
cd $ROOT
python profiles/summarize.py $TAG > $OUT/summary.log 2>&1
# the un-profiled bench line of the same configuration (more steps; cpu_baseline only for the headline tag) — AFTER the summary exists:
# bench.py looks the per-launch HBM traffic up in it (roofline.traffic, measured_in_this_run: false)
python bench.py --steps 10 --warmup 3 ${3:---no-cpu} ${2:-} > profiles/${TAG}_bench.json 2> $OUT/bench.err
tail -40 $OUT/summary.log
