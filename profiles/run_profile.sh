#!/bin/bash
# Collects the rocprofv3 evidence for one workload.  Run ON THE GPU BOX from the repo root:
#   bash profiles/run_profile.sh r03                        (the metric: BASELINE config 3)
#   bash profiles/run_profile.sh r03_p3 "--workload p3"    (any extra bench.py arguments)
# Pass 1: kernel trace + stats (per-kernel durations).  Passes 2..: PMC counters, each in its own run (never combined with other
# trace domains).  Every pass runs the SAME bench command — 10 timed steps after 3 warm-up steps — and keeps that run's own JSON line
# in its log: profiles/summarize.py summarises the TIMED steps' dispatches only and checks the kernel durations against the step
# time and the HIP-event kernel time of the very run they were traced in (a profiled run clocks lower than an un-profiled one:
# durations are comparable within a run, not across).  Output lands in gpurun_out/prof_<tag>/; the condensed files in profiles/<tag>_*.
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-side-configs ${2:-}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o pmc -- $BENCH > $OUT/pmc_wr.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc -- $BENCH > $OUT/pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc_mem -o pmc -- $BENCH > $OUT/pmc_mem.log 2>&1
cd $ROOT
rm -f profiles/${TAG}_bench.json
python profiles/summarize.py $TAG --no-assert > $OUT/summary.log 2>&1
# the un-profiled bench line of the same configuration on the same box (cpu_baseline only when asked for: third argument " ") — AFTER the
# summary exists: bench.py looks the per-launch HBM traffic up in it (roofline.traffic, measured_in_this_run: false)
python bench.py --steps 20 --warmup 5 --no-side-configs ${3:---no-cpu} ${2:-} > profiles/${TAG}_bench.json 2> $OUT/bench.err
# ... and once more, now with the un-profiled line beside the traced one (the profiler's inflation of the step) and the checks enforced
python profiles/summarize.py $TAG > $OUT/summary.log 2>&1
echo "summarize rc=$?"
tail -60 $OUT/summary.log
