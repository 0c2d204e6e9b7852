#!/bin/bash
# after the degree-10 polynomial for sweeping CVs (seeds 31051, 28336): both seeds, whole seconds, the short soaks, config 4 / P3
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1700 python "$@" ) > $OUT/long9_$name.log 2>&1; echo "== $name rc=$?"; tail -8 $OUT/long9_$name.log | cut -c1-200; }
SOAK_VT=16,48000 python tools/dbg_default.py 31051 2>&1 | grep flags
python tools/dbg_default.py 28336 2>&1 | grep flags
for w in p3 cfg4; do bash tools/gpu_r4.sh bench i_$w --workload $w --steps 10 --warmup 2; done
run default tools/fuzz_soak_default.py 26000 32000
SOAK_VT=16,48000 run default_1s tools/fuzz_soak_default.py 32000 33000
