#!/bin/bash
# after the held-CV rule (seed 30111): whole seconds again, the short default / exact soaks, P3 and config 4
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1700 python "$@" ) > $OUT/long8_$name.log 2>&1; echo "== $name rc=$?"; tail -8 $OUT/long8_$name.log | cut -c1-200; }
SOAK_VT=16,48000 python tools/dbg_default.py 30111 2>&1 | grep flags
for w in p3 cfg4 cfg3_poly; do bash tools/gpu_r4.sh bench h_$w --workload $w --steps 10 --warmup 2; done
run default tools/fuzz_soak_default.py 26000 30000
SOAK_VT=16,48000 run default_1s tools/fuzz_soak_default.py 31000 32000
FUZZ_SPECIAL=1 run default_special tools/fuzz_soak_default.py 11500 11800
