#!/bin/bash
# after the one-cycle-per-sample rule (seed 16340): default and exact modes over fresh seeds, the special kernels, the noise family
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1500 python "$@" ) > $OUT/long6_$name.log 2>&1; echo "== $name rc=$?"; tail -6 $OUT/long6_$name.log | cut -c1-200; }
run default tools/fuzz_soak_default.py 18000 26000
run exact tools/fuzz_soak.py 12000 13000
FUZZ_SPECIAL=1 run default_special tools/fuzz_soak_default.py 11000 11500
run noise tools/fuzz_soak_default.py 3600 5000 noise
