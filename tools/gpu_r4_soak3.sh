#!/bin/bash
# a long soak at the end of round 4 (GPU box): fresh seeds for every pinned property
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1500 python "$@" ) > $OUT/long_$name.log 2>&1; echo "== $name rc=$?"; tail -4 $OUT/long_$name.log | cut -c1-200; }
run default tools/fuzz_soak_default.py 10000 14000
run exact tools/fuzz_soak.py 10000 11500
FUZZ_SPECIAL=1 run default_special tools/fuzz_soak_default.py 10000 10400
FUZZ_SPECIAL=1 run exact_special tools/fuzz_soak.py 10000 10300
run noise tools/fuzz_soak_default.py 2000 2600 noise
run tick tools/fuzz_soak_tick.py 600 760
run keep tools/fuzz_soak_keep.py 700 900
run srk tools/srk_soak.py 700 1000
run state tools/state_soak.py 700 900
