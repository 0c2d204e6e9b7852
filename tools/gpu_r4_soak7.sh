#!/bin/bash
# whole seconds: the default and the exact modes over 48 000 samples of 16 voices (the other soaks render 1 300 - 2 300 samples)
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1700 python "$@" ) > $OUT/long7_$name.log 2>&1; echo "== $name rc=$?"; tail -8 $OUT/long7_$name.log | cut -c1-200; }
SOAK_VT=16,48000 run default_1s tools/fuzz_soak_default.py 30000 31000
SOAK_VT=16,48000 run exact_1s tools/fuzz_soak.py 30000 30400
SOAK_VT=16,48000 run noise_1s tools/fuzz_soak_default.py 30000 30300 noise
