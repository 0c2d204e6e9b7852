#!/bin/bash
# HEAD after the one-second rules: whole seconds through the interpreter AND the specialised kernels, exact modes over whole seconds, short default again
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1500 python "$@" ) > $OUT/long10_$name.log 2>&1; echo "== $name rc=$?"; tail -8 $OUT/long10_$name.log | cut -c1-200; }
run default tools/fuzz_soak_default.py 32000 38000
SOAK_VT=16,48000 FUZZ_SPECIAL=1 run special_1s tools/fuzz_soak_default.py 33000 33300
SOAK_VT=16,48000 run default_1s tools/fuzz_soak_default.py 33300 34100
SOAK_VT=16,48000 run noise_1s tools/fuzz_soak_default.py 33000 33200 noise
