#!/bin/bash
# after the cutoff rule (seed 10901): the default modes over the long soak's seeds again, and fresh ones
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1500 python "$@" ) > $OUT/long4_$name.log 2>&1; echo "== $name rc=$?"; tail -4 $OUT/long4_$name.log | cut -c1-200; }
python tools/dbg_default.py 10901 2>&1 | grep "flags"
run default tools/fuzz_soak_default.py 10000 16000
FUZZ_SPECIAL=1 run default_special tools/fuzz_soak_default.py 10000 10600
