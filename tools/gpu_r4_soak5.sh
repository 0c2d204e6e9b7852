#!/bin/bash
# after the noisy-cutoff rule (noise-family seeds 2127, 2203, 2360): the noise family and the default modes again
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 1500 python "$@" ) > $OUT/long5_$name.log 2>&1; echo "== $name rc=$?"; tail -6 $OUT/long5_$name.log | cut -c1-200; }
python tools/dbg_default.py noise 2127 2197 2203 2360 2>&1 | grep "flags"
run noise tools/fuzz_soak_default.py 2000 3600 noise
run noise_exact tools/fuzz_soak.py 2000 2400 noise
run default tools/fuzz_soak_default.py 16000 18000
