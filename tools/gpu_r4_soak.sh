#!/bin/bash
# round 4's soaks past the pinned seeds (GPU box, repo root): default modes (1e-5 band) and exact modes (bit for bit), both patch families
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
A=${1:-2000}; Bn=${2:-4000}
python tools/fuzz_soak_default.py $A $Bn > $OUT/soak_default.log 2>&1; tail -6 $OUT/soak_default.log
python tools/fuzz_soak_default.py 1000 1400 noise > $OUT/soak_default_noise.log 2>&1; tail -4 $OUT/soak_default_noise.log
FUZZ_SPECIAL=1 python tools/fuzz_soak_default.py $A $((A+300)) > $OUT/soak_default_special.log 2>&1; tail -4 $OUT/soak_default_special.log
python tools/fuzz_soak.py $A $((A+1000)) > $OUT/soak_exact.log 2>&1; tail -5 $OUT/soak_exact.log
