#!/bin/bash
# round 3, end-of-round soaks at HEAD past the pinned seeds (tick sessions are in every single-chunk call of the specialised / flagship paths now)
set -u
OUT=gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/fuzz_soak.py 1500 1620 > $OUT/soak_exact.log 2>&1; tail -4 $OUT/soak_exact.log
FUZZ_SPECIAL=1 timeout 900 python tools/fuzz_soak_cont.py 1500 1600 > $OUT/soak_cont_special.log 2>&1; tail -4 $OUT/soak_cont_special.log
timeout 900 python tools/fuzz_soak_cont.py 1500 1580 > $OUT/soak_cont.log 2>&1; tail -4 $OUT/soak_cont.log
timeout 900 python tools/fuzz_soak_keep.py 1500 1580 > $OUT/soak_keep.log 2>&1; tail -4 $OUT/soak_keep.log
timeout 900 python tools/state_soak.py 1500 1560 > $OUT/soak_state.log 2>&1; tail -4 $OUT/soak_state.log
timeout 900 python tools/fuzz_soak_tick.py 1500 1640 > $OUT/soak_tick.log 2>&1; tail -4 $OUT/soak_tick.log
