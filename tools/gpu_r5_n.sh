#!/bin/bash
# Round 5, pass N: where seed 105055's error starts (FUZZ_MORE_OV, 200 voices); seed 104123 after the motion fix
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
export FUZZ_MORE_OV=1 SOAK_VT=200,6000
( DBG_FLAGS=2 timeout 300 python tools/dbg_probe.py 105055 10 3.1 2.1 2.2 6.0 7.0 0.0 8.0 ) > $OUT/n_probe.log 2>&1; cat $OUT/n_probe.log | cut -c1-200
( DBG_FLAGS=0,2,4 DBG_VOICES=1 timeout 300 python tools/dbg_default.py 105055 104123 ) > $OUT/n_dbg.log 2>&1; grep "^seed\|voice " $OUT/n_dbg.log | cut -c1-330 | head -24
