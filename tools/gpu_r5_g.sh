#!/bin/bash
# Round 5, pass G: the soaks on the final library, 16 processes, every render read back twice (is a difference the copy's or the device's?)
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
W=16
( SOAK_RETRY=2 SOAK_VT=16,48000 SOAK_TIMEOUT=1500 timeout 1600 python tools/soak_par.py g_1s 50000 60000 $W ) > $OUT/g_soak_1s.log 2>&1; echo "== soak 1s rc=$?"; tail -6 $OUT/g_soak_1s.log | cut -c1-230
( SOAK_RETRY=2 SOAK_VT=200,6000 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py g_v200 60000 70000 $W ) > $OUT/g_soak_v200.log 2>&1; echo "== soak 200x6000 rc=$?"; tail -6 $OUT/g_soak_v200.log | cut -c1-230
grep -h "^RETRY\|^READBACK" $OUT/soak_g_*_w*.log | cut -c1-360 | head -40
