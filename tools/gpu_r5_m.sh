#!/bin/bash
# Round 5, pass M: shapes the bound has not seen — per-voice overrides of MORE parameters and of initial state (FUZZ_MORE_OV: resonance, amounts,
# envelope times, phases per voice: the analysis works on their ranges), and the specialised kernels at 200 voices x 6 000 samples
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( FUZZ_MORE_OV=1 SOAK_VT=16,48000 SOAK_TIMEOUT=800 timeout 900 python tools/soak_par.py m_more_ov_1s 100000 104000 16 ) > $OUT/m_more.log 2>&1; echo "== more overrides rc=$?"; tail -6 $OUT/m_more.log | cut -c1-230
( FUZZ_MORE_OV=1 SOAK_VT=200,6000 SOAK_TIMEOUT=400 timeout 500 python tools/soak_par.py m_more_ov_v200 104000 108000 16 ) > $OUT/m_more2.log 2>&1; echo "== more overrides, 200 voices rc=$?"; tail -6 $OUT/m_more2.log | cut -c1-230
( FUZZ_SPECIAL=1 SOAK_VT=200,6000 SOAK_TIMEOUT=500 timeout 600 python tools/soak_par.py m_special_v200 108000 110000 16 ) > $OUT/m_special.log 2>&1; echo "== special, 200 voices rc=$?"; tail -6 $OUT/m_special.log | cut -c1-230
