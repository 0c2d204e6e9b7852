#!/bin/bash
# Round 5, pass Q: after "an overdriven ladder: no bound claimed" and the oscillators' magnitudes at 1 — the three finds; the noise family over the
# range that found 123042; the plain family over fresh seeds (the tighter magnitudes loosen decisions elsewhere); the suite; the default line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( SOAK_VT=200,6000 DBG_FLAGS=0,2,4 timeout 200 python tools/dbg_default.py noise 123042 ) > $OUT/q_dbg.log 2>&1; grep "^seed" $OUT/q_dbg.log | cut -c1-200
( SOAK_VT=200,6000 SOAK_TIMEOUT=200 timeout 260 python tools/soak_par.py q_noise_v200 120000 124000 16 noise ) > $OUT/q_noise.log 2>&1; echo "== noise family, 200 voices rc=$?"; tail -6 $OUT/q_noise.log | cut -c1-230
( SOAK_VT=200,6000 SOAK_TIMEOUT=200 timeout 260 python tools/soak_par.py q_plain_v200 130000 134000 16 ) > $OUT/q_plain.log 2>&1; echo "== plain family, 200 voices rc=$?"; tail -6 $OUT/q_plain.log | cut -c1-230
( timeout 600 python -m pytest tests -m gpu -q -n 8 ) > $OUT/q_suite.log 2>&1; echo "== suite rc=$?"; grep -E "passed|failed" $OUT/q_suite.log | tail -3; grep "^FAILED" $OUT/q_suite.log | head -10 | cut -c1-300
( timeout 300 python bench.py ) > $OUT/q_bench.log 2>&1; echo "== bench rc=$?"; grep '^{' $OUT/q_bench.log > $OUT/q_bench.json; cut -c1-400 $OUT/q_bench.json
