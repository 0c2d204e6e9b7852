#!/bin/bash
# Round 5, pass R: the suite at HEAD (the oscillators' magnitudes: 1 up to one cycle per sample); the noise family over fresh seeds; smoke
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 400 python -m pytest tests -m gpu -q -n 8 ) > $OUT/r_suite.log 2>&1; echo "== suite rc=$?"; grep -E "passed|failed" $OUT/r_suite.log | tail -3; grep "^FAILED" $OUT/r_suite.log | head -10 | cut -c1-300
( SOAK_VT=200,6000 SOAK_TIMEOUT=100 timeout 130 python tools/soak_par.py r_noise_v200 134000 136500 16 noise ) > $OUT/r_noise.log 2>&1; echo "== noise family, 200 voices rc=$?"; tail -6 $OUT/r_noise.log | cut -c1-230
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
