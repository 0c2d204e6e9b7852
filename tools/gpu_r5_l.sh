#!/bin/bash
# Round 5, pass L: seed 72223 after "a lowpass keeps its input's edges"; the suite and smoke(); the default modes through the specialised kernels
# over 4 000 fresh seeds; the default line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT gpurun_out/profiles
( SOAK_VT=16,48000 DBG_FLAGS=34,38,0,2,4 timeout 300 python tools/dbg_default.py 72223 ) > $OUT/l_72223.log 2>&1; grep "^seed" $OUT/l_72223.log | cut -c1-160
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/l_tests.log 2>&1; echo "== tests rc=$?"; tail -3 $OUT/l_tests.log | cut -c1-300
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/l_smoke.log 2>&1; echo "== smoke rc=$?"; tail -1 $OUT/l_smoke.log | cut -c1-300
( FUZZ_SPECIAL=1 SOAK_VT=16,48000 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py l_special_1s 92000 96000 16 ) > $OUT/l_soak.log 2>&1; echo "== special soak rc=$?"; tail -5 $OUT/l_soak.log | cut -c1-230
( timeout 900 python bench.py ) > gpurun_out/profiles/r05_default_line.json 2> $OUT/l_bench.err; echo "== bench rc=$?"; cut -c1-200 gpurun_out/profiles/r05_default_line.json
