#!/bin/bash
# Round 5, pass O: seeds 104123 / 105055 after "an audio-rate cutoff: no bound claimed"; the per-voice-parameter soak at 200 voices again over
# the range that found them and 4 000 new seeds; the suite; the default line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( FUZZ_MORE_OV=1 SOAK_VT=200,6000 DBG_FLAGS=0,2,4 timeout 300 python tools/dbg_default.py 105055 104123 ) > $OUT/o_dbg.log 2>&1; grep "^seed" $OUT/o_dbg.log | cut -c1-200
( FUZZ_MORE_OV=1 SOAK_VT=200,6000 SOAK_TIMEOUT=400 timeout 500 python tools/soak_par.py o_more_ov_v200 104000 112000 16 ) > $OUT/o_more.log 2>&1; echo "== more overrides, 200 voices rc=$?"; tail -8 $OUT/o_more.log | cut -c1-230
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/o_suite.log 2>&1; echo "== suite rc=$?"; tail -4 $OUT/o_suite.log
( timeout 600 python bench.py ) > $OUT/o_bench.log 2>&1; echo "== bench rc=$?"; grep '^{' $OUT/o_bench.log > $OUT/o_bench.json; cut -c1-900 $OUT/o_bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
