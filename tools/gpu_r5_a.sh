#!/bin/bash
# Round 5, first GPU pass at round 4's HEAD (no new rule yet): the default contract's TIME axis (tools/horizon.py: a minute of every benchmarked
# workload against the oracle, per second) and the soaks DESIGN section 10 named first — 10 000 fresh seeds at 16 voices x 1 s and at 200 voices
# x 6 000 samples, 1 000 at 16 voices x 10 s — spread over worker processes (tools/soak_par.py); then the default bench line.
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
W=${SOAK_WORKERS:-24}
nproc > $OUT/a_nproc.txt; free -g | head -2 >> $OUT/a_nproc.txt
( HORIZON_FLAGS=0,32,34,1 timeout 900 python tools/horizon.py $OUT/horizon.json ) > $OUT/a_horizon.log 2>&1; echo "== horizon rc=$?"; tail -30 $OUT/a_horizon.log | cut -c1-230
( SOAK_VT=16,48000 SOAK_TIMEOUT=1100 timeout 1200 python tools/soak_par.py 1s 50000 60000 $W ) > $OUT/a_soak_1s.log 2>&1; echo "== soak 1s rc=$?"; tail -12 $OUT/a_soak_1s.log | cut -c1-230
( SOAK_VT=200,6000 SOAK_TIMEOUT=1100 timeout 1200 python tools/soak_par.py v200 60000 70000 $W ) > $OUT/a_soak_v200.log 2>&1; echo "== soak 200x6000 rc=$?"; tail -12 $OUT/a_soak_v200.log | cut -c1-230
( SOAK_VT=16,480000 SOAK_TIMEOUT=1100 timeout 1200 python tools/soak_par.py 10s 70000 71000 $W ) > $OUT/a_soak_10s.log 2>&1; echo "== soak 10s rc=$?"; tail -12 $OUT/a_soak_10s.log | cut -c1-230
( timeout 600 python bench.py ) > $OUT/a_bench.json 2> $OUT/a_bench.err; echo "== bench rc=$?"; cut -c1-400 $OUT/a_bench.json
rm -f $OUT/soak_*_w*.log.keep
