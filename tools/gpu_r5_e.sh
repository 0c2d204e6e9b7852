#!/bin/bash
# Round 5, final pass at HEAD: the rocprofv3 evidence (profiles/r05_*), the error-over-time curves, the soaks (retrying whatever differs on the
# spot, and again alone afterwards), the random-patch survey, the default bench line
set -u
OUT=gpurun_out/r5; mkdir -p $OUT gpurun_out/profiles
prof() { tag=$1; shift; bash profiles/run_profile.sh "$tag" "$*" > $OUT/prof_$tag.log 2>&1; grep -E "^(ok|FAIL) |summarize rc" $OUT/prof_$tag.log | cut -c1-200 | tail -6; cp profiles/${tag}_* gpurun_out/profiles/ 2>/dev/null; rm -rf gpurun_out/prof_$tag; }
prof r05
prof r05_exact --flags 1
prof r05_poly --workload cfg3_poly
prof r05_cfg4 --workload cfg4
prof r05_cfg4_fast --workload cfg4 --flags 64
prof r05_cfg2 --workload cfg2
prof r05_p3 --workload p3
prof r05_p4 --workload p4
( HORIZON_FLAGS=0,32,34,1,64 timeout 900 python tools/horizon.py gpurun_out/profiles/r05_horizon.json ) > $OUT/e_horizon.log 2>&1; echo "== horizon rc=$?"; grep -c worst $OUT/e_horizon.log
W=16
( SOAK_RETRY=1 SOAK_VT=16,48000 SOAK_TIMEOUT=1500 timeout 1600 python tools/soak_par.py final_1s 50000 60000 $W ) > $OUT/e_soak_1s.log 2>&1; echo "== soak 1s rc=$?"; tail -6 $OUT/e_soak_1s.log | cut -c1-230
( SOAK_RETRY=1 SOAK_VT=200,6000 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py final_v200 60000 70000 $W ) > $OUT/e_soak_v200.log 2>&1; echo "== soak 200x6000 rc=$?"; tail -6 $OUT/e_soak_v200.log | cut -c1-230
( SOAK_RETRY=1 SOAK_VT=16,480000 SOAK_TIMEOUT=1500 timeout 1600 python tools/soak_par.py final_10s 70000 71000 $W ) > $OUT/e_soak_10s.log 2>&1; echo "== soak 10s rc=$?"; tail -6 $OUT/e_soak_10s.log | cut -c1-230
( SOAK_RETRY=1 SOAK_VT=16,48000 FUZZ_SPECIAL=1 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py final_special 71000 72000 $W ) > $OUT/e_soak_special.log 2>&1; echo "== soak special rc=$?"; tail -6 $OUT/e_soak_special.log | cut -c1-230
grep -h "^RETRY" $OUT/soak_final_*_w*.log | cut -c1-330 | head -30
( timeout 900 python tools/patch_survey.py 0 60 262144 6000 ) > gpurun_out/profiles/r05_survey.txt 2>&1; echo "== survey rc=$?"
( timeout 900 python bench.py ) > gpurun_out/profiles/r05_default_line.json 2> $OUT/e_bench.err; echo "== bench rc=$?"
