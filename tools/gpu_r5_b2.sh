#!/bin/bash
# Round 5, pass B2: the 10 000-seed soaks (16 voices x 1 s, 200 voices x 6 000 samples), 1 000 seeds at 16 voices x 10 s, and the soak through the
# specialised kernels, all under the derived error bound (same seeds as pass A under round 4's rule list); whatever is flagged is rendered
# again alone (tools/soak_par.py); then the random-patch survey at 262 144 voices
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
W=${SOAK_WORKERS:-16}
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/b2_tests.log 2>&1; echo "== tests rc=$?"; tail -12 $OUT/b2_tests.log | cut -c1-300
( SOAK_VT=16,48000 SOAK_TIMEOUT=1700 timeout 1800 python tools/soak_par.py new_1s 50000 60000 $W ) > $OUT/b2_soak_1s.log 2>&1; echo "== soak 1s rc=$?"; tail -12 $OUT/b2_soak_1s.log | cut -c1-230
( SOAK_VT=200,6000 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py new_v200 60000 70000 $W ) > $OUT/b2_soak_v200.log 2>&1; echo "== soak 200x6000 rc=$?"; tail -12 $OUT/b2_soak_v200.log | cut -c1-230
( SOAK_VT=16,480000 SOAK_TIMEOUT=1500 timeout 1600 python tools/soak_par.py new_10s 70000 71000 $W ) > $OUT/b2_soak_10s.log 2>&1; echo "== soak 10s rc=$?"; tail -12 $OUT/b2_soak_10s.log | cut -c1-230
( SOAK_VT=16,48000 FUZZ_SPECIAL=1 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py new_special 71000 72000 $W ) > $OUT/b2_soak_special.log 2>&1; echo "== soak special rc=$?"; tail -12 $OUT/b2_soak_special.log | cut -c1-230
( timeout 900 python tools/patch_survey.py 0 60 262144 6000 ) > $OUT/b2_survey.txt 2>&1; echo "== survey rc=$?"; python - <<'PY'
import re
ms=[float(m.group(1)) for m in re.finditer(r"\s([0-9.]+) ms/s", open("gpurun_out/r5/b2_survey.txt").read())]
ms.sort(); print("survey: %d patches, median %.1f ms per second of audio, mean %.1f" % (len(ms), ms[len(ms)//2] if ms else 0, sum(ms)/max(len(ms),1)))
PY
