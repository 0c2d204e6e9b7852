#!/bin/bash
# Round 5, pass H: srack_device_to_host through the library's pinned buffer — the suite, then the soaks once more, 16 processes, every render
# read back twice (pass G: the two read-backs of ONE render differed — the copy into pageable memory lost stretches, not the device)
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 1200 ) > $OUT/h_tests.log 2>&1; echo "== tests rc=$?"; tail -4 $OUT/h_tests.log | cut -c1-300
W=16
( SOAK_RETRY=2 SOAK_VT=16,48000 SOAK_TIMEOUT=1500 timeout 1600 python tools/soak_par.py h_1s 50000 60000 $W ) > $OUT/h_soak_1s.log 2>&1; echo "== soak 1s rc=$?"; tail -6 $OUT/h_soak_1s.log | cut -c1-230
( SOAK_RETRY=2 SOAK_VT=200,6000 SOAK_TIMEOUT=900 timeout 1000 python tools/soak_par.py h_v200 60000 70000 $W ) > $OUT/h_soak_v200.log 2>&1; echo "== soak 200x6000 rc=$?"; tail -6 $OUT/h_soak_v200.log | cut -c1-230
grep -h "^RETRY\|^READBACK" $OUT/soak_h_*_w*.log | cut -c1-300 | head -20
echo "READBACK lines: $(grep -h '^READBACK' $OUT/soak_h_*_w*.log | wc -l)"
