#!/bin/bash
# round 6's last soak pass at HEAD (the proved classes' series at degree 9, the read-back's hardening): run ON THE GPU BOX from the repo root
set -u
OUT=gpurun_out/r6
mkdir -p $OUT
( python tools/fm_x_soak.py 3000 4500 2>&1 | tail -4 ) > $OUT/soak_fmx.txt &
( FUZZ_SPECIAL=1 python tools/fuzz_soak_default.py 6000 6300 2>&1 | tail -4 ) > $OUT/soak_default_special.txt &
( python tools/fuzz_soak_default.py 6300 6600 2>&1 | tail -4 ) > $OUT/soak_default.txt &
( FUZZ_NONLIN=1 python tools/fuzz_soak.py 4000 4200 2>&1 | tail -4 ) > $OUT/soak_exact.txt &
wait
python tools/readback_bench.py 1 8 50 2>&1 | tail -6 > $OUT/readback.txt
for f in soak_fmx soak_default_special soak_default soak_exact readback; do echo "== $f"; cat $OUT/$f.txt; done
