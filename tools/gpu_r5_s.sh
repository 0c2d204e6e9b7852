#!/bin/bash
# Round 5, pass S (the round's last GPU minutes): the noise family with the oscillators' SINE port drawn too, 200 voices
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( FUZZ_SINE=1 SOAK_VT=200,6000 SOAK_TIMEOUT=110 timeout 140 python tools/soak_par.py s_noise_sine_v200 140000 144000 16 noise ) > $OUT/s_noise.log 2>&1; echo "== noise family + sine ports, 200 voices rc=$?"; tail -8 $OUT/s_noise.log | cut -c1-230
