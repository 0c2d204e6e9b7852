#!/bin/bash
# Round 5, pass K: the EXACT modes at HEAD, bit for bit against the oracle (the exact oscillator's forms changed this round: the checked sine,
# Markstein's division, the merged cold path) — short renders, whole seconds, with the sine port drawn too, through the specialised kernels;
# and the default modes through the specialised kernels (what 4096 voices and more get) over 6 000 more seeds
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
W=16
run() { tag=$1; shift; ( SOAK_TIMEOUT=1100 timeout 1200 python tools/soak_par.py $tag "$@" ) > $OUT/k_$tag.log 2>&1; echo "== $tag rc=$?"; tail -5 $OUT/k_$tag.log | cut -c1-230; }
SOAK_SCRIPT=tools/fuzz_soak.py run x_short 80000 84000 $W
SOAK_SCRIPT=tools/fuzz_soak.py FUZZ_SINE=1 run x_sine 84000 88000 $W
SOAK_SCRIPT=tools/fuzz_soak.py FUZZ_SINE=1 SOAK_VT=16,48000 run x_sine_1s 88000 89500 $W
SOAK_SCRIPT=tools/fuzz_soak.py FUZZ_SPECIAL=1 FUZZ_SINE=1 run x_special 89500 91500 $W
FUZZ_SPECIAL=1 SOAK_VT=16,48000 run d_special_1s 72000 78000 $W
FUZZ_SINE=1 SOAK_VT=16,48000 run d_sine_1s 78000 80000 $W
