#!/bin/bash
# random patches in the default modes for TWENTY seconds each (16 voices x 960 000 samples), seed ranges side by side — half of them through the
# interpreter / fused kernels (flags 0, 2, 4), half through the specialised kernels (FUZZ_SPECIAL=1: flags 34, 38): the time axis of the
# contract past the benchmarked workloads (tools/horizon.py has those for a minute).  Run ON THE GPU BOX from the repo root: <first seed> [ranges]
set -u
OUT=gpurun_out/r6
mkdir -p $OUT
A=${1:-7000}
N=${2:-8}
for k in $(seq 0 $((N - 1))); do
  ( SOAK_VT=16,960000 python tools/fuzz_soak_default.py $((A + 15 * k)) $((A + 15 * k + 15)) 2>&1 | tail -3 ) > $OUT/soak_long_$k.txt &
  ( FUZZ_SPECIAL=1 SOAK_VT=16,960000 python tools/fuzz_soak_default.py $((A + 15 * k)) $((A + 15 * k + 15)) 2>&1 | tail -3 ) > $OUT/soak_long_s$k.txt &
done
wait
for k in $(seq 0 $((N - 1))); do cat $OUT/soak_long_$k.txt $OUT/soak_long_s$k.txt; done
