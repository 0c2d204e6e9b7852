#!/bin/bash
# More shapes for the CPU soak, after cpu_soak_all.sh's final series: everything drawn at once, half seconds of 64 voices, more seeds of the plain families
set -u
OUT=gpurun_out/cpu_extra; mkdir -p $OUT
W=${WORKERS:-7}
run() {
    local tag=$1 first=$2 last=$3 vt=$4 noise=${5:-}
    python tools/cpu_soak.py $first $last $noise --vt $vt --workers $W --json $OUT/$tag.json > $OUT/$tag.log 2>&1
    tail -1 $OUT/$tag.log >> $OUT/all.log; grep "^   seed" $OUT/$tag.log >> $OUT/all.log
}
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run all_at_once_noise_v200 500000 508000 200,6000 noise
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run all_at_once_v200 510000 518000 200,6000
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run all_at_once_half_s 520000 524000 64,24000
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run all_at_once_noise_half_s 530000 534000 64,24000 noise
run plain_v200_more 540000 560000 200,6000
run noise_v200_more 560000 580000 200,6000 noise
echo "extra done" >> $OUT/all.log
