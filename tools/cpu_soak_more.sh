#!/bin/bash
# After cpu_soak_all.sh: the NonLinear family (FUZZ_NONLIN: half of the MathModules become waveshapers — the fuzzer had none until now, the bound's
# rules for them were derived without a soak) and whole minutes of single voices
set -u
OUT=gpurun_out/cpu; mkdir -p $OUT
W=${WORKERS:-7}
run() {
    local tag=$1 first=$2 last=$3 vt=$4 noise=${5:-}
    python tools/cpu_soak.py $first $last $noise --vt $vt --workers $W --json $OUT/$tag.json > $OUT/$tag.log 2>&1
    tail -1 $OUT/$tag.log >> $OUT/all.log; grep "^   seed" $OUT/$tag.log >> $OUT/all.log
}
FUZZ_NONLIN=1 run nonlin_v200 400000 420000 200,6000
FUZZ_NONLIN=1 run nonlin_noise_v200 420000 430000 200,6000 noise
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run nonlin_sine_more_ov_v200 430000 440000 200,6000
run plain_60s 440000 441500 1,2880000
run noise_60s 441500 443000 1,2880000 noise
echo "more done" >> $OUT/all.log
