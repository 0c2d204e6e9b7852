#!/bin/bash
# The CPU soak's shapes, one after the other (tools/cpu_soak.py); results in gpurun_out/cpu/*.json, one summary line each in gpurun_out/cpu/all.log
set -u
OUT=gpurun_out/cpu; mkdir -p $OUT
W=${WORKERS:-7}
run() { # tag first last vt [noise] (environment handed on)
    local tag=$1 first=$2 last=$3 vt=$4 noise=${5:-}
    python tools/cpu_soak.py $first $last $noise --vt $vt --workers $W --json $OUT/$tag.json > $OUT/$tag.log 2>&1
    tail -1 $OUT/$tag.log >> $OUT/all.log; grep "^   seed" $OUT/$tag.log >> $OUT/all.log
}
run plain_v200 201000 207000 200,6000
run noise_v200 221000 241000 200,6000 noise
FUZZ_MORE_OV=1 run more_ov_v200 241000 247000 200,6000
FUZZ_MORE_OV=1 run more_ov_noise_v200 261000 273000 200,6000 noise
FUZZ_SINE=1 run sine_v200 281000 283000 200,6000
FUZZ_SINE=1 run sine_noise_v200 291000 293000 200,6000 noise
FUZZ_SINE=1 FUZZ_MORE_OV=1 run sine_more_ov_noise_v200 301000 303000 200,6000 noise
run plain_1s 311000 313000 16,48000
run noise_1s 321000 323000 16,48000 noise
FUZZ_MORE_OV=1 run more_ov_1s 331000 333000 16,48000
FUZZ_MORE_OV=1 FUZZ_SINE=1 run sine_more_ov_noise_1s 341000 343000 16,48000 noise
run plain_10s 351000 352000 4,480000
run noise_10s 354000 355000 4,480000 noise
FUZZ_NONLIN=1 run nonlin_v200 400000 408000 200,6000
FUZZ_NONLIN=1 run nonlin_noise_v200 420000 422000 200,6000 noise
FUZZ_NONLIN=1 FUZZ_MORE_OV=1 FUZZ_SINE=1 run nonlin_sine_more_ov_v200 430000 432000 200,6000
run plain_60s 440000 440300 1,2880000
run noise_60s 441500 441800 1,2880000 noise
echo "all done" >> $OUT/all.log
