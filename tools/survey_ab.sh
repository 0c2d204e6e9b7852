#!/bin/bash
# the random-patch survey (tools/patch_survey.py) under three settings on ONE box: round 3's general path (no register budget, no quiet
# groups, f64 saw phases), round 4 without the register budget, round 4 — and the per-seed ratios (tools/survey_compare.py)
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
A=${1:-0}; B=${2:-60}; V=${3:-262144}; T=${4:-6000}
SRACK_JIT_BUDGET=0 SRACK_JIT_QUIET=0 python tools/patch_survey.py $A $B $V $T > $OUT/survey_r3like.txt 2>&1
SRACK_JIT_BUDGET=0 python tools/patch_survey.py $A $B $V $T > $OUT/survey_nobudget.txt 2>&1
python tools/patch_survey.py $A $B $V $T > $OUT/survey_r4.txt 2>&1
python tools/survey_compare.py $OUT/survey_r3like.txt $OUT/survey_nobudget.txt $OUT/survey_r4.txt | tee $OUT/survey_compare.txt | tail -75
