#!/bin/bash
# the survey's seeds with and without extra options for the specialised kernels (SRACK_JIT_OPTS), alternating on one box
OPTS=$1; SEEDS=$2
for round in 1 2; do for which in none opts; do
  if [ $which = none ]; then unset SRACK_JIT_OPTS; else export SRACK_JIT_OPTS="$OPTS"; fi
  echo "== $which"; SURVEY_SEEDS=$SEEDS python tools/patch_survey.py 0 0 262144 24000 2>&1 | grep "^seed" | cut -c1-75
done; done
