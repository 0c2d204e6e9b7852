#!/bin/bash
# Round-3 rocprofv3 evidence: one profile set per workload / mode (profiles/run_profile.sh does the passes, profiles/summarize.py the
# timed-steps-only summaries and their checks against the traced run's own bench line).
set -u
mkdir -p gpurun_out/r3prof
for spec in "r03||x" "r03_cfg2|--workload cfg2|" "r03_cfg4|--workload cfg4|" "r03_cfg4_b1024|--workload cfg4_b1024|" "r03_special|--flags 2|" "r03_p3|--workload p3|" "r03_exact|--flags 1|"; do
  IFS='|' read -r tag args cpu <<< "$spec"
  echo "=== $tag ($args)"
  if [ -n "$cpu" ]; then bash profiles/run_profile.sh "$tag" "$args" " " > gpurun_out/r3prof/$tag.log 2>&1; else bash profiles/run_profile.sh "$tag" "$args" > gpurun_out/r3prof/$tag.log 2>&1; fi
  grep -E "^(ok|FAIL) |summarize rc" gpurun_out/r3prof/$tag.log | cut -c1-220
done
# the hand-written FM pair the specialised kernel replaced by default, for the record
echo "=== r03_cfg4_fused"
SRACK_FM_FUSED=1 bash profiles/run_profile.sh r03_cfg4_fused "--workload cfg4" > gpurun_out/r3prof/r03_cfg4_fused.log 2>&1
grep -E "^(ok|FAIL) |summarize rc" gpurun_out/r3prof/r03_cfg4_fused.log | cut -c1-220
mkdir -p gpurun_out/profiles && cp profiles/r03* gpurun_out/profiles/
rm -rf gpurun_out/prof_r03*/  # the raw rocprofv3 databases stay on the box: only the condensed summaries travel
