#!/bin/bash
# Round-2 rocprofv3 evidence: one profile set per workload / mode (profiles/run_profile.sh does the passes).
set -u
for spec in "r02||x" "r02_exact|--flags 1|" "r02_special|--flags 2|" "r02_special_exact|--flags 3|" "r02_p3_exact|--workload p3 --flags 1|" "r02_cfg4|--workload cfg4|x" "r02_cfg4_b1024|--workload cfg4_b1024|" "r02_p3|--workload p3|" "r02_p3_dist|--workload p3 --force-dist|" "r02_cfg2|--workload cfg2|x"; do
  IFS='|' read -r tag args cpu <<< "$spec"
  echo "=== $tag ($args)"
  if [ -n "$cpu" ]; then bash profiles/run_profile.sh "$tag" "$args" " " > gpurun_out/prof_$tag.log 2>&1; else bash profiles/run_profile.sh "$tag" "$args" > gpurun_out/prof_$tag.log 2>&1; fi
  tail -3 gpurun_out/prof_$tag.log | cut -c1-200
done
mkdir -p gpurun_out/profiles && cp profiles/r02* gpurun_out/profiles/
rm -rf gpurun_out/prof_r02*/  # the raw rocprofv3 databases stay on the box: only the condensed summaries travel
