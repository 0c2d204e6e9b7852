#!/bin/bash
# the round's last GPU pass: the suite, the rocprofv3 evidence of every benchmarked configuration at HEAD (-> gpurun_out/profiles), the default line
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/suite_final.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|error" $OUT/suite_final.log | tail -3
bash tools/gpu_r4.sh prof r04
bash tools/gpu_r4.sh prof r04_poly --workload cfg3_poly
bash tools/gpu_r4.sh prof r04_cfg4 --workload cfg4
bash tools/gpu_r4.sh prof r04_p3 --workload p3
bash tools/gpu_r4.sh prof r04_exact --flags 1
bash tools/gpu_r4.sh prof r04_cfg2 --workload cfg2
bash tools/default_line.sh
