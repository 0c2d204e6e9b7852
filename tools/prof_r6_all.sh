#!/bin/bash
# round 6's rocprofv3 evidence for every workload on the default line, at HEAD (run ON THE GPU BOX from the repo root)
set -u
bash tools/gpu_r6.sh prof r06 "" prof r06_exact "--flags 1" prof r06_poly "--workload cfg3_poly" prof r06_cfg2 "--workload cfg2" \
  prof r06_cfg4 "--workload cfg4" prof r06_cfg4_fast "--workload cfg4 --flags 64" prof r06_cfg4_b1024 "--workload cfg4_b1024" \
  prof r06_cfg4_b1024_fast "--workload cfg4_b1024 --flags 64" prof r06_p3 "--workload p3" prof r06_p4 "--workload p4"
