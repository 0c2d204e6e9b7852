#!/bin/bash
# round 3: frames of a full tile through eight dwordx4 stores from the LDS tile (-DSRK_STORE_X4=1) against the per-sample dword stores
set -u
OUT=gpurun_out/r3r
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/ab.sh "--no-side-configs" tools/ab_libs/base.so tools/ab_libs/x4.so 2>&1 | tee $OUT/ab_x4.log
cp s-rack_amd/libsrack_hip.so /tmp/_keep.so; cp tools/ab_libs/x4.so s-rack_amd/libsrack_hip.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg3 or p1_voices or cfg1" ) > $OUT/pytest_x4.log 2>&1; tail -3 $OUT/pytest_x4.log | cut -c1-300
cp /tmp/_keep.so s-rack_amd/libsrack_hip.so
