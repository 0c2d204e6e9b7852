#!/bin/bash
set -u
OUT=gpurun_out/r3j
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "envelope_scaled or fm_pair or cfg4_golden" ) > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8 | cut -c1-300
