#!/bin/bash
set -u
OUT=gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -12 | cut -c1-300
