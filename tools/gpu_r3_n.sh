#!/bin/bash
set -u
OUT=gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-300
