#!/bin/bash
# round 3, re-entry pass: the whole -m gpu suite at HEAD, the default bench line, smoke()
set -u
OUT=gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-300
( time timeout 300 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json
tail -3 $OUT/bench_default.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
