#!/bin/bash
# round 4's regression soaks (GPU box): the properties the earlier rounds' soaks pinned, past their seeds, after this round's flattener rules,
# quiet groups, fixed-point saws and the Segment refactor
set -u
OUT=gpurun_out/r4; mkdir -p $OUT
run() { name=$1; shift; ( timeout 900 python "$@" ) > $OUT/soak_$name.log 2>&1; echo "== $name rc=$?"; tail -3 $OUT/soak_$name.log | cut -c1-220; }
run keep tools/fuzz_soak_keep.py 300 420
run cont tools/fuzz_soak_cont.py 300 420
run tick tools/fuzz_soak_tick.py 300 380
run srk tools/srk_soak.py 300 420
run state tools/state_soak.py 300 400
run mix tools/mix_soak.py 300 380
run shape tools/shape_soak.py 300 380
run smp tools/smp_soak.py 300 400
run cfg tools/fuzz_soak_cfg.py 300 380
