#!/bin/bash
# tools/trace_dispatches.sh [bench.py args] — every kernel dispatch of one bench step in launch order with its duration (rocprofv3
# --kernel-trace).  With co-scheduled control units the first launches of a render carry the units alone, one more unit per
# launch, so their durations give each unit's time per sample (config 2: LFO / oscillator 68 ns, + envelope / filter 101 ns).
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py ${@:---workload cfg2} --steps 1 --warmup 1 --no-cpu > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-40:]:
    print("%-28s grid %-8s start %9.1f us dur %8.1f us" % (r['Kernel_Name'][:28], r.get('Grid_Size_X', r.get('Grid_Size','')), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
