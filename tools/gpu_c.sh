#!/bin/bash
# A/B of the co-scheduled control units on P3 + a kernel trace of both
set -u
OUT=gpurun_out/c
mkdir -p $OUT
export TMPDIR=/tmp
for mode in 1 0; do
  SRACK_SPECIAL_CTL=$mode python bench.py --workload p3 --steps 5 --warmup 2 --no-cpu > $OUT/p3_ctl$mode.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/p3_ctl$mode.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("special_ctl=$mode ms/step %.2f kernel %s x%d %.3f ms" % (d["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms"]))
PY
done
cd /tmp
for mode in 1 0; do
  SRACK_SPECIAL_CTL=$mode rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace$mode -o t -- python $GRAFT_REPO_ROOT/bench.py --workload p3 --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$OUT/trace$mode.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
for mode in (1, 0):
    dbs = glob.glob(f"gpurun_out/c/trace{mode}/**/*.db", recursive=True)
    if not dbs: print("no db", mode); continue
    con = sqlite3.connect(dbs[0])
    print("mode", mode)
    for r in con.execute("select name, total_calls, total_duration, average from top_kernels"): print("  ", r)
    try:
        rows = list(con.execute("select k.kernel_name, d.start, d.end from kernels k"))
    except Exception as e:
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
        print("   tables:", [t for t in tabs if 'kernel' in t.lower() or 'dispatch' in t.lower()][:12])
PY
