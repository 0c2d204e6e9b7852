export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/pmci
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $ROOT/gpurun_out/pmci/a -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --flags ${FLAGS:-6} > $ROOT/gpurun_out/pmci/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_BRANCH -d $ROOT/gpurun_out/pmci/b -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --flags ${FLAGS:-6} > $ROOT/gpurun_out/pmci/b.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3, glob
for v in ("a","b"):
    db = glob.glob(f"gpurun_out/pmci/{v}/**/*.db", recursive=True)[0]
    con = sqlite3.connect(db)
    q = "select counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%render_interp%' group by counter_name"
    for r in con.execute(q): print(v, r[0], round(r[1]/1e6,2), "M   dur_ms", round(r[2]/1e6,2))
PY
