export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/clk
cd /tmp
for v in frames noframes; do
  A=""; [ $v = noframes ] && A="--no-frames"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $ROOT/gpurun_out/clk/$v -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu $A > $ROOT/gpurun_out/clk/$v.log 2>&1
done
cd $ROOT
python - <<'PY'
import sqlite3, glob
for v in ("frames","noframes"):
    db = glob.glob(f"gpurun_out/clk/{v}/**/*.db", recursive=True)[0]
    con = sqlite3.connect(db)
    q = "select kernel_name, counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%voice_chain%' group by counter_name"
    rows = list(con.execute(q))
    d = {r[1]: r[2] for r in rows}
    dur = rows[0][3]
    print(v, "dur_ns", dur, {k: round(x/1e6,1) for k,x in d.items()}, "clock GHz (GRBM/8/dur)", d["GRBM_GUI_ACTIVE"]/8/dur)
PY
