"""Print the kernel timeline of a rocprofv3 --kernel-trace run: python tools/timeline.py <dir with *_results.db> [max rows]"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, grid_x, start, end from kernels order by start"))
t0 = rows[0][2]
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{r[0][:44]:44s} grid={r[1]:<8d} start={(r[2]-t0)/1e6:8.2f} ms  dur={(r[3]-r[2])/1e6:7.3f} ms")
