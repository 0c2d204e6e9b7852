"""tools/energy_table.py [profiles/r04_energy.txt] — energy per wave-instruction from tools/energy_probe.sh's log: (median package power while
the variant ran - the power of the empty loop) x time per launch / wave-instructions per launch, and the issue cycles per instruction."""
import json, re, statistics, sys
path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_energy.txt"
NAMES = {13: "empty loop (clocks up, nothing issued)", 0: "v_fma_f32, three VGPR sources", 12: "v_fma_f32, SGPR + inline constant", 1: "v_add_f32", 2: "v_mul_f32",
         3: "v_med3_f32 (clamp to constants)", 4: "v_cmp_lt_f32 + v_cndmask_b32 (each)", 5: "v_add_f64", 6: "v_fma_f64", 14: "v_fract_f64",
         7: "v_add_co / v_addc_co / v_cvt_f32_u32 / v_add_u32 (each)", 8: "v_mov_b32", 9: "16 ds_write_b32 + 48 v_fma_f32", 10: "48 v_fma_f32",
         11: "s_add_u32 / s_xor_b32", 16: "v_fma_f32, ONE lane enabled (exec = 1)", 17: "v_fma_f32, sixteen lanes enabled", 15: "v_pk_fma_f32 (two values per lane)"}
PER_ITER = {9: 64, 10: 48, 13: 1, 15: 32}
rows = {}
for ln in open(path):
    if not ln.startswith("{"):
        continue
    d = json.loads(ln[:ln.index("}") + 1])
    s = re.findall(r"\((\d+)Mhz\),([\d.]+)", ln)
    s = [(int(c), float(p)) for c, p in s if int(c) > 2300]   # (a sample taken after the run ended shows the clock falling)
    if not s:
        continue
    rows[d["variant"]] = (d, statistics.median(p for _, p in s), statistics.median(c for c, _ in s))
base = rows[13][1]
print(f"empty loop: {base:.0f} W (idle, clocks down: see the log's first line)")
out = {}
for v, (d, watts, mhz) in rows.items():
    if v in (13, 11):
        continue
    n = d["waves"] * d["iters"] * PER_ITER.get(v, 64)
    t = d["ms_per_launch"] * 1e-3
    nj = (watts - base) * t / n * 1e9
    cyc = t * mhz * 1e6 / (n / 1024)
    out[v] = (nj, cyc, watts)
    print(f"{NAMES[v]:60s} {watts:6.0f} W  {cyc:5.2f} cycles  {nj:5.2f} nJ per wave-instruction")
if 9 in rows and 10 in rows:
  ds = (rows[9][1] - base) * rows[9][0]["ms_per_launch"] * 1e-3 - (rows[10][1] - base) * rows[10][0]["ms_per_launch"] * 1e-3
  print(f"{'ds_write_b32 (difference of the two rows above)':60s} {'':8s} {(rows[9][0]['ms_per_launch'] - rows[10][0]['ms_per_launch']) * 1e-3 * rows[9][2] * 1e6 / (4096 * 2000 * 16 / 1024):5.2f} cycles  {ds / (4096 * 2000 * 16) * 1e9:5.2f} nJ")
