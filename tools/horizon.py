"""Error against the oracle over TIME: every benchmarked workload in the default (approximating) render modes, rendered for a minute or more
in calls of one second (state carried from call to call), max |gpu - ref| / max(|ref|, 1) per second.  An offline renderer's contract
cannot stop at one second: a constant per-sample bias in a phase increment is a ramp, and only a curve over time tells a ramp from a plateau.

usage: horizon.py [out.json]      HORIZON_SECONDS (60), HORIZON_VOICES (64), HORIZON_WORKLOADS (comma list), HORIZON_FLAGS (comma list: 0,32,34)
Test infrastructure: the oracle is the checker (oracle/), the product path is what is rendered."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O

SR = 48000


def horizon(S, name, V, seconds, flags, ref=None, threads=8):
    """-> (per-second dict, ref) for workload `name` at V voices"""
    B, build, overrides = S.bench_workload(name, V)
    if ref is None:
        o = O.OraclePatch(SR, B, 2)
        ids = build(o)
        ref, _ = o.render_batch(V, seconds * SR, overrides(ids), threads=threads)   # [2][T][V]
    p = S.Patch(SR, B, 2)
    ids = build(p)
    p.configure_voices(V)
    for m, f, v in overrides(ids):
        p.set_voice_field(m, f, v)
    per_s, per_s_abs, nonfinite = [], [], 0
    for s in range(seconds):
        fr = p.render_channels(SR, flags)   # [2][SR][V]; the patch continues from where the last call stopped
        r = ref[:, s * SR:(s + 1) * SR].astype(np.float64)
        g = fr.astype(np.float64)
        ok = np.isfinite(r) & np.isfinite(g)
        nonfinite += int((~ok).sum())
        d = np.abs(np.where(ok, g - r, 0.0))
        per_s_abs.append(float(d.max()))
        per_s.append(float((d / np.maximum(np.abs(np.where(ok, r, 0.0)), 1.0)).max()))
    info = p.info()
    return dict(workload=name, voices=V, seconds=seconds, flags=flags, kernel=info.split("kernel=")[-1], max_rel_err_per_second=per_s,
                max_abs_err_per_second=per_s_abs, worst=max(per_s), first_second=per_s[0], last_second=per_s[-1],
                ref_peak=float(np.nanmax(np.abs(ref))), nonfinite_samples=nonfinite), ref


def main():
    S = srack_pkg.load()
    O.build()
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    seconds = int(os.environ.get("HORIZON_SECONDS", "60"))
    V = int(os.environ.get("HORIZON_VOICES", "64"))
    names = os.environ.get("HORIZON_WORKLOADS", "cfg3,cfg3_poly,cfg4,cfg4_b1024,p3,p4").split(",")
    flag_list = [int(x) for x in os.environ.get("HORIZON_FLAGS", "0,32,34").split(",")]
    threads = min(V, os.cpu_count() or 8)
    rows = []
    for name in names:
        ref = None
        for flags in flag_list:
            t0 = time.time()
            try:
                row, ref = horizon(S, name, V, seconds, flags, ref, threads)
            except S.SrackError as e:
                print(f"{name} flags {flags}: {e}", flush=True)
                continue
            row["wall_s"] = time.time() - t0
            rows.append(row)
            e = row["max_rel_err_per_second"]
            q = [e[0], e[len(e) // 4], e[len(e) // 2], e[3 * len(e) // 4], e[-1]]
            print(f"{name:11s} flags {flags:2d} {row['kernel'][:40]:40s} worst {row['worst']:.2e}  at 0 / 25 / 50 / 75 / 100 % of {seconds} s: "
                  + " ".join(f"{x:.2e}" for x in q) + f"  ({row['wall_s']:.0f} s)", flush=True)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(dict(what="max |gpu - oracle| / max(|oracle|, 1) per second of a render carried across one-second calls (tools/horizon.py)",
                           sample_rate=SR, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
