"""tools/fuzz_soak_default.py (or another soak script with the same <first> <last> [noise] arguments and SOAK_JSON) over a seed range split
across W worker processes on one GPU box — the soaks' renders are a few waves each, so a dozen processes share the device without
slowing one another much, and the oracle's threads find idle host cores.
usage: soak_par.py <tag> <first> <last> <workers> [noise]      (environment — SOAK_VT, FUZZ_SPECIAL ... — is handed on)
Writes gpurun_out/r5/soak_<tag>.json (merged) and prints one summary line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, lo, hi, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extra = sys.argv[5:]
script = os.environ.get("SOAK_SCRIPT", "tools/fuzz_soak_default.py")
out_dir = os.path.join(ROOT, "gpurun_out", "r5")
os.makedirs(out_dir, exist_ok=True)
t0 = time.time()
procs = []
for w in range(W):
    a, b = lo + (hi - lo) * w // W, lo + (hi - lo) * (w + 1) // W
    if a == b:
        continue
    js = os.path.join(out_dir, f"soak_{tag}_w{w}.json")
    if os.path.exists(js):
        os.remove(js)
    env = dict(os.environ, SOAK_JSON=js)
    log = open(os.path.join(out_dir, f"soak_{tag}_w{w}.log"), "w")
    procs.append((w, a, b, js, subprocess.Popen([sys.executable, os.path.join(ROOT, script), str(a), str(b)] + extra, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)))
limit = float(os.environ.get("SOAK_TIMEOUT", "3000"))
merged = dict(tag=tag, first=lo, last=hi, workers=W, script=script, vt=os.environ.get("SOAK_VT", ""), special=bool(os.environ.get("FUZZ_SPECIAL")),
              noise=bool(extra), renders=0, bad=0, worst=[], failed_workers=[], seeds_done=0)
for w, a, b, js, p in procs:
    try:
        rc = p.wait(timeout=max(1.0, limit - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
        rc = 124
    if rc != 0 or not os.path.exists(js):
        merged["failed_workers"].append([w, a, b, rc])
        continue
    d = json.load(open(js))
    merged["renders"] += d["renders"]
    merged["bad"] += d["bad"]
    merged["worst"] += d["worst"]
    merged["seeds_done"] += b - a
    os.remove(js)
merged["wall_s"] = time.time() - t0
json.dump(merged, open(os.path.join(out_dir, f"soak_{tag}.json"), "w"), indent=1)
print(f"soak {tag}: seeds {lo}..{hi - 1} ({merged['seeds_done']} done) VT={merged['vt'] or 'default'} special={merged['special']} noise={merged['noise']}: "
      f"{merged['renders']} renders, {merged['bad']} outside the band, failed workers {merged['failed_workers']}, {merged['wall_s']:.0f} s", flush=True)
for w in sorted(merged["worst"], key=lambda x: -x[2])[:20]:
    print("   seed %d flags %d: max rel err %.2e, %.5f of the samples outside, masks equal %s" % tuple(w[:5]), flush=True)
