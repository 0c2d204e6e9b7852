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
# Whatever was flagged is rendered again ALONE, one process, the device otherwise idle: a structural violation of the contract reproduces;
# what only shows while two dozen processes share the device (pass A of round 5 had such) does not.
flagged = sorted(set(int(w[0]) for w in merged["worst"]))
if flagged and script.endswith("fuzz_soak_default.py"):
    import re
    env = dict(os.environ, DBG_FLAGS="34,38" if os.environ.get("FUZZ_SPECIAL") else "0,2,4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg_default.py")] + (["noise"] if extra else []) + [str(x) for x in flagged[:40]],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    again = {}
    for m in re.finditer(r"seed (\d+) flags (\d+): max ([0-9.e+-]+|nan)", r.stdout):
        e = float(m.group(3))
        again[int(m.group(1))] = max(again.get(int(m.group(1)), 0.0), e if e == e else 9.9)
    repro = sorted(s for s, e in again.items() if e > 1e-5)
    merged["flagged"], merged["rerun_alone"], merged["reproduced_alone"] = flagged, again, repro
    json.dump(merged, open(os.path.join(out_dir, f"soak_{tag}.json"), "w"), indent=1)
    print(f"   flagged {len(flagged)} patches; rendered alone afterwards ({len(again)} of them): {len(repro)} reproduce {repro[:20]}", flush=True)
    open(os.path.join(out_dir, f"soak_{tag}_rerun.log"), "w").write(r.stdout + r.stderr[-3000:])
