"""The flattener's decisions for the fuzzer's patches under TWO builds of the library (CPU only: read off the kernel source the generator
writes): per patch, which oscillators get the exact PolyBLEP, which filters the literal ladder, which sines / shapers / saws a loose form,
and whether the whole patch went exact.  Used to check a change of the decision procedure (round 5: the soak-derived rule list against
approx.cpp's error budget) before it costs GPU time: where the new procedure is LOOSER than the old one on a patch, that patch is worth a render.

usage: approx_compare.py <repo root of build A> <first> <last> [noise]          (build B = this tree)
       prints one line per differing patch and a summary; APPROX_DUMP=1: every patch's decisions of this tree only."""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decisions(root, lo, hi, noise):
    """runs in a child (one process can load only one libsrack_hip.so): seed -> decision dict"""
    code = r'''
import sys, re, json
sys.path.insert(0, %r)
import srack_pkg
S = srack_pkg.load()
sys.path.insert(0, %r)
from tests.fuzz_patches import random_patch
out = {}
for seed in range(%d, %d):
    B, build, overrides = random_patch(seed, %r)
    p = S.Patch(48000, B, 2)
    ids = build(p)
    V = 8
    p.configure_voices(V)
    for m, f, fn in overrides:
        p.set_voice_field(ids[m], f, fn(V))
    try:
        src = p.kernel_source(S.RENDER_NO_UNIFORM_HOIST | S.RENDER_NO_FUSION)
    except S.SrackError as e:
        out[seed] = {"error": e.code}
        continue
    osc = [int(x, 16) for x in re.findall(r"osc_step\(\(?(0x[0-9a-f]+)u", src)]
    d = {"exact": int(any(f & 0x40 for f in osc) or "xsaw_" in src or "cosc_tile<true>" in src),
         "exact_blep": sum(1 for f in osc if f & 0x2000), "literal": src.count("vcf_run<false>") + src.count("vcf_run_bounded"),
         "fast_ladder": src.count("vcf_run<true>"), "fixed_saw": src.count("fosc_saw"), "nonlin_loose": len(re.findall(r"nonlin_step\((0x[0-9a-f]*2[0-9a-f]{2})u", src)),
         "sine_loose": sum(1 for f in osc if f & 0x1000), "head": src.split("\n", 1)[0][-60:]}
    out[seed] = d
print(json.dumps(out))
''' % (root, ROOT, lo, hi, noise)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    return {int(k): v for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}


def main():
    other, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    noise = len(sys.argv) > 4
    new = decisions(ROOT, lo, hi, noise)
    if os.environ.get("APPROX_DUMP"):
        for s, d in sorted(new.items()):
            print(s, d)
    old = decisions(other, lo, hi, noise)
    looser, stricter, same = [], [], 0
    for s in range(lo, hi):
        a, b = old[s], new[s]
        if "error" in a or "error" in b:
            continue
        ka = (a["exact"], a["exact_blep"], a["literal"], -a["fixed_saw"], -a["sine_loose"], -a["nonlin_loose"])
        kb = (b["exact"], b["exact_blep"], b["literal"], -b["fixed_saw"], -b["sine_loose"], -b["nonlin_loose"])
        if ka == kb:
            same += 1
        elif a["exact"] and not b["exact"] or (a["exact"] == b["exact"] and any(y < x for x, y in zip(ka[1:], kb[1:]))):
            looser.append((s, a, b))
        else:
            stricter.append((s, a, b))
    fmt = lambda d: "exact" if d["exact"] else "blep %d lit %d fast %d fixed %d sineL %d" % (d["exact_blep"], d["literal"], d["fast_ladder"], d["fixed_saw"], d["sine_loose"])
    for tag, rows in (("LOOSER", looser), ("stricter", stricter)):
        for s, a, b in rows[:400]:
            print(f"{tag:8s} seed {s}: old [{fmt(a)}] new [{fmt(b)}] {b['head'][-34:]}")
    n_exact_old = sum(1 for s in range(lo, hi) if old[s].get("exact"))
    n_exact_new = sum(1 for s in range(lo, hi) if new[s].get("exact"))
    print(f"seeds {lo}..{hi - 1} noise={noise}: same {same}, new looser on {len(looser)}, new stricter on {len(stricter)}; whole-patch exact: old {n_exact_old}, new {n_exact_new}")
    print("looser seeds:", [s for s, _, _ in looser][:200])


if __name__ == "__main__":
    main()
