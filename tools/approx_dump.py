"""csrc/approx.cpp's analysis of a fuzz patch (CPU only, through tests/cpp/approx_probe): decisions, magnitudes and gains per module.
usage: [FUZZ_MORE_OV=1] approx_dump.py <seed> [voices] [noise]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fuzz_patches import random_patch
from tests.test_approx import Rec
noise = "noise" in sys.argv[2:]
args = [a for a in sys.argv[1:] if a != "noise"]
seed = int(args[0]); V = int(args[1]) if len(args) > 1 else 200
B, build, ov = random_patch(seed, noise)
g = Rec(48000, B, 2)
g.set_noise_seed = lambda *a: None
ids = build(g)
for m, f, fn in ov:
    g.override(ids[m], f, fn(V))
probe = os.path.join(ROOT, "tests", "cpp", "approx_probe")
r = g.run(probe)
names = {0: "OUT", 1: "OSC", 2: "VCF", 3: "ADSR", 4: "VCA", 5: "MIX", 6: "MATH", 7: "GRID", 8: "PAT", 9: "NONLIN", 10: "SMP", 11: "NOISE", 12: "VERB"}
types = [int(l.split()[1]) for l in g.lines if l.startswith("mod ")]
for l in g.lines:
    if os.environ.get("DUMP_GRAPH"): print("   ", l[:160])
print("exact_patch", r["exact_patch"], r["why"], "bound %.2e" % r["bound"])
for m, t in enumerate(types):
    if not r["live"][m]: continue
    print(f"{m:2d} {names[t]:5s} osc_exact {r['osc_exact'][m]} blep {r['exact_blep'][m]} literal {r['literal'][m]} sineL {r['sine_loose'][m]} fixed {r['saw_fixed'][m]}  mag {['%.3g' % x for x in r['mag'][m]]}  gain {['%.3g' % x for x in r['gain'][m]]}")
