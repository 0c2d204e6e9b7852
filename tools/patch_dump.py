"""The wiring of a fuzzed patch (CPU only).  usage: python tools/patch_dump.py [noise] seed..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.fuzz_patches import random_patch
NAMES = {0: "OUT", 1: "OSC", 2: "VCF", 3: "ADSR", 4: "VCA", 5: "MIX", 6: "MATH", 7: "GRID", 8: "PAT", 9: "NL", 10: "SMP", 11: "NOISE", 12: "VERB"}


class Rec:
    def __init__(self): self.mods, self.wires, self.fields = [], [], {}
    def add_module(self, t, *a, **k): self.mods.append(int(t)); return len(self.mods) - 1
    def connect(self, a, ap, b, bp): self.wires.append((a, ap, b, bp))
    def set_field(self, m, f, v): self.fields.setdefault(m, {})[f] = v
    def __getattr__(self, name):
        def f(*a, **k): print("   (", name, a[:3], ")")
        return f


args = sys.argv[1:]
noise = bool(args) and args[0] == "noise"
if noise: args = args[1:]
for seed in map(int, args):
    B, build, overrides = random_patch(seed, noise)
    r = Rec(); build(r)
    print(f"seed {seed} B {B}")
    for i, t in enumerate(r.mods):
        print(f"  {i}: {NAMES.get(t, t)} {({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.fields.get(i, {}).items()})}")
    for a, ap, b, bp in r.wires:
        print(f"    {NAMES.get(r.mods[a])}{a}.{ap} -> {NAMES.get(r.mods[b])}{b}.{bp}")
    print("  per voice:", [(m, f) for m, f, _ in overrides])
