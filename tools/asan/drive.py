import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
n = 0
for noise in (False, True):
    for seed in range(0, 400 if not noise else 250):
        B, build, overrides = random_patch(seed, noise)
        for V in (1, 70):
            for flags in range(16):
                p = S.Patch(48000, B, 2)
                ids = build(p)
                p.configure_voices(V)
                if flags & 1: p.keep_state(True)     # (the whole plan is flattened, not only what the output hears)
                for m, f, fn in overrides:
                    p.set_voice_field(ids[m], f, fn(V))
                try:
                    p.info() if False else p.reserve(64, True, flags)
                    p.get_voice_field(ids[0], 0)
                    if seed % 5 == 0:
                        p.kernel_source(flags)   # the kernel generator (source only: no compilation)
                    data = p.save_srk()
                    q = S.Patch.load_srk(data, 48000, B, 2)
                    q.configure_voices(V); q.reserve(64, True, flags)
                except S.SrackError as e:
                    pass
                n += 1
print("flattened", n, "programs")
