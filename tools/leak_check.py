"""Device memory before / after a few hundred edit + render cycles (with and without keep_state, with rings and a reverb)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # first: the library and torch must share one HIP runtime initialisation
import numpy as np, srack_pkg
S = srack_pkg.load()
torch.cuda.init()
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
for keep in (False, True):
    p = S.Patch(48000, 64, 2)
    ids = S.build_p2(p, beta=0.25, index=0.8)
    fv = p.add_module(S.MOD_FREEVERB)
    p.disconnect(ids["out"], 1); p.connect(ids["osc_c"], S.OSC_OUT_SINE, fv, 0); p.connect(fv, 1, ids["out"], 1)
    p.configure_voices(4096)
    p.keep_state(keep)
    p.render_channels(256, 0)
    f0 = free_mb()
    for i in range(300):
        p.set_field(fv, S.FREEVERB_DRY, (i % 7) / 7.0)
        p.render_channels(256, i % 4)
    f1 = free_mb()
    del p
    f2 = free_mb()
    print(f"keep_state={keep}: free before {f0:.0f} MiB, after 300 cycles {f1:.0f} MiB (delta {f0 - f1:+.0f}), after destroy {f2:.0f} MiB")
