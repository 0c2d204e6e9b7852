"""Where does a fuzzed patch's default-mode error start?  Wires module outputs, one after the other, to the output module's channel 1 (beside
the patch's own wiring) and compares each with the oracle.  usage: python tools/dbg_probe.py [noise] seed out_module mod.port ... (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load(); O.build()
args = sys.argv[1:]
NOISE = args[0] == "noise"
if NOISE: args = args[1:]
seed, out = int(args[0]), int(args[1])
flags = int(os.environ.get("DBG_FLAGS", "0"))
for probe in args[2:]:
    m, port = map(int, probe.split("."))
    B, build, overrides = random_patch(seed, NOISE)
    V, T = (67, 1300) if B < 1024 else (131, 2300)
    if os.environ.get("SOAK_VT"):
        V, T = (int(x) for x in os.environ["SOAK_VT"].split(","))
    o = O.OraclePatch(48000, B, 2)
    ids = build(o)
    o.connect(m, port, out, 1)  # (handles are module indices)
    ov = [(ids[mm], f, fn(V)) for mm, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    p = S.Patch(48000, B, 2)
    ids2 = build(p)
    p.connect(m, port, out, 1)
    p.configure_voices(V)
    for mm, f, vals in ov: p.set_voice_field(mm, f, vals)
    fr = p.render_channels(T, flags)
    d = np.abs(fr.astype(np.float64) - ref.astype(np.float64))[1]  # [t, v]
    line = f"probe {probe}: max abs {d.max():.3e}"
    for v in [int(x) for x in os.environ.get("DBG_V", "").split(",") if x]:
        nz = np.flatnonzero(d[:, v] > float(os.environ.get("DBG_EPS", "0")))
        line += f" | v{v}: first t {nz[0] if len(nz) else None} ({len(nz)} samples, max {d[:, v].max():.2e})"
        if len(nz):
            t0 = nz[0]
            line += f" ref {ref[1, max(t0-1,0):t0+3, v]} gpu {fr[1, max(t0-1,0):t0+3, v]}"
    print(line)
