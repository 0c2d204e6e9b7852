"""One-off soak of the differential fuzzer beyond the seeds the test suite pins: exact modes must be bit-identical to the oracle.
usage: python tools/fuzz_soak.py <first_seed> <last_seed> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
O.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, t0 = [], 0, time.time()
for seed in range(lo, hi):
    os.environ["SRACK_WANT_WAVES"] = "1" if seed % 2 else "0"
    B, build, overrides = random_patch(seed, noise)
    V, T = (67, 1300) if B < 1024 else (131, 2300)
    if os.environ.get("SOAK_VT"):  # e.g. SOAK_VT=16,48000: a full second of fewer voices
        V, T = (int(x) for x in os.environ["SOAK_VT"].split(","))
    o = O.OraclePatch(48000, B, 2)
    ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    # FUZZ_SPECIAL=1: only the kernels specialised at run time (hoisted / pipelined, one control unit, everything per voice)
    for flags in ((35, 43, 39) if os.environ.get("FUZZ_SPECIAL") else (1, 3, 5, 7, 9, 11)):
        p = S.Patch(48000, B, 2)
        build(p)
        p.configure_voices(V)
        if os.environ.get("FUZZ_KEEP"): p.keep_state(True)   # (every planned module evaluated, not only what the output hears)
        for m, f, vals in ov:
            p.set_voice_field(m, f, vals)
        if flags & 32:
            try:
                p.kernel_source(flags)
            except S.SrackError:
                continue  # a reverb: the interpreter's
        fr = p.render_channels(T, flags)
        n += 1
        # NaNs compare as NaNs: x86's default NaN has the sign bit set (0xffc00000), the GPU's has not (0x7fc00000)
        same = (fr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(fr) & np.isnan(ref))
        if not same.all():
            bad.append((seed, flags, float(1 - same.mean())))
print(f"seeds {lo}..{hi - 1} noise={noise}: {n} renders, {len(bad)} not bit-identical, {time.time() - t0:.0f} s")
for b in bad[:40]:
    print("  seed %d flags %d: %.5f of the samples differ" % b)
if os.environ.get("SOAK_JSON"):  # (tools/soak_par.py merges its workers' results)
    import json
    with open(os.environ["SOAK_JSON"], "w") as f:
        json.dump(dict(first=lo, last=hi, noise=noise, renders=n, bad=len(bad), seconds=time.time() - t0, vt=os.environ.get("SOAK_VT", ""),
                       special=bool(os.environ.get("FUZZ_SPECIAL")), worst=[[b[0], b[1], b[2], b[2], True] for b in bad]), f)
