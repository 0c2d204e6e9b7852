"""Do the kernels the generator writes for random patches COMPILE (no GPU: hiprtc cross-compiles)?  After a change to the flattener's flags or to
jit.cpp: every combination of oscillator / filter flavours the fuzzer's patches draw.  usage: <first> <last> [noise]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRACK_KERNEL_CACHE_DIR"] = tempfile.mkdtemp()
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, skipped, n, t0 = [], 0, 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    for flags in (32, 36):
        p = S.Patch(48000, B, 2)
        ids = build(p)
        p.configure_voices(128)
        for m, f, fn in overrides:
            p.set_voice_field(ids[m], f, fn(128))
        try:
            p.kernel_compile(flags)
            n += 1
        except S.SrackError as e:
            if e.code == S.ERR_UNSUPPORTED:
                skipped += 1
            else:
                bad.append((seed, flags, str(e)[:300]))
print(f"seeds {lo}..{hi - 1} noise={noise}: {n} kernels compiled, {skipped} unsupported (reverb), {len(bad)} FAILED, {time.time() - t0:.0f} s")
for b in bad[:10]:
    print("  seed %d flags %d: %s" % b)
