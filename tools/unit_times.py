"""How long does ONE control unit take per sample?  Voice-invariant patches (4096 identical voices: everything is hoisted into co-scheduled
control units, the voice kernel only broadcasts) built from one module chain each; the render's time per sample is the slowest unit's.
usage: python tools/unit_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srack_pkg
S = srack_pkg.load()
T, V = 48000, 4096

def timed(name, build, flags=0):
    p = S.Patch(48000, 1024, 2)
    build(p)
    p.configure_voices(V)
    p.reserve(T, True, flags)
    for _ in range(2):
        p.render_raw(T, flags=flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        p.render_raw(T, flags=flags)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} flags {flags}: {dt * 1e3:7.3f} ms per second of audio = {dt / T * 1e9:6.1f} ns per sample   [{p.info()[-60:]}]")

OSC, VCF, ADSR, VCA, OUT = S.MOD_OSCILLATOR, S.MOD_MOOG_FILTER, S.MOD_ADSR, S.MOD_VCA, S.MOD_OUTPUT
def chain(mods, wires, fields=()):
    def b(p):
        ids = [p.add_module(m) for m in mods]
        for m, f, v in fields: p.set_field(ids[m], f, v)
        for s, sp, d, dp in wires: p.connect(ids[s], sp, ids[d], dp)
    return b

for flags in (0, 1):
    timed("saw -> out", chain([OSC, OUT], [(0, 2, 1, 0)]), flags)
    timed("square -> out", chain([OSC, OUT], [(0, 1, 1, 0)]), flags)
    timed("sine -> out", chain([OSC, OUT], [(0, 0, 1, 0)]), flags)
    timed("saw -> filter -> out", chain([OSC, VCF, OUT], [(0, 2, 1, 0), (1, 0, 2, 0)]), flags)
    timed("LFO square -> envelope -> out", chain([OSC, ADSR, OUT], [(0, 1, 1, 0), (1, 0, 2, 0)], [(0, S.OSC_VAL, -2.0)]), flags)
    timed("saw, LFO square -> VCA -> out", chain([OSC, OSC, VCA, OUT], [(0, 2, 2, 0), (1, 1, 2, 1), (2, 0, 3, 0)], [(1, S.OSC_VAL, -2.0)]), flags)
    timed("P1 (config 2)", lambda p: S.build_p1(p, lfo_val=-2.0), flags)
