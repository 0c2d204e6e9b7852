"""Latency of single control-program modules (diagnostic): identical voices, so the whole patch is one control unit."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, srack_pkg
S = srack_pkg.load()
V, T = 128, 48000

def run(name, build):
    p = S.Patch(48000, 1024, 2)
    build(p)
    p.configure_voices(V)
    n_planes, _ = p.planes()
    frames = torch.empty((n_planes, T, V), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        p.render_raw(T, frames.data_ptr(), 0, 0, st); torch.cuda.synchronize()
    t = time.perf_counter()
    p.render_raw(T, frames.data_ptr(), 0, 0, st); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name:28s} {dt*1e3:7.2f} ms  = {dt/T*1e6:.3f} us/sample   {p.info()[:150]}")

def clock_only(p):
    c, o = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_OUTPUT)
    p.set_field(c, S.OSC_VAL, -4.0); p.connect(c, 1, o, 0)
def clock_grid(p):
    c, g, o = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_GRID_SEQUENCER), p.add_module(S.MOD_OUTPUT)
    p.set_field(c, S.OSC_VAL, -4.0); p.set_field(g, S.GRIDSEQ_LENGTH, 8)
    for i in range(8): p.set_step(g, 0, i, 1, i)
    p.connect(c, 1, g, 0); p.connect(g, 0, o, 0); p.connect(g, 1, o, 1)
def clock_pat(p):
    c, g, o = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_PATTERN_SEQUENCER), p.add_module(S.MOD_OUTPUT)
    p.set_field(c, S.OSC_VAL, -4.0); p.set_field(g, S.PATSEQ_LENGTH, 8)
    for i in range(8): p.set_step(g, 1, i, 1); p.set_step(g, 5, i, 2 if i == 2 else 0)
    p.connect(c, 1, g, 0); p.connect(g, 1, o, 0); p.connect(g, 5, o, 1)
def clock_adsr(p):
    c, a, o = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_ADSR), p.add_module(S.MOD_OUTPUT)
    p.set_field(c, S.OSC_VAL, -4.0)
    for f, v in zip((S.ADSR_A_SEC, S.ADSR_D_SEC, S.ADSR_S_VAL, S.ADSR_R_SEC), (0.002, 0.02, 0.6, 0.01)): p.set_field(a, f, v)
    p.connect(c, 1, a, 0); p.connect(a, 0, o, 0)
def clock_math(p):
    c, a, o = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_MATH), p.add_module(S.MOD_OUTPUT)
    p.set_field(c, S.OSC_VAL, -4.0); p.set_field(a, S.MATH_CONSTANT, 0.5)
    p.connect(c, 1, a, 0); p.connect(a, 0, o, 0)
for name, b in (("clock", clock_only), ("clock+math", clock_math), ("clock+grid", clock_grid), ("clock+pattern", clock_pat), ("clock+adsr", clock_adsr)):
    for flags_env in ("",):
        run(name, b)
