"""Interpreter cost by op (diagnostic): growing prefixes of patch P1's voice chain through render_interp (flags 2), 262 144 voices."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, srack_pkg
S = srack_pkg.load()
V, T = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 48000

def run(name, build, flags=2, frames_on=True, mix_on=True):
    p = S.Patch(48000, 1024, 2)
    build(p)
    n_planes, _ = p.planes()
    frames = torch.empty((n_planes, T, V), dtype=torch.float32, device="cuda") if frames_on else None
    mix = torch.empty((2, T), dtype=torch.float32, device="cuda") if mix_on else None
    st = torch.cuda.current_stream().cuda_stream
    args = (T, frames.data_ptr() if frames_on else None, mix.data_ptr() if mix_on else None, flags, st)
    p.render_raw(*args); torch.cuda.synchronize()
    t = time.perf_counter()
    p.render_raw(*args); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name:34s} {dt*1e3:7.1f} ms/step   {p.info()[:110]}", flush=True)

det = lambda: np.linspace(-2, 1, V).astype(np.float32)
def osc_only(p):
    o, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_OUTPUT)
    p.connect(o, S.OSC_OUT_SAW, out, 0); p.connect(o, S.OSC_OUT_SAW, out, 1)
    p.configure_voices(V); p.set_voice_field(o, S.OSC_VAL, det())
def osc_vcf(p):
    o, f, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_MOOG_FILTER), p.add_module(S.MOD_OUTPUT)
    p.connect(o, S.OSC_OUT_SAW, f, 0); p.connect(f, 0, out, 0); p.connect(f, 0, out, 1)
    p.configure_voices(V); p.set_voice_field(o, S.OSC_VAL, det())
def p1(p):
    ids = S.build_p1(p)
    p.configure_voices(V)
    d, c = S.p1_voice_params(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, d); p.set_voice_field(ids["vcf"], S.VCF_FREQ, c)

run("osc -> out", osc_only)
run("osc -> out (no frames)", osc_only, frames_on=False)
run("osc -> out (no mix)", osc_only, mix_on=False)
run("osc -> vcf -> out", osc_vcf)
run("P1 (osc -> vcf -> vca <- track)", p1)
run("P1 fused for reference", p1, flags=0)
