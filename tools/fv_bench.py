"""Reverb patch at scale (diagnostic): per-voice detuned saw -> FreeverbModule -> stereo out; also a noise -> filter voice."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, srack_pkg
S = srack_pkg.load()
V, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 48000

def run(name, build):
    p = S.Patch(48000, 1024, 2)
    build(p)
    n_planes, _ = p.planes()
    frames = torch.empty((n_planes, T, V), dtype=torch.float32, device="cuda")
    mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), 0, st); torch.cuda.synchronize()
    t = time.perf_counter()
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), 0, st); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: {V} voices  {dt*1e3:.1f} ms/step  {V*T/dt/1e9:.2f} G voice-samples/s  {p.info()[:160]}", flush=True)

def reverb(p):
    osc, fv, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_FREEVERB), p.add_module(S.MOD_OUTPUT)
    p.connect(osc, S.OSC_OUT_SAW, fv, 0); p.connect(fv, 0, out, 0); p.connect(fv, 1, out, 1)
    p.configure_voices(V)
    p.set_voice_field(osc, S.OSC_VAL, np.linspace(-2, 1, V).astype(np.float32))

def noise(p):
    nz, vcf, out = p.add_module(S.MOD_NOISE), p.add_module(S.MOD_MOOG_FILTER), p.add_module(S.MOD_OUTPUT)
    p.connect(nz, 0, vcf, 0); p.connect(vcf, 0, out, 0); p.connect(vcf, 0, out, 1)
    p.configure_voices(V)
    p.set_voice_field(vcf, S.VCF_FREQ, np.linspace(0.05, 0.6, V).astype(np.float32))

run("noise->vcf", noise)
run("saw->freeverb", reverb)
