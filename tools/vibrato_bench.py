"""P1 with a saw-LFO vibrato on the audio oscillator's pitch (a tainted value reaches a pitch input, feed-forward): ms per step at scale.
(--flags 1 gives what the pre-round-2 rule gave: the whole patch in the exact flavour.)  usage: python tools/vibrato_bench.py [voices]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, srack_pkg
S = srack_pkg.load()
V, T = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 48000
p = S.Patch(48000, 1024, 2)
ids = S.build_p1(p)
lfo = p.add_module(S.MOD_OSCILLATOR)
depth = p.add_module(S.MOD_MATH)
p.set_field(lfo, S.OSC_VAL, -6.0)              # 6.9 Hz
p.set_field(depth, S.MATH_OPERATION, S.MATH_MULTIPLY)
p.set_field(depth, S.MATH_CONSTANT, 0.02)
p.connect(lfo, S.OSC_OUT_SAW, depth, 0)
p.connect(depth, 0, ids["osc_a"], 0)
p.configure_voices(V)
det, cut = S.p1_voice_params(V)
p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
frames = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
p.reserve(T, True, 0)
for _ in range(2):
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), 0, st)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), 0, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print(f"{dt * 1e3:.2f} ms/step  {V * T / dt / 1e9:.1f} G voice-samples/s  frac {4 * V * T / dt / 8e12:.3f}  {p.info()}")
