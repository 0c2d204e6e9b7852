"""Sample-player patch P4 at scale (diagnostic): per-voice vibrato depth + shaper exponent, clock and LFO voice-invariant."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, srack_pkg
S = srack_pkg.load()
V, T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 48000
for flags in (0, 4):
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p4(p)
    p.configure_voices(V)
    depth, expo = S.p4_voice_params(V)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth)
    p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
    n_planes, _ = p.planes()
    frames = torch.empty((n_planes, T, V), dtype=torch.float32, device="cuda")
    mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), flags, st); torch.cuda.synchronize()
    t = time.perf_counter()
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), flags, st); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"flags={flags}: {dt*1e3:.1f} ms/step  {V*T/dt/1e9:.1f} G voice-samples/s  {p.info()[:200]}")
