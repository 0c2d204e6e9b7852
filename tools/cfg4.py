"""BASELINE config 4 timing: 2-op FM patch with a feedback edge, 65 536 voices, 1 s @ 48 kHz (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, srack_pkg
S = srack_pkg.load()
V, T = 65536, 48000
for B, flags, per_voice in ((1, 0, True), (1024, 0, True), (1, 0, False)):
    p = S.Patch(48000, B, 2)
    ids = S.build_p2(p)
    p.configure_voices(V)
    if per_voice:
        beta, index = S.p2_voice_params(V)
        p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta)
        p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
    frames = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), flags, st); torch.cuda.synchronize()
    t = time.perf_counter()
    p.render_raw(T, frames.data_ptr(), mix.data_ptr(), flags, st); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"B={B} per_voice={per_voice}: {dt*1e3:.1f} ms  {V*T/dt/1e9:.1f} G voice-samples/s  {p.info()}")
