import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, srack_pkg
from oracle import oracle as O
S = srack_pkg.load(); O.build()
V, T = int(sys.argv[1]), int(sys.argv[2])
det, cut = S.p1_voice_params(V)
p = S.Patch(48000, 1024, 2); ids = S.build_p1(p, adsr="finite", lfo_val=-2.0); p.configure_voices(V)
p.set_voice_field(ids["osc_a"], S.OSC_VAL, det); p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
d_fr, d_mx = C.c_void_p(), C.c_void_p()
assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
p.render_raw(T, d_fr, d_mx, 0, None); assert S.lib.srack_device_sync(None) == 0
print(p.info()[-60:])
pick = np.unique(np.concatenate([np.arange(0, V, V // 37 + 1), [0, 63, 64, V - 65, V - 64, V - 1]]))
row = np.empty(V, dtype=np.float32); got = np.empty((T, len(pick)), dtype=np.float32); own = np.empty(T); sc = np.empty(T)
for t in range(T):
    if t % 16 and t not in (T - 1,): continue
    assert S.lib.srack_device_to_host(row.ctypes.data_as(C.c_void_p), C.c_void_p(d_fr.value + t * V * 4), V * 4, None) == 0
    got[t] = row[pick]; own[t] = row.sum(dtype=np.float64); sc[t] = np.abs(row).sum(dtype=np.float64)
mix = np.empty((2, T), dtype=np.float32); S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None)
o = O.OraclePatch(48000, 1024, 2); S.build_p1(o, adsr="finite", lfo_val=-2.0)
ref, _ = o.render_batch(len(pick), T, [(ids["osc_a"], S.OSC_VAL, det[pick]), (ids["vcf"], S.VCF_FREQ, cut[pick])], threads=8)
ts = np.array([t for t in range(T) if t % 16 == 0 or t == T - 1])
err = np.abs(got[ts].astype(np.float64) - ref[0][ts]) / np.maximum(np.abs(ref[0][ts]), 1.0)
print("voices", V, "samples", T, "max rel err", err.max(), "mix ok", bool((np.abs(mix[0][ts] - own[ts]) <= 1e-5 * np.maximum(sc[ts], 1)).all()), "peak", np.abs(ref).max())
