"""Default-mode error of the GPU path against the golden vectors (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
S = srack_pkg.load()
G = "tests/golden"
for adsr in ("default","finite"):
    gold = np.load(os.path.join(G, f"cfg1_p1_{adsr}.npz"))["audio"]
    for flags in (0,2,4,6):
        p = S.Patch(48000,1024,2); S.build_p1(p, adsr=adsr); p.configure_voices(1)
        out = p.render_channels(48000, flags)[0,:,0].astype(np.float64)
        err = np.abs(out-gold)/np.maximum(np.abs(gold),1.0)
        print(adsr, flags, "max rel err %.3e" % err.max(), "rms %.3e" % np.sqrt((err**2).mean()))
z = np.load(os.path.join(G,"cfg3_p1_voices8.npz")); gold=z["audio"]
p = S.Patch(48000,1024,2); ids=S.build_p1(p, lfo_val=float(z["lfo_val"])); p.configure_voices(8)
p.set_voice_field(ids["osc_a"], S.OSC_VAL, z["detune"]); p.set_voice_field(ids["vcf"], S.VCF_FREQ, z["cutoff"])
out = p.render_channels(gold.shape[0], 0)[0].astype(np.float64)
err = np.abs(out-gold)/np.maximum(np.abs(gold),1.0); print("cfg3 voices8 max rel err %.3e"%err.max())
