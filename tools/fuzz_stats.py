import sys, numpy as np, importlib.util
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import srack_pkg
from oracle import oracle as O
from tests import fuzz_patches as fz
S = srack_pkg.load()
tot=0; exact=0; worst=0; nonsilent=0
for seed in range(160):
    B, build, overrides = fz.random_patch(seed)
    V,T = 67, 1300  # as the test
    o = O.OraclePatch(48000, B, 2); ids = build(o)
    ov = [(ids[m], f, fn(V)) for m,f,fn in overrides]
    ref,_ = o.render_batch(V, T, ov, threads=8)
    nonsilent += bool(np.abs(ref).max() > 0.05)
    for flags in (1,3,5,7,9,11):
        p = S.Patch(48000, B, 2); ids2 = build(p); p.configure_voices(V)
        for m,f,vals in ov: p.set_voice_field(m,f,vals)
        fr = p.render_channels(T, flags)
        same = (fr.view(np.uint32)==ref.view(np.uint32))
        tot+=1; exact += bool(same.all())
        if not same.all():
            err = np.abs(fr.astype(np.float64)-ref)/np.maximum(np.abs(ref),1.0)
            worst=max(worst, float((err>1e-5).mean()))
            print("seed",seed,"flags",flags,"neq frac",1-same.mean(),"bad frac",(err>1e-5).mean(), "max",err.max())
print("renders",tot,"bit-exact",exact,"worst bad frac",worst,"non-silent patches",nonsilent)
# default (approximating) modes: how many renders stay within 1e-5 everywhere, and how bad the rest get
tot=ok=0; fr_bad=[]
for seed in range(160):
    B, build, overrides = fz.random_patch(seed)
    V,T = 67, 1300
    o = O.OraclePatch(48000, B, 2); ids = build(o)
    ov = [(ids[m], f, fn(V)) for m,f,fn in overrides]
    ref,_ = o.render_batch(V, T, ov, threads=8)
    for flags in (0,2,4):
        p = S.Patch(48000, B, 2); ids2 = build(p); p.configure_voices(V)
        for m,f,vals in ov: p.set_voice_field(m,f,vals)
        fr = p.render_channels(T, flags)
        err = np.abs(fr.astype(np.float64)-ref)/np.maximum(np.abs(ref),1.0)
        bad = float((err>1e-5).mean()); tot+=1; ok += bad==0
        if bad: fr_bad.append((round(bad,5), seed, flags))
print("default modes: renders",tot,"all samples within 1e-5:",ok,"others (frac off, seed, flags):",sorted(fr_bad)[-12:])
