"""Parameter-space soak of the FUSED patch shapes (the kernels the benchmarks run): P1 (voice chain, hoisted and per-voice), P2 (FM pair,
z^-1 and ring), P3 (sequencer-driven chain).  Random module parameters, per-voice overrides, sample rates and buffer sizes; exact
modes bit for bit against the oracle, default modes within 1e-5 (P2 / P3: the flattener may force exact).  usage: <first> <last>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
S = srack_pkg.load()
O.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, n, kernels, t0 = [], 0, {}, time.time()

extra = {}
def check(tag, seed, sr, B, V, T, build, ov_of, modes):
    global n
    o = O.OraclePatch(sr, B, 2)
    ids = build(o)
    ov = ov_of(ids)
    ref, ref_mix = o.render_batch(V, T, ov, mix=True, threads=8)
    for flags in modes:
        p = S.Patch(sr, B, 2)
        build(p)
        p.configure_voices(V)
        for m, f, vals in ov: p.set_voice_field(m, f, vals)
        fr, mix = p.render(T, flags=flags)
        k = p.info().split("kernel=")[-1]
        kernels[(tag, flags, k)] = kernels.get((tag, flags, k), 0) + 1
        n += 1
        same = (fr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(fr) & np.isnan(ref))
        err = float(np.nanmax(np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0))) if np.isfinite(ref).any() else 0.0
        scale = np.abs(ref.astype(np.float64)).sum(axis=2)
        mix_ok = bool((np.abs(mix - ref_mix) <= 2e-5 * np.maximum(scale, 1.0)).all()) if np.isfinite(ref_mix).all() else True
        ok = (same.all() if flags & 1 else err <= 1e-5) and mix_ok
        if not ok:
            e_v = np.nanmax(np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0), axis=(0, 1))
            bad.append((tag, seed, flags, sr, B, V, T, k, float(1 - same.mean()), err, mix_ok, int((e_v > 1e-5).sum()), extra.get(tag)))

for seed in range(lo, hi):
    r = np.random.default_rng((seed, 0x5A))
    sr = int(r.choice([8000, 22050, 44100, 48000, 65535]))
    B = int(r.choice([1, 16, 64, 1024]))
    V = int(r.choice([1, 64, 70, 130]))
    T = int(r.choice([700, 2500, 5000]))
    # ---- P1
    lfo = float(np.float32(r.uniform(-9, -1)))
    ad = [float(np.float32(x)) for x in (r.choice([0.0, 0.001, 0.02]), r.uniform(0.001, 0.3), r.uniform(0, 1), r.uniform(0.001, 0.5))]
    res, expa, neg = float(np.float32(r.uniform(0, 1))), float(np.float32(r.uniform(0, 1))), int(r.random() < 0.3)
    port_a, port_f = int(r.choice([S.OSC_OUT_SAW, S.OSC_OUT_SQUARE, S.OSC_OUT_SINE], p=[.6, .3, .1])), int(r.integers(0, 3))
    def p1(g):
        ids = S.build_p1(g, lfo_val=lfo)
        for f, v in zip((S.ADSR_A_SEC, S.ADSR_D_SEC, S.ADSR_S_VAL, S.ADSR_R_SEC), ad): g.set_field(ids["adsr"], f, v)
        g.set_field(ids["vcf"], S.VCF_RES, res); g.set_field(ids["vcf"], S.VCF_EXP_AMT, expa); g.set_field(ids["vca"], S.VCA_NEGATIVE, neg)
        if port_a != S.OSC_OUT_SAW or port_f != 0:
            g.disconnect(ids["vcf"], 0); g.connect(ids["osc_a"], port_a, ids["vcf"], 0)
            g.disconnect(ids["vca"], 0); g.connect(ids["vcf"], port_f, ids["vca"], 0)
        return ids
    det = r.uniform(-3, 2.5, V).astype(np.float32); cut = r.uniform(0.02, 0.85, V).astype(np.float32)
    extra["p1"] = dict(res=round(res, 3), exp=round(expa, 3), neg=neg, port_a=port_a, port_f=port_f, adsr=[round(x, 4) for x in ad], lfo=round(lfo, 2))
    check("p1", seed, sr, B, V, T, p1, lambda ids: [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)], (0, 1, 4, 5, 2))
    # ---- P2
    B2 = int(r.choice([1, 32, 64, 256, 512, 1024]))  # (256 ... 1024, default mode: the time-parallel pair)
    beta0, index0 = float(np.float32(r.uniform(0, 0.6))), float(np.float32(r.uniform(0, 2)))
    def p2(g): return S.build_p2(g, beta=beta0, index=index0)
    bet = r.uniform(0, 0.6, V).astype(np.float32); idx = r.uniform(0, 2, V).astype(np.float32); pitch = r.uniform(-2, 2, V).astype(np.float32)
    check("p2", seed, sr, B2, V, min(T, 2500), p2, lambda ids: [(ids["mul_fb"], S.MATH_CONSTANT, bet), (ids["mul_idx"], S.MATH_CONSTANT, idx), (ids["osc_c"], S.OSC_VAL, pitch)], (0, 1, 2, 34))  # 34: the z^-1 pair through the specialised kernel (the wave's vote over this parameter space)
    # ---- P3
    clock, length = float(np.float32(r.uniform(-7, -1.5))), int(r.integers(1, 17))
    def p3(g):
        ids = S.build_p3(g, clock_val=clock, length=length)
        return ids
    tr = (r.uniform(0, 1, V) * 2.5 - 2.0).astype(np.float32); cut3 = r.uniform(0.05, 0.4, V).astype(np.float32)
    check("p3", seed, sr, B, V, T, p3, lambda ids: [(ids["transpose"], S.MATH_CONSTANT, tr), (ids["vcf"], S.VCF_FREQ, cut3)], (0, 1, 2, 8))
print(f"shape soak, seeds {lo}..{hi - 1}: {n} renders, {len(bad)} fail, {time.time() - t0:.0f} s")
print("  kernels:", sorted((k, v) for k, v in kernels.items()))
for b in bad[:30]: print("  ", b)
