"""Parameter-space soak of round 6's FM kernels — config 4's program as default mode renders it (the modulator exact as a whole): render_fm_pair_x
(buffer_size 1) and render_fm_pair_block_x (256 ... 1024, calls of >= 4096 samples).  Random sample rates, ring lengths (any, not only powers of
two), voice counts, render lengths, a second call of random length (shorter ones fall to the general path: the state carries over), per-voice
feedback gains / indices / pitches of both oscillators / initial phases.  Frames against the oracle at the contract's bar (the carrier keeps the
default forms); the MODULATOR's phase after the first call bit for bit where the call is a whole number of blocks.  usage: <first> <last>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one(seed, S, O):
    """-> (kernel of the first call, NaNs at the oracle's positions, max error, modulator states that differ, program info)"""
    r = np.random.default_rng((seed, 0xF6))
    sr = int(r.choice([8000, 22050, 44100, 48000, 65535]))
    B = 1 if r.random() < 0.4 else int(r.choice([256, 512, 1000, 1024, int(r.integers(256, 1025))]))
    V = int(r.choice([1, 31, 32, 33, 64, 70, 130, 200]))
    blocks = int(r.integers(5, 10))
    T1 = blocks * max(B, 1) if B > 1 else int(r.integers(2000, 6000))
    if B > 1 and T1 < 4096: T1 = ((4096 + B - 1) // B) * B
    if r.random() < 0.3: T1 += int(r.integers(1, 64))      # a ragged last chunk (no state check then)
    T2 = int(r.choice([0, 100, 1024, 4096, 5000]))
    bet = r.uniform(0, 0.7, V).astype(np.float32); idx = r.uniform(0, 2.5, V).astype(np.float32)
    if r.random() < 0.2: bet[int(r.integers(0, V))] = 3.0e4                           # a voice whose 2^cv overflows
    if r.random() < 0.2: idx[int(r.integers(0, V))] = float(r.uniform(3, 12))          # a carrier outside every bounded class
    vm = (r.uniform(-2, 2, V) if r.random() < 0.5 else np.zeros(V)).astype(np.float32)
    vcr = r.uniform(-2, 2, V).astype(np.float32)
    pm, pc = r.uniform(0, 1, V), r.uniform(0, 1, V)
    def build(g): return S.build_p2(g)
    ov_of = lambda ids: [(ids["mul_fb"], S.MATH_CONSTANT, bet), (ids["mul_idx"], S.MATH_CONSTANT, idx), (ids["osc_m"], S.OSC_VAL, vm), (ids["osc_c"], S.OSC_VAL, vcr),
                         (ids["osc_m"], S.OSC_POS, pm), (ids["osc_c"], S.OSC_POS, pc)]
    o = O.OraclePatch(sr, B, 2)
    ids = build(o)
    ov = ov_of(ids)
    ref, _ = o.render_batch(V, T1 + T2, ov, threads=8)
    p = S.Patch(sr, B, 2)
    build(p)
    p.configure_voices(V)
    for m, f, vals in ov: p.set_voice_field(m, f, vals)
    a = p.render_channels(T1, 0)
    k = p.info().split("kernel=")[-1]
    state_bad = 0
    if T1 % B == 0:
        pos = p.get_voice_field(ids["osc_m"], S.OSC_POS)
        for v in sorted(set([0, V - 1, V // 2])):
            q = O.OraclePatch(sr, B, 2)
            build(q)
            for m, f, vals in ov: q.set_field(m, f, float(vals[v]))
            q.render(T1)
            w = q.get_field(ids["osc_m"], S.OSC_POS)
            if not (np.float64(pos[v]).view(np.uint64) == np.float64(w).view(np.uint64) or (np.isnan(pos[v]) and np.isnan(w))): state_bad += 1
    out = a[0] if T2 == 0 else np.concatenate([a[0], p.render_channels(T2, 0)[0]])
    nan_ok = bool((np.isnan(out) == np.isnan(ref[0])).all())
    fin = np.isfinite(ref[0]) & np.isfinite(out)
    err = float(np.where(fin, np.abs(out.astype(np.float64) - ref[0]) / np.maximum(np.abs(ref[0]), 1.0), 0.0).max())
    return k, nan_ok, err, state_bad, f"sr {sr} B {B} V {V} T {T1}+{T2}: " + p.info()[-100:]


if __name__ == "__main__":
    import srack_pkg
    from oracle import oracle as O
    S = srack_pkg.load()
    O.build()
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad, n, kernels, worst, t0 = [], 0, {}, 0.0, time.time()
    for seed in range(lo, hi):
        k, nan_ok, err, state_bad, info = one(seed, S, O)
        kernels[k] = kernels.get(k, 0) + 1
        n += 1
        worst = max(worst, err)
        if not nan_ok or err > 1e-5 or state_bad:
            bad.append((seed, k, nan_ok, err, state_bad, info))
    print(f"fm_x soak, seeds {lo}..{hi - 1}: {n} patches, {len(bad)} fail, worst {worst:.2e}, {time.time() - t0:.0f} s")
    print("  kernels of the first call:", sorted(kernels.items()))
    for b in bad[:20]: print("  ", b)
