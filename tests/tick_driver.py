"""One scenario of tests/test_gpu_tick.py in a process of its own (the library reads SRACK_TICK once): a patch driven call by call the way
a host's tick loop would (one srack_render per block), with the things a host may do between ticks — read state back, ask for another
length, render one long call, switch between frames and mix, edit a parameter under keep_state.  Everything rendered and read goes to
an .npz; the test compares the .npz of SRACK_TICK=1 with that of SRACK_TICK=0 bit for bit and holds the renders to the oracle.

    python tests/tick_driver.py <scenario> <flags> <out.npz>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import srack_pkg  # noqa: E402


def make(S, g, scenario, V):
    """The scenario's patch on `g` (a product Patch or an OraclePatch) -> (ids, per-voice overrides)."""
    scenario = scenario.replace("_multi", "")   # the same patches driven in calls of several chunks (SCRIPTS below)
    if scenario in ("p1", "p1_wide"):
        ids = S.build_p1(g, adsr="finite", lfo_val=-2.0)   # gate at 110 Hz: an edge every 218 samples
        det, cut = S.p1_voice_params(V)
        return ids, [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)]
    if scenario == "identical":                             # config 2's shape: everything voice-invariant, five control units + a broadcast
        ids = S.build_p1(g, adsr="finite", lfo_val=-2.0)
        return ids, []
    if scenario in ("p3", "keep"):
        ids = S.build_p3(g, clock_val=-2.0)
        u0, u1 = S.voice_uniform(V, 0), S.voice_uniform(V, 1)
        return ids, [(ids["transpose"], S.MATH_CONSTANT, (u0 * np.float32(2.5) - np.float32(2.0)).astype(np.float32)),
                     (ids["vcf"], S.VCF_FREQ, (np.float32(0.05) + u1 * np.float32(0.35)).astype(np.float32))]
    raise SystemExit("unknown scenario " + scenario)


def voices_of(scenario):
    return {"p1": 192, "p1_wide": 4096 + 37, "identical": 96, "p3": 160, "keep": 130}[scenario.replace("_multi", "")]


# what a host does, in order: ("render", n_samples, what) with what in "fm" / "f" / "m"; ("read",) reads state back; ("edit",) a parameter
SCRIPTS = {
    "p1": [("render", 512, "fm")] * 7 + [("read",)] + [("render", 512, "fm")] * 3 + [("render", 300, "fm")] * 3 + [("render", 5000, "fm")]
          + [("render", 512, "m")] * 2 + [("render", 512, "f")] * 2 + [("read",)] + [("render", 4096, "fm")] * 2 + [("render", 1, "fm")] * 3,
    "p1_wide": [("render", 1024, "fm")] * 4 + [("read",)] + [("render", 1024, "fm")] * 2,
    "identical": [("render", 256, "fm")] * 9 + [("read",)] + [("render", 256, "fm")] * 2 + [("render", 1000, "fm")] * 3 + [("render", 6000, "fm")] + [("render", 256, "m")] * 3,
    "p3": [("render", 256, "fm")] * 12 + [("read",)] + [("render", 1024, "fm")] * 3 + [("render", 4500, "fm")] + [("render", 1024, "f")] * 2 + [("read",)],
    # calls of SEVERAL chunks, back to back (a host that renders long blocks; bench.py's steps): a session whose unit is the chunk — two
    # full chunks and a short one per call —, interrupted by a read-back, another length, a one-chunk call, mix only / frames only
    "p1_multi": [("render", 9000, "fm")] * 3 + [("read",)] + [("render", 9000, "fm")] * 2 + [("render", 5000, "fm")] * 2 + [("render", 512, "fm")] * 2
                + [("render", 9000, "m")] * 2 + [("render", 9000, "f")] + [("read",)] + [("render", 8192, "fm")] * 2,
    "identical_multi": [("render", 9000, "fm")] * 3 + [("read",)] + [("render", 9000, "fm")] + [("render", 4097, "fm")] * 2 + [("render", 9000, "m")] * 2,
    "p3_multi": [("render", 10000, "fm")] * 3 + [("read",)] + [("render", 10000, "fm")] + [("render", 300, "fm")] * 2 + [("render", 10000, "f")] * 2 + [("read",)],
    "keep_multi": [("render", 6000, "fm")] * 3 + [("edit",)] + [("render", 6000, "fm")] * 3 + [("read",)] + [("render", 6000, "fm")],
    "keep": [("render", 512, "fm")] * 5 + [("edit",)] + [("render", 512, "fm")] * 5 + [("read",)] + [("render", 512, "fm")] * 2,
}


def main():
    scenario, flags, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    S = srack_pkg.load()
    V = voices_of(scenario)
    p = S.Patch(48000, 1024, 2)
    ids, over = make(S, p, scenario, V)
    if scenario.startswith("keep"):
        p.keep_state(True)
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    out = {}
    infos = []
    n = 0
    for step, op in enumerate(SCRIPTS[scenario]):
        if op[0] == "render":
            fr, mx = p.render(op[1], frames="f" in op[2], mix="m" in op[2], flags=flags)
            if fr is not None:
                out[f"fr{step}"] = fr
            if mx is not None:
                out[f"mx{step}"] = mx
            n += op[1]
            infos.append(p.info())
        elif op[0] == "read":
            # module state as the patch holds it after the last rendered sample: a control-program module through get_field (one voice), a
            # voice-program module per voice
            for name, m in ids.items():
                t = p.module_type(m)
                if t == S.MOD_OSCILLATOR:
                    out[f"rd{step}_{name}_pos"] = p.get_voice_field(m, S.OSC_POS)
                elif t == S.MOD_ADSR:
                    out[f"rd{step}_{name}_phase"] = p.get_voice_field(m, S.ADSR_PHASE)
                    out[f"rd{step}_{name}_mode"] = p.get_voice_field(m, S.ADSR_MODE)
                elif t == S.MOD_MOOG_FILTER:
                    out[f"rd{step}_{name}_b4"] = p.get_voice_field(m, S.VCF_ST_B4)
                elif t == S.MOD_GRID_SEQUENCER:
                    out[f"rd{step}_{name}_step"] = p.get_voice_field(m, S.GRIDSEQ_CURRENT_STEP)
                elif t == S.MOD_PATTERN_SEQUENCER:
                    out[f"rd{step}_{name}_step"] = p.get_voice_field(m, S.PATSEQ_CURRENT_STEP)
        elif op[0] == "edit":
            p.set_field(ids["vcf"], S.VCF_RES, 0.7)  # a slider moved under the running graph (keep_state: module state carries over)
    out["total"] = np.array(n)
    out["infos"] = np.array(infos)
    np.savez(out_path, **out)


if __name__ == "__main__":
    main()
