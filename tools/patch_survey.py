"""Throughput of random patches at scale (default mode, the kernel the library picks): where a patch shape falls far below its neighbours
there is usually something like P4's two mix tiles behind it.  usage: <first seed> <last seed> [voices] [samples]
Per seed: ms per second of audio, voice-samples/s, planes, ops of the voice program, the kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
V = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
T = int(sys.argv[4]) if len(sys.argv) > 4 else 24000
FLAGS = [int(x) for x in os.environ.get("SURVEY_FLAGS", "0").split(",")]   # e.g. SURVEY_FLAGS=0,4: with and without uniform hoisting
rows = []
SEEDS = [int(x) for x in os.environ["SURVEY_SEEDS"].split(",")] if os.environ.get("SURVEY_SEEDS") else range(lo, hi)   # (a list instead of the range: A/B runs)
for seed in SEEDS:
    B, build, overrides = random_patch(seed)
    p = S.Patch(48000, B, 2)
    ids = build(p)
    p.configure_voices(V)
    for m, f, fn in overrides:
        p.set_voice_field(ids[m], f, fn(V))
    n_planes, _ = p.planes()
    if n_planes == 0:
        continue
    fr = torch.empty((n_planes, T, V), dtype=torch.float32, device="cuda")
    mx = torch.empty((2, T), dtype=torch.float32, device="cuda")
    times, info = [], ""
    try:
        for fl in FLAGS:
            p.render_raw(T, fr.data_ptr(), mx.data_ptr(), fl, None)
            torch.cuda.synchronize()
            t = time.perf_counter()
            p.render_raw(T, fr.data_ptr(), mx.data_ptr(), fl, None)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t) * 48000 / T * 1e3)
            if not info:
                info = p.info()
    except S.SrackError as e:
        print(seed, "error", e)
        continue
    rows.append((times[0], seed, B, n_planes, info, times))
    del fr, mx
    torch.cuda.empty_cache()
for ms, seed, B, n_planes, info, times in sorted(rows):
    vs = V * 48000 / (ms * 1e-3)
    other = "  ".join(f"flags {fl}: {t:7.2f}" for fl, t in zip(FLAGS[1:], times[1:]))
    n_ctl = info.count("ctl[")
    print(f"seed {seed:4d} B={B:4d} planes={n_planes}  {ms:8.2f} ms/s  {vs:9.3e} voice-samples/s  {4 * n_planes * vs / 8e12:5.3f} of HBM  {other}  ctl units {n_ctl}  {info[:info.index(']') + 1]} ... {info[info.rindex('kernel='):]}")
