"""Registers / LDS of the kernels specialised for random patches (no GPU needed: hiprtc cross-compiles): which of tools/patch_survey.py's
patches are held below four waves per SIMD, and by what.  usage: <seed> [<seed> ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
os.environ["SRACK_KERNEL_CACHE_DIR"] = tmp
import numpy as np
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
for seed in [int(x) for x in sys.argv[1:]]:
    B, build, overrides = random_patch(seed)
    p = S.Patch(48000, B, 2)
    ids = build(p)
    V = 128
    p.configure_voices(V)
    for m, f, fn in overrides:
        p.set_voice_field(ids[m], f, fn(V))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    try:
        p.kernel_compile(32)
    except S.SrackError as e:
        print(seed, "no specialised kernel:", e)
        continue
    co = [f for f in os.listdir(tmp) if f.endswith(".hsaco")][0]
    raw = open(os.path.join(tmp, co), "rb").read()[24:]
    elf = os.path.join(tmp, "k.co")
    open(elf, "wb").write(raw)
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    g = lambda k: int(re.findall(r"\." + k + r":\s+(\d+)", txt)[0])
    vg, lds, scr = g("vgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")
    by_vgpr = min(8, 512 // max(vg, 1)) if vg <= 128 else (512 // vg)
    by_lds = (160 * 1024 // max(lds, 1)) / 4.0
    print(f"seed {seed:3d} B={B:4d}: vgpr {vg:3d} lds {lds:6d} scratch {scr:4d}  -> waves per SIMD: {by_vgpr} by registers, {by_lds:.1f} by LDS")
