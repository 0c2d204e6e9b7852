"""tools/disasm_jit.py <p1|poly|p2|p2_b1024|p3|p4> [flags] [out.s] — the kernel specialised for a BASELINE patch (hiprtc cross-compiles for gfx950 without a
GPU), disassembled, with an instruction histogram per loop: the ISA counts quoted for the general path.  The code object is
taken out of a scratch disk cache (jit.cpp: 24 bytes of header, then the ELF)."""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what, flags = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/srack_jit.s"
tmp = tempfile.mkdtemp()
os.environ["SRACK_KERNEL_CACHE_DIR"] = tmp
import numpy as np
import srack_pkg
S = srack_pkg.load()
V = 128
p = S.Patch(48000, 1 if what == "p2" else 1024, 2)
if what == "p1":
    ids = S.build_p1(p); p.configure_voices(V); det, cut = S.p1_voice_params(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det); p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
elif what == "poly":
    ids = S.build_p1(p); p.configure_voices(V)
    for m, f, v in S.p1_poly_overrides(ids, S.p1_poly_voice_params(V)): p.set_voice_field(m, f, v)
elif what in ("p2", "p2_b1024"):
    ids = S.build_p2(p); p.configure_voices(V); beta, index = S.p2_voice_params(V)
    p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta); p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
elif what == "p3":
    ids = S.build_p3(p); p.configure_voices(V)
    p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, np.linspace(-2, 0.5, V).astype(np.float32)); p.set_voice_field(ids["vcf"], S.VCF_FREQ, np.linspace(0.05, 0.4, V).astype(np.float32))
else:
    ids = S.build_p4(p); p.configure_voices(V); depth, expo = S.p4_voice_params(V)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth); p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
p.kernel_compile(flags)
co = [f for f in os.listdir(tmp) if f.endswith(".hsaco")]
assert len(co) == 1, co
raw = open(os.path.join(tmp, co[0]), "rb").read()[24:]
elf = os.path.join(tmp, "k.co")
open(elf, "wb").write(raw)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", elf], capture_output=True, text=True, check=True).stdout
open(out, "w").write(txt)
lines = txt.split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <srk_voice>:$", l))
end = next((i for i in range(start + 1, len(lines)) if lines[i] == ""), len(lines))
body = lines[start + 1:end]
print(lines[start], len(body), "instructions ->", out)
addr = lambda l: int(l.split("//")[1].split(":")[0], 16)
a2i = {addr(l): i for i, l in enumerate(body) if "//" in l}
base = int(lines[start].split()[0], 16)
F64 = re.compile(r"^v_\w+_f64|^v_cvt_f64|^v_cvt_\w+_f64|^v_fract_f64|^v_ldexp_f64|^v_rndne_f64|^v_floor_f64")
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+\d+\s.*<.*\+0x([0-9a-f]+)>", l)
    if not m:
        continue
    j = a2i.get(base + int(m.group(1), 16))
    if j is None or j >= i:
        continue
    ops = collections.Counter(x.split()[0] for x in body[j:i + 1] if x.strip())
    n64 = sum(v for k, v in ops.items() if F64.match(k))
    if i - j > 100:
        print(f"loop [{j}, {i}] {i - j + 1} instructions, {n64} f64-rate: " + " ".join(f"{k}:{v}" for k, v in ops.most_common(14)))
