"""tests/cpp/forms_emu.c from Python: the oracle with the GPU default mode's cheaper forms in single modules (CPU only; TEST INFRASTRUCTURE).

EmuPatch is an OraclePatch on a library of its own (srack_oracle.c compiled together with the emulation); `apply_plan` sets, module by
module, the forms csrc/approx.cpp's analysis chose for a patch (tests/cpp/approx_probe's JSON) — or every form there is (`everything=True`:
what a flattener without the bound would render)."""
import ctypes as C
import os
import subprocess

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(ROOT, "tests", "cpp", "forms_emu.c")
_LIB = os.path.join(ROOT, "tests", "cpp", "libforms_emu.so")
OSC_F32_BLEP, OSC_SINE_LOOSE, OSC_SINE_FAST, OSC_ONE_PORT, OSC_FIXED, VCF_CONTRACTED, NONLIN_LOOSE = 1, 2, 4, 8, 16, 1, 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [_SRC, os.path.join(ROOT, "oracle", "srack_oracle.c"), os.path.join(ROOT, "include", "srack_hip.h")]
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(d) for d in deps):
            tmp = f"{_LIB}.{os.getpid()}.tmp"
            subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-unsafe-math-optimizations", "-shared", "-o", tmp, _SRC,
                            "-lm", "-lpthread"], check=True)
            os.replace(tmp, _LIB)
        L = oracle.bind(C.CDLL(_LIB))
        L.emu_set_forms.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        _lib = L
    return _lib


class EmuPatch(oracle.OraclePatch):
    def __init__(self, sample_rate=48000, buffer_size=1024, channels=2, _handle=None, _lib=None):
        super().__init__(sample_rate, buffer_size, channels, _handle=_handle, _lib=_lib or lib())

    def set_forms(self, module, forms):
        if self.L.emu_set_forms(self.h, module, forms) < 0:
            raise ValueError(f"emu_set_forms({module}, {forms}) failed")

    def apply_plan(self, types, plan=None, everything=False, per_voice=None):
        """types: module types in creation order; plan: approx_probe's result; per_voice: the modules with per-voice overrides, or None for
        "every module is evaluated per voice" (render flag 4) — a constant-pitch oscillator WITHOUT one is voice-invariant, hoisted into
        the control program, and keeps its f64 phase there (flatten.cpp: OSC_FIXED_PHASE is a voice program's).  -> the forms set, {module: word}"""
        out = {}
        for m, t in enumerate(types):
            if plan is not None and (not plan["live"][m] or plan["exact_patch"]):
                continue
            w = 0
            if t == oracle.MOD_OSCILLATOR:
                one_port = plan is not None and bin(plan["port_live"][m]).count("1") == 1   # (flatten.cpp's OSC_CONST_FAST: the carried-phase forms)
                if everything:
                    w = OSC_F32_BLEP | OSC_SINE_LOOSE | (OSC_ONE_PORT if one_port else 0)
                elif not plan["osc_exact"][m]:
                    fixed = plan["saw_fixed"][m] and one_port and (per_voice is None or m in per_voice)
                    w = (0 if plan["exact_blep"][m] else OSC_F32_BLEP | (OSC_ONE_PORT if one_port else 0) | (OSC_FIXED if fixed else 0)) | (OSC_SINE_LOOSE if plan["sine_loose"][m] else OSC_SINE_FAST)
            elif t == oracle.MOD_MOOG_FILTER:
                w = VCF_CONTRACTED if everything or not plan["literal"][m] else 0
            elif t == oracle.MOD_NONLINEAR:
                w = NONLIN_LOOSE if everything or plan["nonlin_loose"][m] else 0
            if w:
                self.set_forms(m, w)
                out[m] = w
        return out
