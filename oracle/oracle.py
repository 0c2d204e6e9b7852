"""ctypes binding of the CPU oracle (oracle/srack_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (s-rack_amd/) never does.  See the header of srack_oracle.c for what the oracle
restates (reference file:line) and for its parity-pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsrack_oracle.so")

# module types / fields: numeric vocabulary of include/srack_hip.h
MOD_OUTPUT, MOD_OSCILLATOR, MOD_MOOG_FILTER, MOD_ADSR, MOD_VCA, MOD_MONO_MIXER, MOD_MATH, MOD_GRID_SEQUENCER, MOD_PATTERN_SEQUENCER, MOD_NONLINEAR, MOD_SAMPLE, MOD_NOISE, MOD_FREEVERB = range(13)


def build(force=False):
    """Compile the oracle with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "srack_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "srack_hip.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def bind(L):
    """Argument types of the oracle's C entry points on a loaded library (this one, or a test's own build of srack_oracle.c)."""
    vp, i32, u32, u64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.or_patch_new.restype = vp
    L.or_patch_new.argtypes = [u32, u32, u32]
    L.or_patch_free.argtypes = [vp]
    L.or_patch_clone.restype = vp
    L.or_patch_clone.argtypes = [vp]
    L.or_add_module.argtypes = [vp, i32]
    L.or_num_modules.argtypes = [vp]
    L.or_connect.argtypes = [vp, i32, i32, i32, i32]
    L.or_disconnect.argtypes = [vp, i32, i32]
    L.or_set_field.argtypes = [vp, i32, i32, dbl]
    L.or_get_field.argtypes = [vp, i32, i32, dp]
    L.or_set_step.argtypes = [vp, i32, i32, i32, i32, i32]
    L.or_set_wave.argtypes = [vp, i32, fp, u32, C.c_float]
    L.or_set_output_buffer.argtypes = [vp, i32, i32, fp]
    L.or_set_noise_seed.restype = None
    L.or_set_noise_seed.argtypes = [vp, u64, u64]
    L.or_plan.argtypes = [vp]
    L.or_plan_list.argtypes = [vp, i32, ip, i32]
    L.or_get_plan.argtypes = [vp, ip, i32]
    L.or_get_removed_edges.argtypes = [vp, ip, i32]
    L.or_module_calc.argtypes = [vp, i32]
    L.or_execute.argtypes = [vp]
    L.or_get_output.argtypes = [vp, i32, i32, fp]
    L.or_render.argtypes = [vp, u32, fp, i32, i32, fp]
    L.or_render_batch.argtypes = [vp, u32, u32, i32, ip, ip, C.POINTER(dp), fp, dp, i32]
    L.or_voice_uniform.restype = C.c_float
    L.or_voice_uniform.argtypes = [u64, u64, u32]
    return L


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = bind(C.CDLL(_LIB_PATH))
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OraclePatch:
    """One module-object graph, as the reference's workspace holds it."""

    def __init__(self, sample_rate=48000, buffer_size=1024, channels=2, _handle=None, _lib=None):
        self.L = _lib or lib()
        self.sample_rate, self.buffer_size, self.channels = sample_rate, buffer_size, channels
        self.h = _handle if _handle is not None else self.L.or_patch_new(sample_rate, buffer_size, channels)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_patch_free(self.h)
            self.h = None

    def clone(self):
        return type(self)(self.sample_rate, self.buffer_size, self.channels, _handle=self.L.or_patch_clone(self.h), _lib=self.L)

    def add_module(self, mtype):
        r = self.L.or_add_module(self.h, mtype)
        if r < 0:
            raise ValueError("or_add_module failed")
        return r

    def num_modules(self):
        return self.L.or_num_modules(self.h)

    def connect(self, src, src_port, sink, sink_port):
        r = self.L.or_connect(self.h, src, src_port, sink, sink_port)
        if r < 0:
            raise ValueError(f"or_connect failed ({r})")

    def disconnect(self, sink, sink_port):
        if self.L.or_disconnect(self.h, sink, sink_port) < 0:
            raise ValueError("or_disconnect failed")

    def set_field(self, module, field, value):
        if self.L.or_set_field(self.h, module, field, float(value)) < 0:
            raise ValueError("or_set_field failed")

    def set_step(self, module, channel, step, state, value=0):
        if self.L.or_set_step(self.h, module, channel, step, state, value) < 0:
            raise ValueError("or_set_step failed")

    def set_wave(self, module, samples, sample_rate):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        if self.L.or_set_wave(self.h, module, _fp(a), a.size, float(sample_rate)) < 0:
            raise ValueError("or_set_wave failed")

    def set_output_buffer(self, module, port, samples):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        assert a.size == self.buffer_size
        if self.L.or_set_output_buffer(self.h, module, port, _fp(a)) < 0:
            raise ValueError("or_set_output_buffer failed")

    def set_noise_seed(self, seed, voice=0):
        """Noise streams: seed and the GLOBAL index of the voice this patch is (render_batch: of its voice 0)."""
        self.L.or_set_noise_seed(self.h, seed, voice)

    def get_field(self, module, field):
        v = C.c_double()
        if self.L.or_get_field(self.h, module, field, C.byref(v)) < 0:
            raise ValueError("or_get_field failed")
        return v.value

    def plan(self, output=None, all_modules=None):
        """plan_execution; with all_modules=None plans like the workspace (list order 0..n-1)."""
        if all_modules is None and output is None:
            n = self.L.or_plan(self.h)
        else:
            if all_modules is None:
                all_modules = list(range(self.num_modules()))
            arr = (C.c_int * len(all_modules))(*all_modules)
            n = self.L.or_plan_list(self.h, output, arr, len(all_modules))
        out = (C.c_int * max(n, 1))()
        self.L.or_get_plan(self.h, out, n)
        return list(out[:n])

    def removed_edges(self):
        buf = (C.c_int * 256)()
        n = self.L.or_get_removed_edges(self.h, buf, 128)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]

    def module_calc(self, module):
        self.L.or_module_calc(self.h, module)

    def execute(self):
        self.L.or_execute(self.h)

    def get_output(self, module, port):
        a = np.empty(self.buffer_size, dtype=np.float32)
        if self.L.or_get_output(self.h, module, port, _fp(a)) < 0:
            raise ValueError("or_get_output failed")
        return a

    def render(self, n_samples, tap=None):
        """-> [channels][n_samples] f32 (and the tapped wire if tap=(module, port))."""
        out = np.empty((self.channels, n_samples), dtype=np.float32)
        if tap is None:
            self.L.or_render(self.h, n_samples, _fp(out), -1, 0, None)
            return out
        t = np.empty(n_samples, dtype=np.float32)
        self.L.or_render(self.h, n_samples, _fp(out), tap[0], tap[1], _fp(t))
        return out, t

    def render_batch(self, n_voices, n_samples, overrides=(), frames=True, mix=False, threads=1):
        """overrides: [(module, field, values[n_voices])].  -> frames [C][T][V] f32, mix [C][T] f64."""
        n_ov = len(overrides)
        mods = (C.c_int * max(n_ov, 1))(*[o[0] for o in overrides])
        flds = (C.c_int * max(n_ov, 1))(*[o[1] for o in overrides])
        vals = [np.ascontiguousarray(o[2], dtype=np.float64) for o in overrides]
        for v in vals:
            assert v.shape == (n_voices,)
        ptrs = (C.POINTER(C.c_double) * max(n_ov, 1))(*[v.ctypes.data_as(C.POINTER(C.c_double)) for v in vals])
        fr = np.empty((self.channels, n_samples, n_voices), dtype=np.float32) if frames else None
        mx = np.empty((self.channels, n_samples), dtype=np.float64) if mix else None
        self.L.or_render_batch(self.h, n_voices, n_samples, n_ov, mods, flds, ptrs,
                               _fp(fr) if frames else None,
                               mx.ctypes.data_as(C.POINTER(C.c_double)) if mix else None, threads)
        return fr, mx


def voice_uniform(seed, n_voices, k, first_voice=0):
    """Counter-based per-voice U[0,1) f32 (splitmix64, top 24 bits) — SURVEY §8(d) cfg3."""
    L = lib()
    return np.array([L.or_voice_uniform(seed, first_voice + v, k) for v in range(n_voices)], dtype=np.float32)
