"""srack_amd — host-side binding of the MI355X batch-render path (libsrack_hip.so, C ABI in include/srack_hip.h).

This module is only a ctypes binding plus a thin `Patch` class whose methods carry the reference's
names (`add_module`, `set_input`/`connect`, `plan_execution` -> `plan`, `execute` -> `render`).
All computation happens in the HIP library; there is NO CPU fallback: if the shared library is
missing the import fails, and a render on a host without a GPU returns SRACK_ERR_DEVICE.
"""
import ctypes as C
import os

import numpy as np

from . import workloads  # noqa: F401  (patch builders; pure Python)
from .workloads import *  # noqa: F401,F403  (module / field / port enums)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrack_hip.so")

OK, ERR_INVALID, ERR_PORT, ERR_NO_OUTPUT, ERR_SELF_LOOP, ERR_STATE, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7, -8
RENDER_DEFAULT, RENDER_EXACT_OSC, RENDER_NO_FUSION, RENDER_NO_UNIFORM_HOIST, RENDER_NO_CTL_STAGES, RENDER_NO_SPECIALIZE, RENDER_SPECIALIZE, RENDER_KEEP_DEFAULT = 0, 1, 2, 4, 8, 16, 32, 64

# every symbol include/srack_hip.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = [
    "srack_abi_version", "srack_last_error", "srack_patch_create", "srack_patch_destroy", "srack_patch_add_module",
    "srack_patch_num_modules", "srack_patch_module_type", "srack_module_num_inputs", "srack_module_num_outputs",
    "srack_patch_set_field", "srack_patch_get_field", "srack_patch_keep_state", "srack_patch_set_step", "srack_patch_get_step", "srack_patch_set_wave", "srack_patch_get_wave", "srack_patch_load_srk", "srack_patch_save_srk", "srack_patch_module_id",
    "srack_patch_set_module_position", "srack_patch_get_module_position", "srack_patch_set_output_buffer", "srack_patch_get_output_buffer", "srack_patch_set_noise_seed", "srack_patch_connect", "srack_patch_disconnect", "srack_patch_get_input",
    "srack_patch_plan", "srack_patch_plan_list", "srack_patch_removed_edges", "srack_patch_delayed_edges",
    "srack_voices_configure", "srack_voices_set_field_f32", "srack_voices_set_field_f64", "srack_render_planes", "srack_render", "srack_render_reserve",
    "srack_render_info", "srack_render_kernel_source", "srack_render_kernel_compile", "srack_render_kernel_ms", "srack_voices_get_field", "srack_kernel_cache_set_dir", "srack_kernel_cache_stats", "srack_device_count", "srack_device_set", "srack_device_get",
    "srack_device_alloc", "srack_device_free", "srack_device_to_host", "srack_device_sync",
    "srack_dist_unique_id", "srack_dist_init", "srack_dist_comm_count", "srack_dist_destroy", "srack_dist_reduce_mix",
]


class KernelCacheInfo(C.Structure):
    """srack_kernel_cache_info (include/srack_hip.h)"""
    _fields_ = [("compiled", C.c_uint64), ("disk_hits", C.c_uint64), ("memory_hits", C.c_uint64), ("modules_loaded", C.c_uint64),
                ("code_evictions", C.c_uint64), ("module_evictions", C.c_uint64), ("resident_code_objects", C.c_uint64),
                ("resident_modules", C.c_uint64), ("compile_ms", C.c_double), ("directory", C.c_char * 512)]


class SrackError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"srack error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path.")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, dbl, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_size_t
    ip, dp, fp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.srack_last_error.restype = C.c_char_p
    L.srack_patch_create.argtypes = [u32, u32, u32, C.POINTER(vp)]
    L.srack_patch_destroy.argtypes = [vp]
    L.srack_patch_add_module.argtypes = [vp, i32]
    L.srack_patch_num_modules.argtypes = [vp]
    L.srack_patch_module_type.argtypes = [vp, i32]
    L.srack_module_num_inputs.argtypes = [vp, i32]
    L.srack_module_num_outputs.argtypes = [vp, i32]
    L.srack_patch_set_field.argtypes = [vp, i32, i32, dbl]
    L.srack_patch_get_field.argtypes = [vp, i32, i32, dp]
    L.srack_patch_set_step.argtypes = [vp, i32, i32, i32, i32, i32]
    L.srack_patch_get_step.argtypes = [vp, i32, i32, i32, ip, ip]
    L.srack_patch_set_wave.argtypes = [vp, i32, fp, u32, C.c_float]
    L.srack_patch_get_wave.argtypes = [vp, i32, fp, u32, fp]
    L.srack_patch_load_srk.argtypes = [C.c_char_p, sz, u32, u32, u32, C.POINTER(vp)]
    L.srack_patch_save_srk.argtypes = [vp, C.c_char_p, sz, C.POINTER(sz)]
    L.srack_patch_module_id.argtypes = [vp, i32, C.c_char_p, sz]
    L.srack_patch_set_module_position.argtypes = [vp, i32, C.c_float, C.c_float]
    L.srack_patch_get_module_position.argtypes = [vp, i32, fp, fp]
    L.srack_patch_set_output_buffer.argtypes = [vp, i32, i32, fp, u32]
    if hasattr(L, "srack_patch_get_output_buffer"):  # (tools/ab.sh alternates older builds of the library under this binding)
        L.srack_patch_get_output_buffer.argtypes = [vp, i32, i32, fp, u32]
    L.srack_patch_set_noise_seed.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.srack_patch_keep_state.argtypes = [vp, i32]
    L.srack_patch_connect.argtypes = [vp, i32, i32, i32, i32]
    L.srack_patch_disconnect.argtypes = [vp, i32, i32]
    L.srack_patch_get_input.argtypes = [vp, i32, i32, ip, ip]
    L.srack_patch_plan.argtypes = [vp, ip, i32]
    L.srack_patch_plan_list.argtypes = [vp, i32, ip, i32, ip, i32]
    L.srack_patch_removed_edges.argtypes = [vp, ip, i32]
    L.srack_patch_delayed_edges.argtypes = [vp, ip, i32]
    L.srack_voices_configure.argtypes = [vp, u32]
    L.srack_voices_set_field_f32.argtypes = [vp, i32, i32, fp]
    L.srack_voices_set_field_f64.argtypes = [vp, i32, i32, dp]
    L.srack_render_planes.argtypes = [vp, ip, i32]
    L.srack_render.argtypes = [vp, u32, vp, vp, u32, vp]
    L.srack_render_reserve.argtypes = [vp, u32, i32, u32]
    L.srack_render_info.argtypes = [vp, C.c_char_p, sz]
    L.srack_render_kernel_ms.argtypes = [vp, dp, ip, i32]
    L.srack_render_kernel_source.argtypes = [vp, u32, C.c_char_p, sz]
    L.srack_render_kernel_compile.argtypes = [vp, u32]
    L.srack_voices_get_field.argtypes = [vp, i32, i32, dp]
    L.srack_kernel_cache_set_dir.argtypes = [C.c_char_p]
    L.srack_kernel_cache_stats.argtypes = [C.POINTER(KernelCacheInfo)]
    L.srack_device_count.argtypes = [ip]
    L.srack_device_set.argtypes = [i32]
    L.srack_device_get.argtypes = [ip, C.c_char_p, sz]
    L.srack_device_alloc.argtypes = [C.POINTER(vp), sz]
    L.srack_device_free.argtypes = [vp]
    L.srack_device_to_host.argtypes = [vp, vp, sz, vp]
    L.srack_device_sync.argtypes = [vp]
    L.srack_dist_unique_id.argtypes = [C.c_char_p]
    L.srack_dist_init.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
    L.srack_dist_comm_count.argtypes = [vp, ip]
    L.srack_dist_destroy.argtypes = [vp]
    L.srack_dist_reduce_mix.argtypes = [vp, vp, sz, i32, vp]
    return L


lib = _load()


def _check(rc):
    if rc < 0:
        raise SrackError(rc, lib.srack_last_error().decode(errors="replace"))
    return rc


def device_count():
    n = C.c_int(0)
    lib.srack_device_count(C.byref(n))
    return n.value


def kernel_cache_set_dir(path):
    """The disk level of the specialised-kernel cache: a directory, "off", or None for the default resolution."""
    _check(lib.srack_kernel_cache_set_dir(None if path is None else os.fsencode(path)))


def kernel_cache_stats():
    st = KernelCacheInfo()
    _check(lib.srack_kernel_cache_stats(C.byref(st)))
    d = {k: getattr(st, k) for k, _ in KernelCacheInfo._fields_}
    d["directory"] = st.directory.decode(errors="replace")
    return d


def device_get():
    """(current device of this thread — the one a render launches on —, its PCI bus id)"""
    d, bus = C.c_int(-1), C.create_string_buffer(64)
    _check(lib.srack_device_get(C.byref(d), bus, 64))
    return d.value, bus.value.decode(errors="replace")


DIST_ID_BYTES = 128


class MixComm:
    """The path's one collective behind the C ABI: an RCCL communicator over the ranks that share a render (one process per
    GPU), used for nothing but the sum of the per-rank partial mixes (srack_dist_*).

    Rank 0 calls `MixComm.unique_id()` and hands the 128 bytes to every rank over whatever side channel the host has;
    every rank then constructs `MixComm(id, n_ranks, rank)` with its device already selected (srack_device_set)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(DIST_ID_BYTES)
        _check(lib.srack_dist_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id, n_ranks, rank):
        assert len(unique_id) == DIST_ID_BYTES
        self.comm = C.c_void_p()
        _check(lib.srack_dist_init(bytes(unique_id), n_ranks, rank, C.byref(self.comm)))
        self.n_ranks, self.rank = n_ranks, rank

    def count(self):
        n = C.c_int()
        _check(lib.srack_dist_comm_count(self.comm, C.byref(n)))
        return n.value

    def reduce_mix(self, d_mix, count, root=0, stream=None):
        """ncclReduce(sum, f32) of `count` floats at device pointer `d_mix`, in place, asynchronous on `stream`."""
        _check(lib.srack_dist_reduce_mix(self.comm, d_mix, count, root, stream))

    def destroy(self):
        if self.comm:
            _check(lib.srack_dist_destroy(self.comm))
            self.comm = C.c_void_p()


class Patch:
    """The workspace's module list + plan + N voices, behind the C ABI.

    Reference API mirrored: `SynthModule::set_input` -> connect, `disconnect_input` -> disconnect,
    `get_input`, `get_num_inputs/outputs`, `plan_execution` -> plan, `execute` (x ceil(T/B) blocks)
    -> render.  Errors the reference reports as `Err(())` / panics raise SrackError(code).
    """

    def __init__(self, sample_rate=48000, buffer_size=1024, channels=2, _handle=None):
        self.sample_rate, self.buffer_size, self.channels = sample_rate, buffer_size, channels
        h = _handle if _handle is not None else C.c_void_p()
        if _handle is None:
            _check(lib.srack_patch_create(sample_rate, buffer_size, channels, C.byref(h)))
        self.h = h
        self.n_voices = 0

    # ---- .srk rack files (FileFormat, ui.rs:578-586) -------------------------------------------------
    @classmethod
    def load_srk(cls, data, sample_rate=48000, buffer_size=1024, channels=2):
        """SynthModuleWorkspaceImpl::deserialize (ui.rs:116-135) against this AudioConfig."""
        h = C.c_void_p()
        data = bytes(data)
        _check(lib.srack_patch_load_srk(data, len(data), sample_rate, buffer_size, channels, C.byref(h)))
        return cls(sample_rate, buffer_size, channels, _handle=h)

    def save_srk(self):
        n = C.c_size_t()
        _check(lib.srack_patch_save_srk(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib.srack_patch_save_srk(self.h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def module_id(self, module):
        buf = C.create_string_buffer(64)
        _check(lib.srack_patch_module_id(self.h, module, buf, 64))
        return buf.value.decode()

    def set_module_position(self, module, x, y):
        _check(lib.srack_patch_set_module_position(self.h, module, x, y))

    def get_module_position(self, module):
        x, y = C.c_float(), C.c_float()
        return (x.value, y.value) if _check(lib.srack_patch_get_module_position(self.h, module, C.byref(x), C.byref(y))) else None

    def set_output_buffer(self, module, port, samples):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        _check(lib.srack_patch_set_output_buffer(self.h, module, port, a.ctypes.data_as(C.POINTER(C.c_float)), a.size))

    def get_output_buffer(self, module, port):
        """The block the patch holds for this port before the first tick (a loaded file's, or set_output_buffer's); empty: fresh zeros."""
        n = _check(lib.srack_patch_get_output_buffer(self.h, module, port, None, 0))
        a = np.zeros(n, dtype=np.float32)
        if n:
            _check(lib.srack_patch_get_output_buffer(self.h, module, port, a.ctypes.data_as(C.POINTER(C.c_float)), n))
        return a

    def keep_state(self, keep=True):
        """Edits between renders no longer restart the voices: the modules' device state is carried into the re-flattened program."""
        _check(lib.srack_patch_keep_state(self.h, 1 if keep else 0))

    def set_noise_seed(self, seed, first_voice=0):
        """Noise modules: stream = (seed, module, first_voice + voice); a rank owning global voices [r*V, (r+1)*V) passes r*V."""
        _check(lib.srack_patch_set_noise_seed(self.h, seed, first_voice))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # `lib` is already gone at interpreter shutdown
            lib.srack_patch_destroy(self.h)
            self.h = None

    # ---- graph ------------------------------------------------------------------------------
    def add_module(self, module_type):
        return _check(lib.srack_patch_add_module(self.h, module_type))

    def num_modules(self):
        return _check(lib.srack_patch_num_modules(self.h))

    def module_type(self, module):
        return _check(lib.srack_patch_module_type(self.h, module))

    def get_num_inputs(self, module):
        return _check(lib.srack_module_num_inputs(self.h, module))

    def get_num_outputs(self, module):
        return _check(lib.srack_module_num_outputs(self.h, module))

    def set_field(self, module, field, value):
        _check(lib.srack_patch_set_field(self.h, module, field, float(value)))

    def get_field(self, module, field):
        v = C.c_double()
        _check(lib.srack_patch_get_field(self.h, module, field, C.byref(v)))
        return v.value

    def set_step(self, module, channel, step, state, value=0):
        _check(lib.srack_patch_set_step(self.h, module, channel, step, state, value))

    def get_step(self, module, channel, step):
        st, v = C.c_int(), C.c_int()
        _check(lib.srack_patch_get_step(self.h, module, channel, step, C.byref(st), C.byref(v)))
        return st.value, v.value

    def set_wave(self, module, samples, sample_rate):
        """WaveBox::load's result for a SampleModule: first-channel f32 samples + the file's sample rate."""
        a = np.ascontiguousarray(samples, dtype=np.float32)
        _check(lib.srack_patch_set_wave(self.h, module, a.ctypes.data_as(C.POINTER(C.c_float)), a.size, float(sample_rate)))

    def get_wave(self, module):
        sr = C.c_float()
        n = _check(lib.srack_patch_get_wave(self.h, module, None, 0, C.byref(sr)))
        a = np.zeros(n, dtype=np.float32)
        _check(lib.srack_patch_get_wave(self.h, module, a.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(sr)))
        return a, sr.value

    def connect(self, src, src_port, sink, sink_port):
        _check(lib.srack_patch_connect(self.h, src, src_port, sink, sink_port))

    def disconnect(self, sink, sink_port):
        _check(lib.srack_patch_disconnect(self.h, sink, sink_port))

    def get_input(self, sink, sink_port):
        m, p = C.c_int(), C.c_int()
        _check(lib.srack_patch_get_input(self.h, sink, sink_port, C.byref(m), C.byref(p)))
        return None if m.value < 0 else (m.value, p.value)

    def plan(self, output=None, all_modules=None):
        buf = (C.c_int * 1024)()
        if output is None and all_modules is None:
            n = _check(lib.srack_patch_plan(self.h, buf, 1024))
        else:
            if all_modules is None:
                all_modules = list(range(self.num_modules()))
            arr = (C.c_int * len(all_modules))(*all_modules)
            n = _check(lib.srack_patch_plan_list(self.h, output, arr, len(all_modules), buf, 1024))
        return list(buf[:n])

    def removed_edges(self):
        buf = (C.c_int * 512)()
        n = _check(lib.srack_patch_removed_edges(self.h, buf, 256))
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]

    def delayed_edges(self):
        buf = (C.c_int * 1024)()
        n = _check(lib.srack_patch_delayed_edges(self.h, buf, 256))
        return [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]

    # ---- voices -----------------------------------------------------------------------------
    def configure_voices(self, n_voices):
        _check(lib.srack_voices_configure(self.h, n_voices))
        self.n_voices = n_voices

    def set_voice_field(self, module, field, values):
        values = np.asarray(values)
        assert values.shape == (self.n_voices,)
        if values.dtype == np.float64:
            a = np.ascontiguousarray(values)
            _check(lib.srack_voices_set_field_f64(self.h, module, field, a.ctypes.data_as(C.POINTER(C.c_double))))
        else:
            a = np.ascontiguousarray(values, dtype=np.float32)
            _check(lib.srack_voices_set_field_f32(self.h, module, field, a.ctypes.data_as(C.POINTER(C.c_float))))

    def get_voice_field(self, module, field):
        out = np.empty(self.n_voices, dtype=np.float64)
        _check(lib.srack_voices_get_field(self.h, module, field, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    # ---- render -----------------------------------------------------------------------------
    def planes(self):
        buf = (C.c_int * 8)()
        n = _check(lib.srack_render_planes(self.h, buf, 8))
        return n, list(buf[:self.channels])

    def reserve(self, n_samples, want_mix=True, flags=0):
        """First-use set-up (flatten, upload, scratch buffers) ahead of the first render."""
        _check(lib.srack_render_reserve(self.h, n_samples, 1 if want_mix else 0, flags))

    def render_raw(self, n_samples, d_frames=None, d_mix=None, flags=0, stream=None):
        """Device pointers (ints) in; asynchronous on `stream`."""
        _check(lib.srack_render(self.h, n_samples, d_frames, d_mix, flags, stream))

    def info(self):
        n = _check(lib.srack_render_info(self.h, None, 0))   # (the length: a description names every control unit and may pass any fixed size)
        buf = C.create_string_buffer(n + 1)
        _check(lib.srack_render_info(self.h, buf, n + 1))
        return buf.value.decode()

    def kernel_source(self, flags=0):
        """The HIP source of the voice kernel specialised for this patch (SrackError(ERR_UNSUPPORTED) if the generator cannot express it)."""
        n = _check(lib.srack_render_kernel_source(self.h, flags, None, 0))
        buf = C.create_string_buffer(n + 1)
        _check(lib.srack_render_kernel_source(self.h, flags, buf, n + 1))
        return buf.value.decode()

    def kernel_compile(self, flags=0):
        """Compile that source for gfx950 with hiprtc (no GPU needed)."""
        _check(lib.srack_render_kernel_compile(self.h, flags))

    def kernel_ms(self, reset=True):
        ms, n = C.c_double(), C.c_int()
        _check(lib.srack_render_kernel_ms(self.h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def render(self, n_samples, frames=True, mix=True, flags=0):
        """Convenience for tests: allocates device buffers through the C ABI's device helpers,
        renders, copies back.  -> (frames [planes][T][V] f32 or None, mix [C][T] f32 or None)."""
        if self.n_voices == 0:
            self.configure_voices(1)
        n_planes, _ = self.planes()
        V, T, Cn = self.n_voices, n_samples, self.channels
        d_fr, d_mx = C.c_void_p(), C.c_void_p()
        try:
            if frames and n_planes > 0:
                _check(lib.srack_device_alloc(C.byref(d_fr), max(1, n_planes * T * V * 4)))
            if mix:
                _check(lib.srack_device_alloc(C.byref(d_mx), max(1, Cn * T * 4)))
            self.render_raw(T, d_fr if d_fr.value else None, d_mx if d_mx.value else None, flags, None)
            fr = mx = None
            if frames:
                fr = np.zeros((n_planes, T, V), dtype=np.float32)
                if d_fr.value:
                    _check(lib.srack_device_to_host(fr.ctypes.data_as(C.c_void_p), d_fr, fr.nbytes, None))
            if mix:
                mx = np.empty((Cn, T), dtype=np.float32)
                _check(lib.srack_device_to_host(mx.ctypes.data_as(C.c_void_p), d_mx, mx.nbytes, None))
            _check(lib.srack_device_sync(None))
            return fr, mx
        finally:
            if d_fr.value:
                lib.srack_device_free(d_fr)
            if d_mx.value:
                lib.srack_device_free(d_mx)

    def render_channels(self, n_samples, flags=0):
        """-> [channels][T][V] f32 (planes expanded to channels, silence for unconnected ones)."""
        fr, _ = self.render(n_samples, frames=True, mix=False, flags=flags)
        _, cp = self.planes()
        out = np.zeros((self.channels, n_samples, self.n_voices), dtype=np.float32)
        for c, p in enumerate(cp):
            if p >= 0:
                out[c] = fr[p]
        return out
