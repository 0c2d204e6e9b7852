"""Second, independent CPU restatement of the s-rack tick (NumPy float32 scalars + Python float64).

TEST INFRASTRUCTURE ONLY — same rule as oracle/srack_oracle.c: never imported by the product.

Why it exists: the reference's own tests pin only the oscillator's sine port and the planner
order.  Everything else is "parity unpinned" by the reference, so fidelity is argued
by two restatements written separately (C, and this file) agreeing bit for bit on the same
patches (tests/test_oracle.py, tests/golden/make_golden.py).  NumPy float32 scalar arithmetic is
IEEE-exact with no contraction; math.pow / math.sin / math.fmod call the same glibc that Rust's
f64::powf / sin / % reach on x86_64-linux-gnu.

Written object-style like the reference (one class per module, calc() fills block buffers);
citations are to /root/reference/src.
"""
import ctypes
import math

import numpy as np

# f32::powf is libm's powf (math.rs:204, sample.rs:236); Python has no f32 pow of its own, and NumPy's float32 power is
# NumPy's own SIMD routine, so call the same glibc symbol Rust reaches
_libm = ctypes.CDLL("libm.so.6")
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
# math.pow / math.sin / math.fmod raise OverflowError / ValueError where libm returns inf / NaN (a patch that blows up): call libm itself
for _name in ("pow", "fmod"):
    getattr(_libm, _name).restype = ctypes.c_double
    getattr(_libm, _name).argtypes = [ctypes.c_double, ctypes.c_double]
_libm.sin.restype = ctypes.c_double
_libm.sin.argtypes = [ctypes.c_double]


def powf(a, b):
    return np.float32(_libm.powf(float(a), float(b)))

f32 = np.float32
ZERO, ONE, TWO = f32(0.0), f32(1.0), f32(2.0)


def fmin(a, b):  # Rust's f32::min: if one operand is NaN the other is returned (Python's min(a, b) would keep a NaN first operand)
    return b if a != a else (a if b != b else (b if b < a else a))


def fmax(a, b):  # Rust's f32::max
    return b if a != a else (a if b != b else (b if b > a else a))


class TransitionDetector:  # synth.rs:276-298
    def __init__(self):
        self.last = True

    def is_transition(self, val):
        above = bool(val > ZERO)
        t = above and not self.last
        self.last = above
        return t


class Module:
    n_in = 0
    n_out = 0

    def __init__(self, cfg):
        self.cfg = cfg
        self.inputs = [None] * self.n_in  # (module, port) or None
        self.outs = [np.zeros(cfg["buffer_size"], dtype=f32) for _ in range(self.n_out)]

    def resolve(self, k):  # synth.rs:249-254
        if self.inputs[k] is None:
            return None
        m, port = self.inputs[k]
        return m.outs[port]


class Oscillator(Module):  # oscillator.rs
    n_in, n_out = 2, 3

    def __init__(self, cfg):
        super().__init__(cfg)
        self.val = f32(0.0)
        self.sample_rate = cfg["sample_rate"]
        self.pos = 0.0
        self.antialiasing = True
        self.sync = TransitionDetector()

    @staticmethod
    def poly_blep(t, dt):  # oscillator.rs:50-67
        if dt == 0.0:
            return 0.0
        if t < dt:
            t /= dt
            return t + t - t * t - 1.0
        elif t > 1.0 - dt:
            t = (t - 1.0) / dt
            return t * t + t + t + 1.0
        return 0.0

    def calc(self):  # oscillator.rs:108-158
        cv, sync_in = self.resolve(0), self.resolve(1)
        sine, square, saw = self.outs
        for i in range(len(sine)):
            sync_val = sync_in[i] if sync_in is not None else ZERO
            if self.sync.is_transition(sync_val):
                self.pos = 0.0
            if cv is not None:
                hz = 440.0 * _libm.pow(2.0, float(cv[i]) + float(self.val))
            else:
                hz = 440.0 * _libm.pow(2.0, float(self.val))
            delta = hz / float(self.sample_rate)
            sine[i] = f32(_libm.sin(self.pos * math.pi * 2.0))
            lvl = f32(-1.0) if self.pos < 0.5 else f32(1.0)
            if self.antialiasing:
                square[i] = lvl - f32(self.poly_blep(self.pos, delta) - self.poly_blep(_libm.fmod(self.pos + 0.5, 1.0), delta))
                saw[i] = (f32(self.pos) * TWO - ONE) - f32(self.poly_blep(self.pos, delta))
            else:
                square[i] = lvl - ZERO
                saw[i] = (f32(self.pos) * TWO - ONE) - ZERO
            self.pos += delta
            self.pos = _libm.fmod(self.pos, 1.0)


class MoogFilter(Module):  # filter.rs
    n_in, n_out = 2, 3

    def __init__(self, cfg):
        super().__init__(cfg)
        self.freq, self.res, self.exp_amt = f32(0.2), f32(0.5), f32(0.5)
        self.f = self.p = self.q = ZERO
        self.b = [ZERO] * 5
        self.sfreq = self.sres = ZERO

    def state_calc(self, x, frequency, res):  # filter.rs:58-92
        if frequency != self.sfreq or res != self.sres:
            self.sfreq, self.sres = frequency, res
            self.q = ONE - self.sfreq
            self.p = self.sfreq + f32(0.8) * self.sfreq * self.q
            self.f = self.p * TWO - ONE
            self.q = self.sres * (ONE + f32(0.5) * self.q * (ONE - self.q + f32(5.6) * self.q * self.q))
        b = self.b
        x = x - (self.q * b[4])
        t1 = b[1]
        b[1] = (x + b[0]) * self.p - b[1] * self.f
        t2 = b[2]
        b[2] = (b[1] + t1) * self.p - b[2] * self.f
        t1 = b[3]
        b[3] = (b[2] + t2) * self.p - b[3] * self.f
        b[4] = (b[3] + t1) * self.p - b[4] * self.f
        b[4] = b[4] - (b[4] * b[4] * b[4]) * f32(0.166667)
        b[0] = x
        for k in range(5):
            b[k] = fmax(fmin(b[k], ONE), f32(-1.0))  # x.min(1.0).max(-1.0), filter.rs:89
        return b[4], x - b[4], f32(3.0) * (b[3] - b[4])

    def calc(self):  # filter.rs:182-221
        audio_in, cv_in = self.resolve(0), self.resolve(1)
        lowpass, bandpass, highpass = self.outs
        res = fmin(fmax(self.res, ZERO), ONE)
        for idx in range(len(lowpass)):
            audio = audio_in[idx] if audio_in is not None else ZERO
            cv = cv_in[idx] if cv_in is not None else ZERO
            frequency = fmin(fmax(self.freq + cv * self.exp_amt, ZERO), f32(0.9))
            lowpass[idx], highpass[idx], bandpass[idx] = self.state_calc(audio, frequency, res)


ATTACK, DECAY, SUSTAIN, RELEASE, NONE = range(5)  # adsr.rs:27-33


class ADSR(Module):  # adsr.rs
    n_in, n_out = 1, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.a_sec, self.d_sec, self.s_val, self.r_sec = f32(0.0), f32(0.5), f32(0.25), f32(0.5)
        self.phase = ZERO
        self.mode = NONE
        self.r_val = self.from_a_val = ZERO
        self.sample_rate = f32(cfg["sample_rate"])
        self.td = TransitionDetector()

    def calc(self):  # adsr.rs:134-217
        gate = self.resolve(0)
        out = self.outs[0]
        with np.errstate(divide="ignore"):
            for idx in range(len(out)):
                is_transition = self.td.is_transition(gate[idx] if gate is not None else ZERO)
                high = gate is not None and bool(gate[idx] > ZERO)
                if self.mode == NONE:
                    if high:
                        self.phase, self.mode = ZERO, ATTACK
                elif self.mode == ATTACK:
                    self.phase = self.phase + ONE / (self.sample_rate * self.a_sec)
                    if self.phase >= ONE:
                        self.phase, self.mode = ZERO, DECAY
                    elif is_transition:
                        self.phase = ZERO
                        self.r_val = self.from_a_val
                elif self.mode == DECAY:
                    self.phase = self.phase + ONE / (self.sample_rate * self.d_sec)
                    if self.phase >= ONE:
                        self.phase, self.mode = ZERO, SUSTAIN
                    if is_transition:
                        self.phase, self.mode = ZERO, ATTACK
                elif self.mode == SUSTAIN:
                    if gate is None or bool(gate[idx] <= ZERO):  # adsr.rs:175 spells it `<= 0.0`: a NaN gate holds the sustain
                        self.phase, self.mode = ZERO, RELEASE
                    if is_transition:
                        self.phase, self.mode = ZERO, ATTACK
                elif self.mode == RELEASE:
                    if high:
                        self.phase, self.mode = ZERO, ATTACK
                    self.phase = self.phase + ONE / (self.sample_rate * self.r_sec)
                    if self.phase >= ONE:
                        self.phase, self.r_val, self.mode = ZERO, ZERO, NONE
                if self.mode == NONE:
                    o = ZERO
                elif self.mode == ATTACK:
                    o = self.r_val + (ONE - self.r_val) * self.phase
                elif self.mode == DECAY:
                    o = self.s_val + (ONE - self.s_val) * (ONE - self.phase)
                elif self.mode == SUSTAIN:
                    o = self.s_val
                else:
                    o = self.s_val * (ONE - self.phase)
                out[idx] = o
                if self.mode != ATTACK:
                    self.r_val = out[idx]
                else:
                    self.from_a_val = out[idx]


class VCA(Module):  # vca.rs:117-148
    n_in, n_out = 2, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.negative = False

    def calc(self):
        audio, cv = self.resolve(0), self.resolve(1)
        out = self.outs[0]
        if audio is not None and cv is not None:
            for i in range(len(out)):
                out[i] = audio[i] * cv[i] if (self.negative or cv[i] > ZERO) else ZERO
        else:
            out[:] = ZERO


class MonoMixer(Module):  # mixer.rs:101-122
    n_in, n_out = 4, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.gain = [ONE] * 4

    def calc(self):
        out = self.outs[0]
        out[:] = ZERO
        for k in range(4):
            buf = self.resolve(k)
            if buf is None:
                continue
            for i in range(len(out)):
                out[i] = out[i] + buf[i] * self.gain[k]


ADD, SUBTRACT, MULTIPLY = range(3)


class Math(Module):  # math.rs:139-160
    n_in, n_out = 2, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.constant = ZERO
        self.operation = ADD

    def calc(self):
        i1, i2 = self.resolve(0), self.resolve(1)
        out = self.outs[0]
        for i in range(len(out)):
            a = i1[i] if i1 is not None else ZERO
            b = i2[i] if i2 is not None else self.constant
            out[i] = a + b if self.operation == ADD else (a - b if self.operation == SUBTRACT else a * b)


class Output(Module):  # output.rs:46-60
    def __init__(self, cfg):
        self.n_in = cfg["channels"]
        super().__init__(cfg)
        self.outs = [np.zeros(cfg["buffer_size"], dtype=f32) for _ in range(cfg["channels"])]  # bufs

    def calc(self):
        for c in range(self.n_in):
            buf = self.resolve(c)
            if buf is not None:
                self.outs[c][:] = buf
            else:
                self.outs[c][:] = ZERO


class _Sequencer(Module):  # the stepping shared by both sequencers: sequencer.rs:219-231, 504-516
    n_in = 2

    def __init__(self, cfg):
        super().__init__(cfg)
        self.current_step = 0
        self.td = TransitionDetector()
        self.sync_td = TransitionDetector()

    def advance(self, step_in, sync_in, length):
        if self.td.is_transition(step_in):
            self.current_step += 1
        if self.sync_td.is_transition(sync_in):
            self.current_step = 0
        cs = self.current_step
        if cs >= length:
            self.current_step = 0
            cs = 0
        return cs


class GridSequencer(_Sequencer):  # sequencer.rs:12-246
    n_out = 3

    def __init__(self, cfg):
        super().__init__(cfg)
        self.sequence = [None] * 64  # None | (val, hold)
        self.steps_per_octave = 12
        self.octaves = 2
        self.last = ZERO

    def calc(self):  # sequencer.rs:190-246
        step_buf, sync_buf = self.resolve(0), self.resolve(1)
        cv_out, gate_out, sync_out = self.outs
        for idx in range(len(cv_out)):
            step_in = step_buf[idx] if step_buf is not None else ZERO
            sync_in = sync_buf[idx] if sync_buf is not None else ZERO
            cs = self.advance(step_in, sync_in, len(self.sequence))
            cell = self.sequence[cs]
            if cell is not None:
                val, hold = cell
                cv_out[idx] = f32(val) * (ONE / f32(self.steps_per_octave))
                gate_out[idx] = ONE if hold else step_in
            else:
                cv_out[idx] = self.last
                gate_out[idx] = ZERO
            sync_out[idx] = ONE if cs == 0 else ZERO
            self.last = cv_out[idx]


class PatternSequencer(_Sequencer):  # sequencer.rs:336-533
    n_out = 9

    def __init__(self, cfg):
        super().__init__(cfg)
        self.sequence = [[None] * 64 for _ in range(8)]  # None | False | True

    def calc(self):  # sequencer.rs:482-533
        step_buf, sync_buf = self.resolve(0), self.resolve(1)
        for idx in range(len(self.outs[0])):
            step_in = step_buf[idx] if step_buf is not None else ZERO
            sync_in = sync_buf[idx] if sync_buf is not None else ZERO
            cs = self.advance(step_in, sync_in, len(self.sequence[0]))
            for c in range(8):
                v = self.sequence[c][cs]
                self.outs[c][idx] = ZERO if v is None else (ONE if v else step_in)
            self.outs[8][idx] = ONE if cs == 0 else ZERO


class NonLinear(Module):  # math.rs:176-311
    n_in, n_out = 2, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.constant = ONE

    @staticmethod
    def operation(a, b):  # math.rs:203-205
        return powf(a, b) if a > ZERO else -powf(-a, b)

    def calc(self):  # math.rs:291-311
        i1, i2 = self.resolve(0), self.resolve(1)
        out = self.outs[0]
        for i in range(len(out)):
            out[i] = self.operation(i1[i] if i1 is not None else ZERO, i2[i] if i2 is not None else self.constant)


def as_usize(x):  # Rust `f32 as usize`: saturating, NaN -> 0
    x = float(x)
    if not x > 0.0:
        return 0
    return min(int(x), 2**64 - 1)


class Sample(Module):  # sample.rs:72-240
    n_in, n_out = 2, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.td = TransitionDetector()
        self.pos = ZERO
        self.playing = False
        self.sample_rate = f32(cfg["sample_rate"])
        self.samples = np.zeros(0, dtype=f32)  # WaveBox::default()
        self.wave_sample_rate = ZERO
        self.wave_new = False

    def load(self, samples, sample_rate):  # what WaveBox::load leaves behind, sample.rs:31-69
        self.samples = np.array(samples, dtype=f32)
        self.wave_sample_rate = f32(sample_rate)
        self.wave_new = True

    def calc(self):  # sample.rs:192-240
        gate_in, cv_in = self.resolve(0), self.resolve(1)
        out = self.outs[0]
        if self.wave_new:
            self.pos, self.playing, self.wave_new = ZERO, False, False
        with np.errstate(all="ignore"):
            for idx in range(len(out)):
                if self.td.is_transition(gate_in[idx] if gate_in is not None else ZERO):
                    self.pos, self.playing = ZERO, True
                if as_usize(self.pos) >= len(self.samples):
                    self.pos, self.playing = ZERO, False
                out[idx] = self.samples[as_usize(self.pos)] if len(self.samples) else ZERO
                if self.playing:
                    self.pos = self.pos + self.wave_sample_rate / self.sample_rate * powf(TWO, cv_in[idx] if cv_in is not None else ZERO)


_M64 = (1 << 64) - 1


def splitmix64(x):  # the output function applied to x + golden gamma (Steele, Lea, Flood 2014); Python ints, mod 2^64
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


class Noise(Module):  # oscillator.rs:308-393
    """`(rand::random::<f32>() - 0.5) * 2.0` per sample (oscillator.rs:385).  rand 0.8's Standard f32 is 24 random bits
    * 2^-24; the reference's bits come from an OS-seeded thread-local ChaCha12 (unreproducible), here from the
    counter-based stream documented in include/srack_hip.h: sample n = output n of splitmix64 seeded with
    key = sm(sm(seed ^ sm(module)) ^ global_voice)."""
    n_in, n_out = 0, 1

    def __init__(self, cfg):
        super().__init__(cfg)
        self.module_index = 0  # set by the graph adapter
        self.n = 0

    def calc(self):
        base = splitmix64(self.cfg.get("noise_seed", 0) ^ splitmix64(self.module_index))
        key = splitmix64(base ^ self.cfg.get("noise_voice", 0))
        out = self.outs[0]
        for i in range(len(out)):
            z = splitmix64((key + (self.n + i) * 0x9E3779B97F4A7C15) & _M64)
            r = f32(z >> 40) * f32(2.0 ** -24)
            out[i] = (r - f32(0.5)) * TWO
        self.n += len(out)


class _FvDelay:  # freeverb crate, delay_line.rs
    def __init__(self, length):
        self.buf = [0.0] * length
        self.i = 0

    def read(self):
        return self.buf[self.i]

    def write_and_advance(self, v):
        self.buf[self.i] = v
        self.i = 0 if self.i == len(self.buf) - 1 else self.i + 1


class _FvComb:  # comb.rs
    def __init__(self, length):
        self.d = _FvDelay(length)
        self.feedback, self.filter_state, self.dampening, self.dampening_inverse = 0.5, 0.0, 0.5, 0.5

    def tick(self, x):
        out = self.d.read()
        self.filter_state = out * self.dampening_inverse + self.filter_state * self.dampening
        self.d.write_and_advance(x + self.filter_state * self.feedback)
        return out


class _FvAllPass:  # all_pass.rs
    def __init__(self, length):
        self.d = _FvDelay(length)

    def tick(self, x):
        delayed = self.d.read()
        self.d.write_and_advance(x + delayed * 0.5)
        return -x + delayed


class _Freeverb:
    """freeverb crate 0.1.0 (Cargo.lock:1479-1482), un-vendored: restated from its published algorithm (Jezar's Freeverb in
    Ian Hobson's Rust port) — PARITY UNPINNED, see oracle/srack_oracle.c.  Python floats are IEEE doubles, as the crate's f64."""
    COMBS = (1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617)
    ALLPASSES = (556, 441, 341, 225)

    def __init__(self, sr):
        adj = lambda n: int(float(n) * float(sr) / 44100.0)
        self.combs = [(_FvComb(adj(n)), _FvComb(adj(n + 23))) for n in self.COMBS]
        self.allpasses = [(_FvAllPass(adj(n)), _FvAllPass(adj(n + 23))) for n in self.ALLPASSES]
        self.wet_gains = (0.0, 0.0)
        self.wet = self.width = self.dry = self.input_gain = self.dampening = self.room_size = 0.0
        self.frozen = False
        self.set_wet(1.0)
        self.set_width(0.5)
        self.set_dampening(0.5)
        self.set_room_size(0.5)
        self.frozen, self.input_gain = False, 1.0  # set_frozen(false)
        self.update_combs()

    def update_combs(self):
        fb, damp = (1.0, 0.0) if self.frozen else (self.room_size, self.dampening)
        for pair in self.combs:
            for c in pair:
                c.feedback, c.dampening, c.dampening_inverse = fb, damp, 1.0 - damp

    def update_wet_gains(self):
        self.wet_gains = (self.wet * (self.width / 2.0 + 0.5), self.wet * ((1.0 - self.width) / 2.0))

    def set_dampening(self, v):
        self.dampening = v * 0.4
        self.update_combs()

    def set_freeze(self, frozen):
        self.frozen = frozen
        self.update_combs()

    def set_wet(self, v):
        self.wet = v * 3.0
        self.update_wet_gains()

    def set_width(self, v):
        self.width = v
        self.update_wet_gains()

    def set_room_size(self, v):
        self.room_size = v * 0.28 + 0.7
        self.update_combs()

    def set_dry(self, v):
        self.dry = v

    def tick(self, in0, in1):
        x = (in0 + in1) * 0.015 * self.input_gain
        o0 = o1 = 0.0
        for a, b in self.combs:
            o0 += a.tick(x)
            o1 += b.tick(x)
        for a, b in self.allpasses:
            o0 = a.tick(o0)
            o1 = b.tick(o1)
        g0, g1 = self.wet_gains
        return o0 * g0 + o1 * g1 + in0 * self.dry, o1 * g0 + o0 * g1 + in1 * self.dry


class Freeverb(Module):  # freeverb.rs:8-274
    n_in, n_out = 2, 2
    PARAMS = ("dampening", "freeze", "wet", "width", "room_size", "dry")

    def __init__(self, cfg):
        super().__init__(cfg)
        self.freeverb = None
        self.sample_rate = int(cfg["sample_rate"])
        self.dampening = self.dampening_ctl = 0.5
        self.freeze = self.freeze_ctl = False
        self.wet = self.wet_ctl = 1.0
        self.width = self.width_ctl = 0.5
        self.room_size = self.room_size_ctl = 0.5
        self.dry = self.dry_ctl = 0.0

    def set_freeverb(self, all_):  # freeverb.rs:88-114
        for name in self.PARAMS:
            ctl = getattr(self, name + "_ctl")
            if ctl != getattr(self, name) or all_:
                setattr(self, name, ctl)
                getattr(self.freeverb, "set_" + name)(ctl)

    def calc(self):  # freeverb.rs:208-270
        if self.freeverb is None:
            self.freeverb = _Freeverb(self.sample_rate)
            self.set_freeverb(True)
        else:
            self.set_freeverb(False)
        l, r = self.resolve(0), self.resolve(1)
        for i in range(len(self.outs[0])):
            o0, o1 = self.freeverb.tick(float(l[i]) if l is not None else 0.0, float(r[i]) if r is not None else 0.0)
            self.outs[0][i], self.outs[1][i] = f32(o0), f32(o1)


CLASSES = [Output, Oscillator, MoogFilter, ADSR, VCA, MonoMixer, Math, GridSequencer, PatternSequencer, NonLinear, Sample, Noise, Freeverb]  # index = SRACK_MOD_*


def get_inputs(m):  # synth.rs:214-218
    return [m.inputs[k] for k in range(m.n_in)]


def is_loop(module, edges):  # synth.rs:107-126
    to_search, visited = [module], set()
    while True:
        current = next((m for m in to_search if id(m) not in visited), None)
        if current is None:
            return None
        visited.add(id(current))
        to_add = []
        for dep in edges[id(current)]:
            if dep is module:
                return current
            to_add.append(dep)
        to_search.extend(to_add)


def plan_execution(output, all_modules):  # synth.rs:128-212
    edges, visited = {}, set()
    to_search = list(all_modules) + [output]
    while to_search:
        module = to_search.pop()
        if id(module) in visited:
            continue
        visited.add(id(module))
        srcs = []
        for inp in get_inputs(module):
            if inp is not None:
                to_search.append(inp[0])
                srcs.append(inp[0])
        edges[id(module)] = srcs
    to_search = list(all_modules) + [output]
    visited = set()
    removed = []
    while to_search:
        module = to_search.pop()
        if id(module) in visited:
            continue
        visited.add(id(module))
        to_search.extend(edges[id(module)])
        while True:
            frm = is_loop(module, edges)
            if frm is None:
                break
            edges[id(frm)] = [m for m in edges[id(frm)] if m is not module]
            removed.append((frm, module))
    visited = set()
    plan = []
    while True:
        node = next((m for m in all_modules if id(m) not in visited and all(id(d) in visited for d in edges[id(m)])), None)
        if node is None:
            break
        visited.add(id(node))
        plan.append(node)
    return plan, removed


def execute(plan):  # synth.rs:97-101
    for m in plan:
        m.calc()


def render(modules, n_samples, cfg, tap=None):
    """Offline counterpart of the audio callback (main.rs:59-90). -> [channels][n_samples]."""
    output = next((m for m in modules if isinstance(m, Output)), None)  # ui.rs:84-96
    plan, _ = plan_execution(output, modules) if output is not None else ([], [])
    B = cfg["buffer_size"]
    out = np.zeros((cfg["channels"], n_samples), dtype=f32)
    tapped = np.zeros(n_samples, dtype=f32) if tap is not None else None
    for base in range(0, n_samples, B):
        execute(plan)
        n = min(B, n_samples - base)
        if output is not None:
            for c in range(cfg["channels"]):
                out[c, base:base + n] = output.outs[c][:n]
        if tap is not None:
            tapped[base:base + n] = tap[0].outs[tap[1]][:n]
    return (out, tapped) if tap is not None else out
