"""Adapter giving oracle/srack_numpy.py the add_module / connect / set_field graph API."""
import numpy as np

from oracle import srack_numpy as N

_OSC = {0: "val", 1: "antialiasing", 2: "pos"}
_VCF = {0: "freq", 1: "res", 2: "exp_amt"}
_ADSR = {0: "a_sec", 1: "d_sec", 2: "s_val", 3: "r_sec", 4: "phase", 5: "mode", 6: "r_val", 7: "from_a_val", 8: "sample_rate"}


class NumpyGraph:
    def __init__(self, sample_rate=48000, buffer_size=1024, channels=2):
        self.cfg = dict(sample_rate=sample_rate, buffer_size=buffer_size, channels=channels)
        self.modules = []

    def add_module(self, mtype):
        self.modules.append(N.CLASSES[mtype](self.cfg))
        self.modules[-1].module_index = len(self.modules) - 1  # (NoiseModule keys its stream by it)
        return len(self.modules) - 1

    def set_noise_seed(self, seed, voice=0):
        self.cfg["noise_seed"], self.cfg["noise_voice"] = int(seed), int(voice)

    def connect(self, src, src_port, sink, sink_port):
        self.modules[sink].inputs[sink_port] = (self.modules[src], src_port)

    def set_field(self, module, field, value):
        m = self.modules[module]
        if isinstance(m, N.Oscillator):
            if field == 2:
                m.pos = float(value)
            elif field == 1:
                m.antialiasing = bool(value)
            elif field == 3:
                m.sync.last = bool(value)
            else:
                m.val = np.float32(value)
        elif isinstance(m, N.MoogFilter):
            setattr(m, _VCF[field], np.float32(value))
        elif isinstance(m, N.ADSR):
            if field == 5:
                m.mode = int(value)
            elif field == 9:
                m.td.last = bool(value)
            else:
                setattr(m, _ADSR[field], np.float32(value))
        elif isinstance(m, N.VCA):
            m.negative = bool(value)
        elif isinstance(m, N.MonoMixer):
            m.gain[field] = np.float32(value)
        elif isinstance(m, N.Math):
            if field == 0:
                m.constant = np.float32(value)
            else:
                m.operation = int(value)
        elif isinstance(m, N.GridSequencer):
            if field == 0:
                m.steps_per_octave = int(value)
            elif field == 1:
                m.octaves = int(value)
            elif field == 2:
                m.sequence = (m.sequence + [None] * 64)[:int(value)]
            elif field == 3:
                m.current_step = int(value)
            elif field == 4:
                m.td.last = bool(value)
            elif field == 5:
                m.sync_td.last = bool(value)
            else:
                m.last = np.float32(value)
        elif isinstance(m, N.PatternSequencer):
            if field == 0:
                m.sequence = [(ch + [None] * 64)[:int(value)] for ch in m.sequence]
            elif field == 1:
                m.current_step = int(value)
            elif field == 2:
                m.td.last = bool(value)
            else:
                m.sync_td.last = bool(value)
        elif isinstance(m, N.NonLinear):
            m.constant = np.float32(value)
        elif isinstance(m, N.Sample):
            if field == 0:
                m.sample_rate = np.float32(value)
            elif field == 1:
                m.wave_sample_rate = np.float32(value)
            elif field == 2:
                m.wave_new = bool(value)
            elif field == 3:
                m.pos = np.float32(value)
            elif field == 4:
                m.playing = bool(value)
            else:
                m.td.last = bool(value)
        elif isinstance(m, N.Freeverb):
            name = N.Freeverb.PARAMS[field] + "_ctl"
            setattr(m, name, bool(value) if field == 1 else float(value))
        else:
            raise ValueError("no fields")

    def set_step(self, module, channel, step, state, value=0):
        m = self.modules[module]
        if isinstance(m, N.GridSequencer):
            if step < len(m.sequence):
                m.sequence[step] = None if state == 0 else (int(value), state == 2)
        else:
            if step < len(m.sequence[channel]):
                m.sequence[channel][step] = None if state == 0 else (state == 2)

    def set_wave(self, module, samples, sample_rate):
        self.modules[module].load(samples, sample_rate)

    def plan(self):
        output = next(m for m in self.modules if isinstance(m, N.Output))
        plan, removed = N.plan_execution(output, self.modules)
        idx = {id(m): i for i, m in enumerate(self.modules)}
        return [idx[id(m)] for m in plan], [(idx[id(a)], idx[id(b)]) for a, b in removed]

    def render(self, n_samples, tap=None):
        t = (self.modules[tap[0]], tap[1]) if tap is not None else None
        return N.render(self.modules, n_samples, self.cfg, tap=t)
