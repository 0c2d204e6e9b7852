#!/usr/bin/env python3
"""bench.py — voice-samples/sec of the batch render on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg3_poly|cfg2|cfg4|cfg4_b1024|p3|p4] [--flags F]   (F: 1 exact, 64 keep the fast forms ...)

Workloads (BASELINE.json `configs`; SURVEY 8(d) spells them out):
    cfg3 (default; = config 5 at 8 GPUs)  patch P1 saw VCO -> ladder VCF -> VCA, ADSR gated by an LFO square; 262 144 voices per GPU,
                 per-voice randomised detune / cutoff — the configuration the metric is quoted on
    cfg3_poly    the same patch and voice count with NOTHING voice-invariant: per-voice gate-LFO rate and per-voice envelope times on top of
                 detune / cutoff (every voice its own notes: real polyphony) — the whole of P1 evaluated per voice, one voice per lane
    cfg2         the same patch, 4 096 IDENTICAL voices (plumbing: everything is voice-invariant, one latency chain)
    cfg4         patch P2, 2-operator FM with a feedback edge, 65 536 voices, buffer_size 1 (z^-1 held in a register)
    cfg4_b1024   the same at the app's buffer_size 1024 (the feedback delay is a ring in HBM)
    p3           the sequencer-driven patch of scope row (f)1, two output planes (diagnostic)
    p4           the clocked sample player with vibrato and waveshaper of scope row (f)4, two output planes (diagnostic)
A step = one srack_render() of all this rank's voices for T samples (frames resident in HBM) + the mix-down (+ for N > 1
the RCCL sum of the [2][T] partial mixes to rank 0, through the product's own communicator: srack_dist_init /
srack_dist_reduce_mix).  Voices are sharded by global voice index, no exchange during the render => weak scaling.

N > 1: either a launcher provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run), or —
when WORLD_SIZE is not set — bench.py starts the N ranks itself (one per device) and relays rank 0's JSON line.
The control plane (barrier, max over ranks, handing the RCCL unique id around) is torch.distributed/gloo on CPU tensors;
the data path never touches it.

The default run (cfg3, 1 GPU, default sizes) ALSO times, after the headline steps, SIDE_STEPS steps each after one warm-up, same bracketing
(device sync both sides): the headline workload in the exact render mode (`cfg3_exact`: the reference's own arithmetic, bit for bit), the fully
per-voice variant (`cfg3_poly`), config 2 (4 096 identical voices), config 4 (P2 FM pair, 65 536 voices) as the library renders it by default — the
modulator inside its feedback loop exact as a whole (csrc/approx.cpp: a loop through a pitch has no error bound; profiles/r05_horizon.json) — and
with the fast kernels a host may ask for (`cfg4_fast`: SRACK_RENDER_KEEP_DEFAULT, inside the contract for ~30 s of audio), both again at the app's
buffer_size 1024 (`cfg4_b1024`, `cfg4_b1024_fast`), and the workloads of scope rows (f)1 and (f)4 (`p3`, `p4`: two output planes each).  They ride on
the same line as four flat scalars each inside `roofline` (`<name>_ms_per_step` / `_frac_hbm` / `_frac_hbm_kernel` / `_kernel_ms`); each one's
own full bench line (`configs`) and the long prose go to bench_detail.json beside this file and to stderr, NOT onto the line: the driver's record keeps
the last 8 KB of stdout, the line is held under 6 KB (shape_line; tests/test_dist.py).  metric / value / config / roofline.frac stay the
headline's.  `--no-side-configs` skips them.

Prints ONE JSON line (rank 0): metric/value/unit per the driver's contract, plus
  roofline     — achieved = algorithmic bytes of a step / step time (4 B x planes x V x T / t_render, per GPU),
                 peak 8 TB/s HBM; `frac_kernel` is the same for the dominant kernel alone (HIP events on the kernel's own
                 stream, read back through srack_render_kernel_ms — the figure rocprofv3's per-kernel average reproduces)
  cpu_baseline — the C oracle in the reference's structure (block-major execute, one object graph per voice, all
                 oscillator ports computed) on the host cores, bounded sample (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL across processes fails in hipIpcGetMemHandle (set before HIP starts in
# any rank, whoever launched it — a launcher's environment usually has it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured streaming ceiling ~6290
HBM_STREAM_GBS = 6290.0
BYTES_PER_VOICE_SAMPLE = 4   # one f32 frame per voice per sample per distinct output plane
# Vector f64 issue ceiling: 256 CUs x 4 SIMDs x 16 lanes per clock (half the f32 rate) x 2.4 GHz = 39.3e12 lane-operations/s
# (the spec's 78.6 TFLOP/s counts an FMA as two).
F64_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9
# f64-rate VALU instructions per voice-sample in the loop body that render_fm_pair<false, 3> runs for BASELINE config 4's draw
# (feedback gain 0.1 ... 0.4, index 0.5 ... 1.5: every wave proves "increments finite, modulator CV within 1/2, carrier CV within 2"
# and takes the loop with val folded into a per-voice scale, no range reduction in the modulator's 2^x, (2^(cv/4))^4 in the carrier's and
# one-instruction phase wraps): 48, counted in the gfx950 ISA (tools/disasm.sh; the mix is in NOTES.md section 4.  The literal loop —
# nothing proved — has 58, round 2's first kernel had 60).  tools/ubench.hip measures the pipe itself: v_fma_f64 saturates at
# 33.3 T lane-ops/s on this part (8 waves per SIMD), and ONE wave per SIMD — all that 65 536 voices give — reaches 25-30 T with 4-8
# independent chains.
FM_PAIR_F64_OPS = {"render_fm_pair": 48, "render_fm_pair_ring": 48,
                   # config 4 through the kernel specialised at run time (the default since round 3): the generator's vote picks the same loop —
                   # modulator small, carrier (2^(cv/4))^4 — and tools/disasm_jit.py p2 counts the same 384 f64-rate instructions per 8 samples
                   "render_specialized": 48,
                   # the time-parallel pair (buffer_size 256 ... 1024): 809 f64-rate instructions per 16 samples and lane in the copy config 4's draw
                   # takes (tools/disasm.sh ... render_fm_pair_block: 538 between the two barriers of a chunk, 271 after) — the same polynomials, two
                   # additions and a fract per phase instead of one and one, the slice scan, less the carrier sine's f64 fold
                   "render_fm_pair_block": 50.6,
                   # round 6, config 4 as default mode renders it (the modulator exact as a whole): the z^-1 pair on two waves per 64 voices — 336 f64-rate
                   # instructions per 8 samples in the modulator wave's speculative tile (the libm's 2^e: 18, the correctly rounded quotient: 4, the
                   # sine, its fold and the two-conversion decision: 18, the phase: 2), 176 + 8 in the carrier wave's loop for config 4's class ((2^(cv/4))^4; the
                   # 8: its series at degree 9 since the minute's curve, notes/r06.md R6.12) ...
                   "render_fm_pair_x": 65,
                   # ... and across time lanes at the app's block size: 62 per voice-sample in the chunk loop's fast copy (tools/disasm.sh) + the scan's
                   # add and fract on a half-filled wave (4 lane-slots per voice-sample) + 1 (the carrier's series at degree 9)
                   "render_fm_pair_block_x": 67}
F64_LANE_OPS_MEASURED = 33.3e12

WORKLOADS = ("cfg3", "cfg3_poly", "cfg2", "cfg4", "cfg4_b1024", "p3", "p4")


# ---------------------------------------------------------------------------------------------------------------------
# launcher: bench.py --gpus N without an external launcher
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, cmd, env=None, timeout=None):
    """Starts `cmd` n_ranks times (RANK = LOCAL_RANK = 0 .. n-1, WORLD_SIZE = n, rendezvous on 127.0.0.1), relays rank 0's
    stdout to ours and every rank's stderr to ours.  Returns the first non-zero exit code (the other ranks are then stopped), else 0."""
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    procs = []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc, out0 = 0, b""
    deadline = None if timeout is None else time.time() + timeout
    try:
        pending = set(range(n_ranks))
        while pending:
            for r in sorted(pending):
                try:
                    if r == 0:
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or b""
                    else:
                        procs[r].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0 and rc == 0:
                    rc = procs[r].returncode
            if rc != 0 or (deadline is not None and time.time() > deadline):
                rc = rc or 124
                break
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.kill()
                p.wait()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# control plane: barrier / max over ranks / one broadcast, on CPU tensors (gloo)
# ---------------------------------------------------------------------------------------------------------------------
class ControlPlane:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.dist = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the ranks meet on the loopback interface (a container's hostname may not resolve)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x):
        """every rank's x, in rank order, on every rank"""
        if not self.dist:
            return [x]
        import torch
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, torch.tensor([x], dtype=torch.float64))
        return [float(t.item()) for t in out]

    def bcast_bytes(self, payload, n):
        """rank 0's `payload` (n bytes) on every rank"""
        if not self.dist:
            return payload
        import torch
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
        self.dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# the product path on one rank
# ---------------------------------------------------------------------------------------------------------------------
class HipBackend:
    """One rank of the render: patch + voices on this rank's GPU, frames / mix buffers in HBM, the product's RCCL communicator."""

    name = "hip"

    def __init__(self, args, world, rank, local_rank, cp):
        import torch
        import srack_pkg
        self.torch = torch
        self.S = S = srack_pkg.load()
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        if S.lib.srack_device_set(local_rank) != 0:  # the library renders on the calling thread's current device: say it in its own words too
            raise SystemExit("srack_device_set(%d): %s" % (local_rank, S.lib.srack_last_error().decode(errors="replace")))
        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        V, T, C = args.voices, args.samples, 2
        p, self.what, B = self.make_patch()
        self.p = p
        self.buffer_size = B
        self.n_planes, _ = p.planes()
        self.frames = None if args.no_frames else torch.empty((self.n_planes, T, V), dtype=torch.float32, device=self.dev)
        self.mix = torch.empty((C, T), dtype=torch.float32, device=self.dev)
        self.stream = torch.cuda.current_stream(self.dev)
        p.reserve(T, want_mix=not args.no_mix, flags=args.flags)  # set-up, like the allocations above: flatten, upload, scratch buffers
        self.comm = None
        self.ranks_seen = 1
        if world > 1 or args.force_dist:
            uid = cp.bcast_bytes(S.MixComm.unique_id() if rank == 0 else None, S.DIST_ID_BYTES)
            self.comm = S.MixComm(uid, world, rank)
            self.ranks_seen = self.comm.count()

    def make_patch(self):
        """This rank's shard of the workload as a fresh patch: (patch, description, buffer_size).  Voices are drawn by GLOBAL voice index
        (rank * V + v), so a shard is the same voices whatever the number of ranks."""
        S, args = self.S, self.args
        V = args.voices
        w = args.workload
        first = self.rank * V  # global voice index => same draw as the 1-GPU run of the same voices
        B, build, overrides = S.bench_workload(w, V, first_voice=first)   # (s-rack_amd/workloads.py: the workloads as data — the parity tests build the same)
        p = S.Patch(48000, B, 2)
        ids = build(p)
        p.configure_voices(V)
        for m, f, v in overrides(ids):
            p.set_voice_field(m, f, v)
        what = {
            "cfg3_poly": (f"config 3's patch and size with nothing voice-invariant: P1, {V} voices/GPU with per-voice detune / cutoff / gate-LFO rate "
                          "(1.7 ... 13.8 Hz) / envelope times — saw VCO, gate LFO, ladder VCF, ADSR and VCA all evaluated per voice"),
            "cfg3": ("BASELINE config 3 per GPU (config 5 at 8 GPUs): patch P1 saw VCO->ladder VCF->ADSR->VCA, "
                     f"{V} voices/GPU with per-voice randomised detune/cutoff"),
            "cfg2": (f"BASELINE config 2: patch P1, {V} IDENTICAL voices (every module is voice-invariant: one wave evaluates the patch "
                     "once — a latency chain — and the frames are a broadcast)"),
            "cfg4": (f"BASELINE config 4: patch P2 2-op FM with a feedback edge, {V} voices with per-voice feedback / index, "
                     f"buffer_size {B} (z^-1 feedback in a register)"),
            "cfg4_b1024": (f"BASELINE config 4: patch P2 2-op FM with a feedback edge, {V} voices with per-voice feedback / index, "
                           f"buffer_size {B} (the app's block size: the feedback delay is a ring in HBM)"),
            "p4": ("patch P4 (scope row (f)4): clock -> sample player with per-voice vibrato depth -> sign-preserving waveshaper with a per-voice "
                   f"exponent, raw sample on channel 2, {V} voices/GPU"),
            "p3": ("patch P3 (scope row (f)1): clock -> grid + pattern sequencers -> per-voice transposed saw VCO -> VCF swept by an "
                   f"envelope -> VCA, raw gate on channel 2, {V} voices/GPU with per-voice transpose/cutoff"),
        }[w]
        return p, what, B

    def dump(self, directory):
        """Test hook (SRACK_BENCH_DUMP, tests/test_gpu_dist.py): a CHECKED render beside the timed ones.  A fresh patch of this rank's
        shard is rendered from sample 0 through the same entry points and the same communicator; what a checker needs is left in
        `directory`/rank<r>.npz: the device this rank rendered on, sampled voices' frames (first and last of the shard among them), the
        f64 sum over all of the shard's voices, the partial mix, and — rank 0 — the mix after the RCCL reduce."""
        torch, a = self.torch, self.args
        V, T = a.voices, a.samples
        p, _, _ = self.make_patch()
        frames = torch.empty((self.n_planes, T, V), dtype=torch.float32, device=self.dev)
        mix = torch.empty((2, T), dtype=torch.float32, device=self.dev)
        p.render_raw(T, frames.data_ptr(), mix.data_ptr(), a.flags, self.stream.cuda_stream)
        self.sync()
        partial = mix.cpu().numpy().copy()
        reduced = None
        if self.comm is not None:
            self.comm.reduce_mix(mix.data_ptr(), mix.numel(), 0, self.stream.cuda_stream)
            self.sync()
            reduced = mix.cpu().numpy()
        idx = np.unique(np.concatenate([[0, V - 1], np.linspace(0, V - 1, 6).astype(np.int64)]))
        hip_device, bus = self.S.device_get()  # the device the LIBRARY renders on, in its own words
        np.savez(os.path.join(directory, f"rank{self.rank}.npz"), rank=self.rank, world=self.world, local_rank=self.local_rank, hip_device=hip_device,
                 pci_bus_id=bus, ranks_seen=self.ranks_seen, voices=idx, frames=frames[:, :, torch.from_numpy(idx).to(self.dev)].cpu().numpy(),
                 frames_sum_f64=frames.double().sum(dim=2).cpu().numpy(), frames_abs_sum_f64=frames.double().abs().sum(dim=2).cpu().numpy(),
                 partial_mix=partial, reduced_mix=reduced if reduced is not None else partial, planes=np.array(p.planes()[1]), info=p.info())

    def step(self):
        a = self.args
        if a.block:  # diagnostic: the step as the app's tick loop would drive it — one call per `block` samples (main.rs:59-63)
            V, T, st = a.voices, a.samples, self.stream.cuda_stream
            if self.n_planes > 1 and self.frames is not None:
                raise SystemExit("--block: frames of a multi-plane patch are [planes][n][V] per call (use --no-frames)")
            for t in range(0, T, a.block):
                n = min(a.block, T - t)
                self.p.render_raw(n, self.frames.data_ptr() + 4 * t * V if self.frames is not None else None,
                                  None if a.no_mix else self.mix.data_ptr() + 4 * 2 * t, a.flags, st)  # this call's mix is [2][n] at 2 * t
        else:
            self.p.render_raw(a.samples, self.frames.data_ptr() if self.frames is not None else None, None if a.no_mix else self.mix.data_ptr(),
                              a.flags, self.stream.cuda_stream)
        if self.comm is not None and not a.no_mix:
            self.comm.reduce_mix(self.mix.data_ptr(), self.mix.numel(), 0, self.stream.cuda_stream)  # RCCL over xGMI: [2][T] f32 partial mixes

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def arm_kernel_timer(self):
        self.p.kernel_ms(reset=True)

    def kernel_ms(self):
        return self.p.kernel_ms(reset=True)

    def info(self):
        return self.p.info()

    def close(self):
        if self.comm is not None:
            self.sync()
            self.comm.destroy()
            self.comm = None


def cpu_baseline(S, workload, n_samples=48000):
    """The oracle timed on this host: B1 — `cores` threads x a few voices x 1 s of the workload's patch, best of 3 (the
    `value`: about 10-30 s of CPU work in total across cores, wall time a few seconds) — and B0 — ONE voice on ONE thread, the faithful analogue
    of the reference's single audio thread (main.rs:59-63), best of 5."""
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    fm = workload in ("cfg4", "cfg4_b1024")
    voices_per_core = 2 if fm else 6
    V = cores * voices_per_core
    B = 1 if workload == "cfg4" else 1024
    g = O.OraclePatch(48000, B, 2)
    if fm:
        ids = S.build_p2(g)
        beta, index = S.p2_voice_params(V)
        ov = [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)]
        patch = "P2 (cfg4 draw)"
    elif workload == "cfg3_poly":
        ids = S.build_p1(g)
        ov = S.p1_poly_overrides(ids, S.p1_poly_voice_params(V))
        patch = "P1 (cfg3_poly draw)"
    else:
        ids = S.build_p1(g)
        det, cut = S.p1_voice_params(V)
        ov = [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)] if workload != "cfg2" else []
        patch = "P1 (cfg3 draw)" if workload != "cfg2" else "P1 (identical voices, each evaluated: the reference has no notion of sharing)"
    g.render_batch(min(V, cores), 4800, [], frames=False, mix=True, threads=cores)  # warm the threads
    best = None
    for _ in range(3):
        t = time.perf_counter()
        g.render_batch(V, n_samples, ov, frames=False, mix=True, threads=cores)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    single = None
    for _ in range(5):  # B0: one voice, one thread, best of 5
        t = time.perf_counter()
        g.render_batch(1, n_samples, [(m, f, v[:1]) for m, f, v in ov], frames=False, mix=True, threads=1)
        dt1 = time.perf_counter() - t
        single = dt1 if single is None else min(single, dt1)
    return {
        "value": V * n_samples / best, "unit": "voice-samples/s", "cores": cores, "kind": "port",
        "sample": f"B1: {V} voices x {n_samples} samples of patch {patch}, buffer_size {B}, {cores} threads, best of 3",
        "per_core_value": V * n_samples / best / cores,
        "single_thread_value": n_samples / single,
        "single_thread_sample": f"B0 (BASELINE.md section 2): 1 voice x {n_samples} samples, 1 thread, best of 5",
        "single_thread_x_realtime": n_samples / single / 48000.0,
        "note": "C restatement of the reference tick (oracle/srack_oracle.c); the Rust reference cannot be built here. "
                "Omits the reference's per-block RwLock/Arc/Vec overhead, so it is a slightly optimistic stand-in.",
    }


# (name on the line, workload, render flags): the headline in the exact render mode (the reference's arithmetic bit for bit), the fully
# per-voice variant of the headline, BASELINE.json configs[1] and configs[3] — the other single-GPU configurations —
KEEP_DEFAULT = 64  # SRACK_RENDER_KEEP_DEFAULT
SIDE_CONFIGS = (("cfg3_exact", "cfg3", 1), ("cfg3_poly", "cfg3_poly", 0), ("cfg2", "cfg2", 0),
                # config 4 as the library renders it by default — its modulator exact as a whole: the feedback loop runs through a pitch, where only
                # the reference's own bits follow the reference for longer than seconds (csrc/approx.cpp; profiles/r05_horizon.json) — and
                # with the fast kernels a host may ask for (KEEP_DEFAULT: within the contract for ~30 s); the same at the app's block size
                ("cfg4", "cfg4", 0), ("cfg4_fast", "cfg4", KEEP_DEFAULT), ("cfg4_b1024", "cfg4_b1024", 0), ("cfg4_b1024_fast", "cfg4_b1024", KEEP_DEFAULT),
                # the workloads of scope rows (f)1 and (f)4 (two output planes each)
                ("p3", "p3", 0), ("p4", "p4", 0))
SIDE_STEPS, SIDE_WARMUP = 5, 1


def side_config(args, workload, flags=0):
    """Another single-GPU configuration, timed like the headline: SIDE_WARMUP untimed steps, then exactly SIDE_STEPS
    steps between two device syncs; the dominant kernel by HIP events beside it.  Returns the figures of its own bench line."""
    a = argparse.Namespace(**vars(args))
    a.workload, a.voices, a.flags, a.force_dist, a.no_frames, a.no_mix = workload, default_voices(workload), flags, False, False, False
    be = HipBackend(a, 1, 0, int(os.environ.get("LOCAL_RANK", "0")), None)
    try:
        for _ in range(SIDE_WARMUP):
            be.step()
        be.sync()
        be.arm_kernel_timer()
        t0 = time.perf_counter()
        for _ in range(SIDE_STEPS):
            be.step()
        be.sync()
        step_s = (time.perf_counter() - t0) / SIDE_STEPS
        kernel_ms, n_launch = be.kernel_ms()
        V, T = a.voices, a.samples
        info = be.info()
        kname = info.split("kernel=")[-1] if "kernel=" in info else ""
        launches = max(1, n_launch // SIDE_STEPS)
        bytes_per_step = BYTES_PER_VOICE_SAMPLE * be.n_planes * V * T
        out = {"workload": be.what, "render_flags": flags, "arithmetic": arithmetic_note(flags, info), "voices": V, "samples_per_step": T, "buffer_size": be.buffer_size,
               "steps": SIDE_STEPS, "warmup": SIDE_WARMUP,
               "ms_per_step": step_s * 1e3, "voice_samples_per_s": V * T / step_s,
               "frac_hbm": bytes_per_step / step_s / 1e9 / HBM_PEAK_GBS, "kernel": kname, "kernel_ms": kernel_ms, "launches_per_step": launches,
               "frac_hbm_kernel": (bytes_per_step / launches / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms > 0 else 0.0, "program": info}
        # (the ISA counts are those of the fast kernels' loops: KEEP_DEFAULT; the exact flavour runs the libm's pow and ocml's sin)
        ops = FM_PAIR_F64_OPS.get(kname) if (kname in ("render_fm_pair_x", "render_fm_pair_block_x") or ((flags & KEEP_DEFAULT) and (workload == "cfg4" or kname != "render_specialized"))) else None
        if ops:
            out.update({"f64_ops_per_voice_sample": ops, "frac_valu_f64": ops * V * T / step_s / F64_LANE_OPS_PEAK,
                        "frac_of_measured_f64_rate": ops * V * T / step_s / F64_LANE_OPS_MEASURED})
        return out
    finally:
        be.close()
        be.frames = be.mix = be.p = None
        be.torch.cuda.empty_cache()


def profiled_traffic(kernel_name, workload, flags, V, T):
    """HBM bytes per launch of `kernel_name` from the newest committed rocprofv3 PMC summary (profiles/rNN<tag>_summary.json, written
    by profiles/summarize.py from separate WRITE_SIZE / FETCH_SIZE passes of this very command), or None.  NOT measured in
    this run — a profiler cannot run inside the timed region — and only valid for the default size of the workload."""
    import glob, re
    if (V, T) != (default_voices(workload), 48000):
        return None
    tag = {("cfg3", 0): "", ("cfg3", 1): "_exact", ("cfg3", 2): "_special", ("cfg3", 3): "_special_exact", ("p3", 0): "_p3", ("p3", 1): "_p3_exact",
           ("cfg4", 0): "_cfg4", ("cfg4", 64): "_cfg4_fast", ("cfg4_b1024", 64): "_cfg4_b1024_fast", ("cfg4", 2): "_cfg4_special", ("cfg4_b1024", 0): "_cfg4_b1024", ("cfg4_b1024", 2): "_cfg4_b1024_special", ("cfg2", 0): "_cfg2", ("p4", 0): "_p4", ("cfg3_poly", 0): "_poly"}.get((workload, flags))
    if tag is None:
        return None
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_summary.json"))):
        base = os.path.basename(path)
        if not re.fullmatch(r"r\d\d" + re.escape(tag) + r"_summary\.json", base):
            continue
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, v in d.get("derived", {}).items():
            # (a kernel specialised at run time appears under its entry point's name in the profiler; its control-only launches are a group
            # of their own — same name, another grid —: the voice launches are the group that moves the bytes)
            if kernel_name and (kernel_name in name or (kernel_name == "render_specialized" and name.startswith("srk_voice"))) and "hbm_traffic_bytes" in v:
                if best is None or best["source"] != "profiles/" + base or v["hbm_traffic_bytes"] > best["bytes_per_launch"]:
                    best = {"bytes_per_launch": v["hbm_traffic_bytes"], "write": v.get("hbm_write_bytes"),
                            "read_corrected": v.get("hbm_read_bytes_gfx950_corrected"), "source": "profiles/" + base}
    return best


def arithmetic_note(flags, info=""):
    """What the render mode computes in, where it is not the reference's own operation sequence.  `info`: srack_render_info —
    "approx[exact: ...]" where the flattener's error bound gave the patch the exact flavour although default mode was asked for."""
    if "; exact osc " in info and not flags & 1:
        return ("default mode with oscillator(s) " + info.split("; exact osc ")[1].split("]")[0] + " evaluated exactly as a whole — 2^cv by the host libm's pow operation for operation, the "
                "reference's sine, f64 PolyBLEP: bit-identical to the CPU tick — because the flattener's error bound finds an unbounded gain behind them (csrc/approx.cpp: "
                "config 4's feedback loop runs through a pitch); everything else: " + arithmetic_note(flags))
    if flags & 1 or "approx[exact" in info:
        return (("exact flavour by the flattener's error bound (csrc/approx.cpp: " + info.split("approx[exact: ")[1].split("]")[0] + "): " if "approx[exact" in info else "") + "exact mode: the reference's operations one by one — f64 phase / 2^cv (the host libm's pow, operation for operation) / PolyBLEP with its f64 division, the ladder "
                "uncontracted with min/max clamps; frames bit-identical to the CPU tick (oscillator.rs:108-158, filter.rs:58-92)")
    return ("default mode, within the 1e-5 contract but NOT the reference's arithmetic everywhere: PolyBLEP evaluated in f32 (reference: f64, "
            "oscillator.rs:50-67); the ladder with one product of each a*b - c*d folded into an fma and v_med3 clamps (reference: uncontracted, "
            "min/max, filter.rs:69-89); the audio saw's phase accumulator in 2^-64 fixed point where the flattener proves nothing integrates it "
            "(reference: f64 with fmod; gate-producing oscillators keep the f64 phase); where a pitch CV is connected: 2^cv by a polynomial (3e-16; 1e-12 inside proved FM loops; "
            "the libm's own pow per held value for a CV that holds: sequencer notes, envelopes), sine by a polynomial (f64, one rounding). "
            "ADSR, VCA, mixer, math, sequencers: the reference's f32 operations in its order, bit-identical in every mode. "
            "`cfg3_exact_*` on this line is the same workload in the reference's own arithmetic")


LINE_BUDGET = 6000  # bytes: the driver's record keeps the last 8 KB of stdout — the ONE line has to fit whole (tests/test_dist.py holds it to this)
SIDE_KEYS = ("ms_per_step", "frac_hbm", "frac_hbm_kernel", "kernel_ms")  # what a side configuration puts on the line; the rest is in bench_detail.json


def shape_line(out):
    """(line, detail): `detail` is everything measured; `line` is what is printed — the contract's keys, `roofline` (the headline's, plus four
    flat scalars per side configuration), `cpu_baseline` — with the prose stated once and briefly.  Whatever else was measured (`configs`: every
    side configuration's own full bench line; the long `arithmetic` paragraphs; the traffic's detail) goes to bench_detail.json / stderr.
    Holds the line to LINE_BUDGET bytes: if it is still longer, optional keys go, least important first, and the line says which."""
    import copy
    detail = out
    line = copy.deepcopy(out)
    line.pop("configs", None)
    cfg = line.get("config", {})
    if len(cfg.get("arithmetic", "")) > 200:
        cfg["arithmetic"] = cfg["arithmetic_short"] if "arithmetic_short" in cfg else cfg["arithmetic"][:197] + "..."
    cfg.pop("arithmetic_short", None)
    detail.get("config", {}).pop("arithmetic_short", None)
    rf = line.get("roofline", {})
    td = rf.pop("traffic_detail", None)
    if td:
        rf["traffic_source"] = td.get("source")
    side = set()
    for k in list(rf):
        for w, _, _ in SIDE_CONFIGS:
            if k.startswith(w + "_") and (k[len(w) + 1:] in SIDE_KEYS or k[len(w) + 1:] in ("frac_of_measured_f64_rate", "f64_ops_per_voice_sample")):
                side.add(k)
    for k in list(rf):  # side scalars outside SIDE_KEYS stay in the detail only
        if k not in side and any(k.startswith(w + "_") for w, _, _ in SIDE_CONFIGS) and k != "cfg3_ticked_1024_ms_per_step":
            del rf[k]
    for k, v in list(rf.items()):
        if isinstance(v, float):
            rf[k] = float("%.6g" % v)
    cb = line.get("cpu_baseline")
    if cb:
        cb.pop("single_thread_sample", None)
        cb["note"] = "C restatement of the reference tick (oracle/srack_oracle.c); the Rust reference cannot be built here"
    line["detail"] = "bench_detail.json"
    dropped = []
    for path in (("roofline", "definition"), ("config", "program"), ("cpu_baseline", "note"), ("config", "arithmetic"), ("roofline", "note"),
                 ("host_enqueue_ms_per_step",), ("cpu_baseline", "sample"), ("config", "workload")):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k, {})
        if path[-1] in d:
            if path == ("config", "workload"):
                d["workload"] = d["workload"][:160]
            else:
                del d[path[-1]]
            dropped.append(".".join(path))
    if len(json.dumps(line)) > LINE_BUDGET:  # last resort: the side scalars, longest names first (the headline never goes)
        for k in sorted(side, key=len, reverse=True):
            if len(json.dumps(line)) <= LINE_BUDGET:
                break
            rf.pop(k, None)
            dropped.append("roofline." + k)
    if dropped:
        line["dropped_for_length"] = dropped
    return line, detail


def write_detail(detail):
    """bench_detail.json beside bench.py (or $SRACK_BENCH_DETAIL), and the same on stderr — never on stdout."""
    text = json.dumps(detail, indent=1)
    path = os.environ.get("SRACK_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    try:
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)
    print("[bench] detail:\n" + text, file=sys.stderr, flush=True)


def arithmetic_short(flags, info=""):
    """arithmetic_note in one sentence (the line carries this; the paragraph is in bench_detail.json and DESIGN.md section 2)"""
    if "; exact osc " in info and not flags & 1:
        return "default mode (f32 PolyBLEP, contracted ladder, fixed-point saw phase: within 1e-5), oscillator(s) " + info.split("; exact osc ")[1].split("]")[0] + " exact as a whole; DESIGN.md section 2"
    if flags & 1 or "approx[exact" in info:
        return "exact mode: the reference's operations one by one, frames bit-identical to the CPU tick; DESIGN.md section 2"
    return ("default mode: within the 1e-5 contract, not the reference's operation sequence everywhere (f32 PolyBLEP, ladder with one fma per a*b-c*d, "
            "2^-64 fixed-point saw phase); cfg3_exact_* is the same workload bit for bit; DESIGN.md section 2")


def default_voices(workload):
    return {"cfg3": 262144, "cfg3_poly": 262144, "p3": 262144, "p4": 131072, "cfg2": 4096, "cfg4": 65536, "cfg4_b1024": 65536}[workload]


# ---------------------------------------------------------------------------------------------------------------------
# the timing protocol (every rank)
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args, backend_cls=HipBackend):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    cp = ControlPlane(world, rank)
    be = backend_cls(args, world, rank, local_rank, cp)

    def fence():
        be.sync()
        cp.barrier()
        be.sync()

    for _ in range(args.warmup):
        be.step()
    fence()
    be.arm_kernel_timer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        be.step()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue K steps (diagnostic: is the host the bottleneck?)
    fence()
    mine = time.perf_counter() - t0
    elapsed = cp.max(mine)
    per_rank_ms = [x / args.steps * 1e3 for x in cp.gather(mine)]
    kernel_ms, n_launch = be.kernel_ms()
    if os.environ.get("SRACK_BENCH_DUMP") and hasattr(be, "dump"):
        be.dump(os.environ["SRACK_BENCH_DUMP"])

    out = None
    if rank == 0:
        V, T = args.voices, args.samples
        n_planes = be.n_planes
        voice_samples = float(world) * V * T * args.steps
        step_s = elapsed / args.steps
        # a step may be several launches of the render kernel (chunks that pipeline against the control program):
        # the kernel-only figures are per launch, like rocprofv3's per-kernel average
        launches_per_step = max(1, n_launch // max(1, args.steps))
        bytes_per_step = BYTES_PER_VOICE_SAMPLE * n_planes * V * T        # per GPU: one f32 per voice-sample per distinct output plane
        bytes_per_launch = bytes_per_step / launches_per_step
        achieved = bytes_per_step / step_s / 1e9
        achieved_kernel = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        info = be.info()
        kname = info.split("kernel=")[-1] if "kernel=" in info else ""
        out = {
            "metric": "voice-samples/sec @48 kHz offline render",
            "value": voice_samples / elapsed,
            "unit": "voice-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "ranks_seen": getattr(be, "ranks_seen", world),
            "per_rank_ms_per_step": per_rank_ms,
            "config": {
                "workload": be.what + f", {T} samples/step @48 kHz, f32 frames [{n_planes}][T][V] in HBM + stereo mix-down"
                            + (" + RCCL reduce of the [2][T] mix (srack_dist_reduce_mix)" if getattr(be, "comm", None) is not None else ""),
                "name": args.workload, "voices_per_gpu": V, "samples_per_step": T, "buffer_size": getattr(be, "buffer_size", 1024),
                "render_flags": args.flags, "backend": be.name, "samples_per_call": args.block or T,
                "arithmetic": arithmetic_note(args.flags, info), "arithmetic_short": arithmetic_short(args.flags, info),
                "frames_written": not args.no_frames, "mix_down": not args.no_mix, "program": info,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "definition": "algorithmic bytes of a step (4 B x planes x voices x samples, per GPU) / step time (max over ranks)",
                "frac_of_measured_stream_ceiling": achieved / HBM_STREAM_GBS,
                "achieved_kernel": achieved_kernel, "frac_kernel": achieved_kernel / HBM_PEAK_GBS,
                "kernel": kname, "kernel_ms": kernel_ms, "kernel_launches": n_launch, "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_step": bytes_per_step,
                "voice_samples_per_s_kernel": V * T / launches_per_step / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
                "traffic": None,
            },
        }
        if args.workload in ("cfg4", "cfg4_b1024") and ((args.flags & KEEP_DEFAULT) or kname in ("render_fm_pair_x", "render_fm_pair_block_x")) and FM_PAIR_F64_OPS.get(kname) and not (kname == "render_specialized" and args.workload != "cfg4"):
            ops = FM_PAIR_F64_OPS[kname]
            lane_ops = ops * V * T / step_s
            out["roofline"].update({
                "bound": "valu_f64", "achieved": lane_ops / 1e12, "peak": F64_LANE_OPS_PEAK / 1e12, "unit": "T f64 lane-ops/s",
                "frac": lane_ops / F64_LANE_OPS_PEAK, "frac_of_measured_f64_rate": lane_ops / F64_LANE_OPS_MEASURED, "f64_ops_per_voice_sample": ops,
                "definition": "f64-rate VALU instructions per voice-sample of the kernel's loop body (ISA count) x voice-samples/s, "
                              "against 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "achieved_kernel": achieved_kernel, "frac_kernel": achieved_kernel / HBM_PEAK_GBS},
            })
        if args.workload == "cfg2":
            out["roofline"]["note"] = ("identical voices: the whole patch is voice-invariant and evaluated once, by a pipeline of one-wave control units (bound by the "
                                       "ladder filter's 13-deep recurrence, ~65 ns per sample); the frames are a broadcast of its track — a few percent of the HBM "
                                       "roofline by construction (plumbing configuration)")
        tr = profiled_traffic(kname, args.workload, args.flags, V, T) if be.name == "hip" and not args.no_frames and not args.no_mix else None
        if tr:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_detail"] = dict(tr, measured_in_this_run=False,
                                                     note="rocprofv3 PMC passes of this command (profiles/run_profile.sh); a profiler cannot run inside the timed region")
        if world > 1:  # a straggler shows here (the value is paced by the slowest rank)
            print("[bench] ms per step by rank: " + " ".join(f"{r}:{m:.3f}" for r, m in enumerate(per_rank_ms)), file=sys.stderr, flush=True)
        if (world == 1 and be.name == "hip" and args.workload == "cfg3" and not args.no_side_configs and args.flags == 0 and not args.block and not args.no_frames
                and not args.no_mix and (V, T) == (default_voices("cfg3"), 48000)):
            # the same workload driven the way the reference's audio callback drives `execute` — one call per buffer_size samples
            # (main.rs:59-63) — instead of one call per second of audio: what a host that ticks pays (tick sessions)
            be.args.block = 1024
            be.step()
            be.sync()
            t_tick = time.perf_counter()
            for _ in range(SIDE_STEPS):
                be.step()
            be.sync()
            out["roofline"]["cfg3_ticked_1024_ms_per_step"] = (time.perf_counter() - t_tick) / SIDE_STEPS * 1e3
            be.args.block = 0
            # the other single-GPU BASELINE configurations on the same line (the headline's buffers are released first)
            be.close()
            be.frames = be.mix = be.p = None
            be.torch.cuda.empty_cache()
            out["configs"] = {}
            for w, wl, fl in SIDE_CONFIGS:
                c = out["configs"][w] = side_config(args, wl, fl)
                rf = out["roofline"]
                rf[w + "_ms_per_step"] = c["ms_per_step"]
                rf[w + "_voice_samples_per_s"] = c["voice_samples_per_s"]
                rf[w + "_frac_hbm"] = c["frac_hbm"]
                rf[w + "_frac_hbm_kernel"] = c["frac_hbm_kernel"]
                rf[w + "_kernel_ms"] = c["kernel_ms"]
                rf[w + "_launches_per_step"] = c["launches_per_step"]
                if "frac_valu_f64" in c:
                    rf[w + "_frac_valu_f64"] = c["frac_valu_f64"]
                    rf[w + "_f64_ops_per_voice_sample"] = c["f64_ops_per_voice_sample"]
                    rf[w + "_frac_of_measured_f64_rate"] = c["frac_of_measured_f64_rate"]
        if world == 1 and not args.no_cpu and args.workload not in ("p3", "p4") and be.name == "hip":
            out["cpu_baseline"] = cpu_baseline(be.S, args.workload)
    be.close()
    cp.close()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=WORKLOADS + ("p1",), default="cfg3", help="cfg3 = the metric (p1 is its old name); see the module docstring")
    ap.add_argument("--voices", type=int, default=0, help="voices per GPU (default: the workload's BASELINE size)")
    ap.add_argument("--samples", type=int, default=48000, help="samples per step (1 s @ 48 kHz)")
    ap.add_argument("--flags", type=int, default=0, help="SRACK_RENDER_* flags (1 exact osc, 2 no fusion, 4 no uniform hoist, 8 no control stages)")
    ap.add_argument("--no-frames", action="store_true", help="mix only (diagnostic; not the metric)")
    ap.add_argument("--no-mix", action="store_true", help="frames only (diagnostic; not the metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-side-configs", action="store_true", help="skip configs 2 and 4 after the headline steps of the default run")
    ap.add_argument("--block", type=int, default=0,
                    help="diagnostic: render each step in calls of this many samples (the app's tick loop: one call per buffer_size), not in one call")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL communicator and run the mix reduce even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    args = ap.parse_args(argv)
    if args.workload == "p1":
        args.workload = "cfg3"
    if args.voices <= 0:
        args.voices = default_voices(args.workload)
    return args


def main(argv=None, backend_cls=HipBackend, self_cmd=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (one rank per device)
        cmd = self_cmd or [sys.executable, os.path.abspath(__file__)]
        return launch_ranks(args.gpus, cmd + argv)
    # Libraries chat on stdout through C stdio (gloo: "[Gloo] Rank 0 is connected to ...", RCCL: its version banner).  stdout is for
    # the ONE JSON line: while the ranks work, file descriptor 1 points at stderr.
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        out = run_rank(args, backend_cls)
    finally:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
    if out is not None:
        line, detail = shape_line(out)
        write_detail(detail)
        print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
