#!/usr/bin/env python3
"""bench.py — voice-samples/sec of the batch render on N MI355X (one process per GPU).

Workload (BASELINE.json config 3 per GPU; config 5 = 8 ranks of it):
    patch P1 (saw VCO -> ladder VCF -> VCA, ADSR gated by an LFO square), 262 144 voices per GPU with
    per-voice randomised detune / cutoff, 1 s @ 48 kHz per step, f32 frames [T][V] + stereo mix-down.
A step = one srack_render() of all voices for T samples (frames resident in HBM) + the mix-down
(+ for N > 1 the RCCL sum of the [2][T] partial mixes to rank 0).  Voices are sharded by global
voice index, no exchange during the render => weak scaling.

Prints ONE JSON line (rank 0): metric/value/unit per the driver's contract, plus
  roofline     — achieved = 4 B x V x T / avg duration of the render kernel (HIP events on the
                 kernel's own stream, read back through srack_render_kernel_ms), peak 8 TB/s HBM
  cpu_baseline — the C oracle in the reference's structure (block-major execute, one object graph
                 per voice, all oscillator ports computed) on the host cores, bounded sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured streaming ceiling ~6290
HBM_STREAM_GBS = 6290.0
BYTES_PER_VOICE_SAMPLE = 4   # one f32 frame per voice per sample


def cpu_baseline(S, voices_per_core=6, n_samples=48000):
    """The oracle timed on this host: `cores` threads x voices_per_core voices x 1 s of P1 (about 10-30 s of CPU work
    in total across cores; wall time a few seconds)."""
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    V = cores * voices_per_core
    g = O.OraclePatch(48000, 1024, 2)
    ids = S.build_p1(g)
    det, cut = S.p1_voice_params(V)
    ov = [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)]
    g.render_batch(min(V, cores), 4800, ov[:0], frames=False, mix=True, threads=cores)  # warm the threads
    best = None
    for _ in range(3):
        t = time.perf_counter()
        g.render_batch(V, n_samples, ov, frames=False, mix=True, threads=cores)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    t = time.perf_counter()
    g.render_batch(voices_per_core, n_samples, [(m, f, v[:voices_per_core]) for m, f, v in ov], frames=False, mix=True, threads=1)
    dt1 = time.perf_counter() - t
    return {
        "value": V * n_samples / best, "unit": "voice-samples/s", "cores": cores, "kind": "port",
        "sample": f"{V} voices x {n_samples} samples of patch P1 (cfg3 draw), buffer_size 1024, {cores} threads, best of 3",
        "single_thread_value": voices_per_core * n_samples / dt1,
        "note": "C restatement of the reference tick (oracle/srack_oracle.c); the Rust reference cannot be built here. "
                "Omits the reference's per-block RwLock/Arc/Vec overhead, so it is a slightly optimistic stand-in.",
    }


def measured_traffic(kernel_name, V, T):
    """HBM bytes per launch of `kernel_name` from the newest committed rocprofv3 PMC summary (profiles/*_summary.json,
    written by profiles/summarize.py from separate WRITE_SIZE / FETCH_SIZE passes of this very command), or None.
    Only valid for the default workload the profile was taken on."""
    import glob
    if (V, T) != (262144, 48000):
        return None
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, v in d.get("derived", {}).items():
            if kernel_name and kernel_name in name and "hbm_traffic_bytes" in v:
                best = {"bytes_per_launch": v["hbm_traffic_bytes"], "write": v.get("hbm_write_bytes"),
                        "read_corrected": v.get("hbm_read_bytes_gfx950_corrected"), "source": os.path.basename(path)}
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--voices", type=int, default=262144, help="voices per GPU")
    ap.add_argument("--samples", type=int, default=48000, help="samples per step (1 s @ 48 kHz)")
    ap.add_argument("--flags", type=int, default=0, help="SRACK_RENDER_* flags (1 exact osc, 2 no fusion, 4 no uniform hoist)")
    ap.add_argument("--no-frames", action="store_true", help="mix only (diagnostic; not the metric)")
    ap.add_argument("--no-mix", action="store_true", help="frames only (diagnostic; not the metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--workload", choices=["p1", "p3"], default="p1",
                    help="p1 = BASELINE config 3 (the metric); p3 = the sequencer-driven patch of scope row (f)1, two output planes (diagnostic)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the mix reduce even with one rank (smoke-tests the N > 1 code path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import srack_pkg
    S = srack_pkg.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    V, T, C = args.voices, args.samples, 2
    p = S.Patch(48000, 1024, C)
    if args.workload == "p1":
        ids = S.build_p1(p)
        p.configure_voices(V)
        det, cut = S.p1_voice_params(V, first_voice=rank * V)  # global voice index => same draw as the 1-GPU run
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        what = "BASELINE config 3 per GPU (config 5 at 8 GPUs): patch P1 saw VCO->ladder VCF->ADSR->VCA"
    else:
        ids = S.build_p3(p)
        p.configure_voices(V)
        u0, u1 = S.voice_uniform(V, 0, first_voice=rank * V), S.voice_uniform(V, 1, first_voice=rank * V)
        p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, (u0 * np.float32(2.5) - np.float32(2.0)).astype(np.float32))
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, (np.float32(0.05) + u1 * np.float32(0.35)).astype(np.float32))
        what = "patch P3 (diagnostic): clock -> grid + pattern sequencers -> per-voice transposed saw VCO -> VCF swept by an envelope -> VCA, raw gate on channel 2"
    n_planes, _ = p.planes()

    frames = None if args.no_frames else torch.empty((n_planes, T, V), dtype=torch.float32, device=dev)
    mix = torch.empty((C, T), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    p.reserve(T, want_mix=not args.no_mix, flags=args.flags)  # set-up, like the allocations above: flatten, upload, scratch buffers

    def step():
        p.render_raw(T, frames.data_ptr() if frames is not None else None, None if args.no_mix else mix.data_ptr(), args.flags, stream.cuda_stream)
        if use_dist and not args.no_mix:
            dist.reduce(mix, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: [2][T] f32 partial mixes

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    p.kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue K steps (diagnostic: is the host the bottleneck?)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, n_launch = p.kernel_ms(reset=True)

    if rank == 0:
        voice_samples = float(world) * V * T * args.steps
        # a step may be several launches of the render kernel (chunks that pipeline against the control program):
        # roofline figures are per launch, like rocprofv3's per-kernel average
        launches_per_step = max(1, n_launch // max(1, args.steps))
        bytes_per_launch = BYTES_PER_VOICE_SAMPLE * n_planes * V * T / launches_per_step  # one f32 per voice-sample per distinct output plane
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "voice-samples/sec @48 kHz offline render",
            "value": voice_samples / elapsed,
            "unit": "voice-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": what + f", {V} voices/GPU with per-voice randomised "
                            + ("detune/cutoff" if args.workload == "p1" else "transpose/cutoff") + f", {T} samples/step @48 kHz, "
                            f"f32 frames [{n_planes}][T][V] in HBM + stereo mix-down" + (" + RCCL reduce of the [2][T] mix" if use_dist else ""),
                "voices_per_gpu": V, "samples_per_step": T, "buffer_size": 1024, "render_flags": args.flags,
                "arithmetic": "f32 wires and modules; oscillator phase accumulator 64-bit: f64 as the reference, 2^-64 fixed point in the default-mode fused saw kernel (DESIGN.md section 3)",
                "frames_written": frames is not None, "mix_down": not args.no_mix, "program": p.info(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_detail": None,
                "frac_of_measured_stream_ceiling": achieved / HBM_STREAM_GBS,
                "kernel_ms": kernel_ms, "kernel_launches": n_launch, "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "voice_samples_per_s_kernel": V * T / launches_per_step / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
            },
        }
        kname = p.info().split("kernel=")[-1] if "kernel=" in p.info() else ""
        tr = measured_traffic(kname, V, T) if args.workload == "p1" and args.flags == 0 and frames is not None and not args.no_mix else None
        if tr:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_detail"] = tr
        if world == 1 and not args.no_cpu and args.workload == "p1":
            out["cpu_baseline"] = cpu_baseline(S)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL prints its banner through C stdio: push it out before the one JSON line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
