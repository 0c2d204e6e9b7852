"""The N > 1 path without GPUs: world_size-2 gloo processes check the sharding contract bench.py relies on.

What is multi-GPU about this path: voices shard by GLOBAL voice index with no exchange during the
render; the only collective is the sum of the per-rank partial mixes.  Here each rank renders its shard
with the CPU oracle (test infrastructure standing in for the GPU), draws its per-voice parameters exactly
like bench.py does, reduces the [2][T] partial mix to rank 0 over torch.distributed (gloo on CPU, RCCL on
the GPU box) and rank 0 compares with the single-process render of all voices.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, V, T, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import srack_pkg
    from oracle import oracle as O
    W = srack_pkg.load_workloads()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = O.OraclePatch(48000, 1024, 2)
    ids = W.build_p1(g, adsr="finite", lfo_val=-2.0)
    det, cut = W.p1_voice_params(V, first_voice=rank * V)  # same call bench.py makes
    frames, mix = g.render_batch(V, T, [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)], mix=True, threads=1)
    # a noise voice on top: its streams are keyed by the GLOBAL voice index (set_noise_seed(seed, first_voice = rank * V))
    nz = O.OraclePatch(48000, 1024, 2)
    n, o = nz.add_module(O.MOD_NOISE), nz.add_module(O.MOD_OUTPUT)
    nz.connect(n, 0, o, 0)
    nz.connect(n, 0, o, 1)
    nz.set_noise_seed(2024, rank * V)
    mix = mix + nz.render_batch(V, T, mix=True)[1]
    part = torch.from_numpy(mix.astype(np.float32))
    dist.reduce(part, dst=0, op=dist.ReduceOp.SUM)
    gathered = [torch.zeros(V, dtype=torch.float32) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(det.copy()), gathered, dst=0)
    if rank == 0:
        q.put((part.numpy(), np.concatenate([t.numpy() for t in gathered])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_process(W, oracle):
    import torch.multiprocessing as mp
    world, V, T = 2, 24, 1500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    reduced, det_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all voices
    det, cut = W.p1_voice_params(world * V)
    np.testing.assert_array_equal(det_all, det)  # shards draw exactly the single-process parameters
    g = oracle.OraclePatch(48000, 1024, 2)
    ids = W.build_p1(g, adsr="finite", lfo_val=-2.0)
    frames, mix = g.render_batch(world * V, T, [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)], mix=True, threads=2)
    nz = oracle.OraclePatch(48000, 1024, 2)
    n, o = nz.add_module(oracle.MOD_NOISE), nz.add_module(oracle.MOD_OUTPUT)
    nz.connect(n, 0, o, 0)
    nz.connect(n, 0, o, 1)
    nz.set_noise_seed(2024, 0)
    nframes, nmix = nz.render_batch(world * V, T, mix=True)
    mix = mix + nmix
    scale = np.abs(frames.astype(np.float64)).sum(axis=2) + np.abs(nframes.astype(np.float64)).sum(axis=2)
    assert (np.abs(reduced - mix) <= 1e-5 * np.maximum(scale, 1.0)).all()
    assert np.abs(mix).max() > 0.5


def test_voice_draw_is_shard_invariant(W):
    full = W.p1_voice_params(1000)
    for v0, v1 in ((0, 250), (250, 777), (777, 1000)):
        part = W.p1_voice_params(v1 - v0, first_voice=v0)
        np.testing.assert_array_equal(part[0], full[0][v0:v1])
        np.testing.assert_array_equal(part[1], full[1][v0:v1])
