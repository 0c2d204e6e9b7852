"""The N > 1 path without GPUs: world_size-2 gloo processes check the sharding contract bench.py relies on.

What is multi-GPU about this path: voices shard by GLOBAL voice index with no exchange during the
render; the only collective is the sum of the per-rank partial mixes.  Here each rank renders its shard
with the CPU oracle (test infrastructure standing in for the GPU), draws its per-voice parameters exactly
like bench.py does, reduces the [2][T] partial mix to rank 0 over torch.distributed (gloo on CPU, RCCL on
the GPU box) and rank 0 compares with the single-process render of all voices.
"""
import json
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


# ---- bench.py's own N > 1 machinery (launcher, rendezvous, control plane, timing protocol, one JSON line), no GPU -----------
def _run_bench_cpu(tmp_path, gpus, extra=(), env_extra=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SRACK_TEST_MIX_OUT"] = str(tmp_path / "mix.npy")
    env["SRACK_TEST_STEPS_OUT"] = str(tmp_path / "steps")
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_rank.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--voices", "12", "--samples", "1200", "--no-cpu", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0 only
    assert len(lines[0]) < 8000, len(lines[0])  # the driver's record keeps the last 8 KB of stdout: the line has to fit whole
    return json.loads(lines[0])


def test_bench_line_fits_the_drivers_record():
    """Round 5's default line was 21.9 KB and the driver's record (last 8 KB of stdout) lost the headline.  bench.shape_line
    is what stands between everything bench.py measures and the ONE line: fed round 5's own full output (profiles/r05_default_line.json:
    ten side configurations, the long `arithmetic` paragraphs), and the same with every string doubled, it must stay under 6 KB and keep the
    contract's keys, `roofline` with its fraction and the side configurations' scalars, and `cpu_baseline`."""
    import copy
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_default_line.json")))
    assert len(json.dumps(full)) > 20000

    def double(x):
        if isinstance(x, dict):
            return {k: double(v) for k, v in x.items()}
        return x + " " + x if isinstance(x, str) and len(x) > 40 else x
    for src in (full, double(full)):
        line, detail = bench.shape_line(copy.deepcopy(src))
        text = json.dumps(line)
        assert len(text) <= bench.LINE_BUDGET < 8000, len(text)
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert back[k] == src[k], k
        assert back["config"]["name"] == "cfg3" and "262144 voices" in back["config"]["workload"]
        rf = back["roofline"]
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - src["roofline"]["frac"]) < 1e-5 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-5
        assert rf["traffic"] == float("%.6g" % src["roofline"]["traffic"]) and rf["kernel_ms"] > 0 and rf["frac_kernel"] > 0
        for w in ("cfg3_exact", "cfg3_poly", "cfg2", "cfg4", "cfg4_fast", "cfg4_b1024", "cfg4_b1024_fast", "p3", "p4"):
            for key in bench.SIDE_KEYS:
                assert abs(rf[f"{w}_{key}"] - src["roofline"][f"{w}_{key}"]) <= 1e-5 * abs(src["roofline"][f"{w}_{key}"]), (w, key)
        cb = back["cpu_baseline"]
        assert cb["value"] == src["cpu_baseline"]["value"] and cb["cores"] == 256 and cb["kind"] == "port" and cb["sample"]
        assert "configs" not in back and "configs" in detail  # the side configurations' full lines are in the detail, not on the line
    # a line that is already short goes through untouched (but for the pointer to the detail)
    small = {"metric": "m", "value": 1.0, "n_gpus": 1, "per_rank_ms_per_step": [1.0], "config": {"workload": "w", "arithmetic": "a"}, "roofline": {"frac": 0.5}}
    line, _ = bench.shape_line(copy.deepcopy(small))
    assert line == dict(small, detail="bench_detail.json")


def test_bench_self_launches_two_ranks(tmp_path, W, oracle):
    """`bench.py --gpus 2` with no launcher around it: bench.py starts the ranks itself, rank 0 prints the one line, the value is
    the whole job's, and the reduced mix equals a single-process render of all the voices."""
    out = _run_bench_cpu(tmp_path, 2)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["unit"] == "voice-samples/s"
    assert abs(out["value"] - 2 * 12 * 1200 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    for r in range(2):  # every rank ran warmup + steps, no more
        assert open(str(tmp_path / "steps") + f".{r}").read() == "3"
    reduced = np.load(tmp_path / "mix.npy")
    det, cut = W.p1_voice_params(24)
    g = oracle.OraclePatch(48000, 1024, 2)
    ids = W.build_p1(g, adsr="finite", lfo_val=-2.0)
    frames, mix = g.render_batch(24, 1200, [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)], mix=True, threads=2)
    scale = np.abs(frames.astype(np.float64)).sum(axis=2)
    assert (np.abs(reduced - mix) <= 1e-5 * np.maximum(scale, 1.0)).all()
    assert np.abs(mix).max() > 0.5


def test_bench_under_an_external_launcher(tmp_path):
    """The driver's form: a launcher exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; bench.py must not start ranks of its own."""
    import bench
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_rank.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--voices", "4", "--samples", "600", "--no-cpu"]
    env = dict(os.environ, SRACK_TEST_STEPS_OUT=str(tmp_path / "steps"))
    rc = bench.launch_ranks(2, cmd, env=env, timeout=240)  # stands in for torch.distributed.run: same environment contract
    assert rc == 0
    assert sorted(os.listdir(tmp_path)) == ["steps.0", "steps.1"]  # two ranks, not four


def test_bench_under_torch_distributed_run(tmp_path):
    """The driver's command, literally: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` — here with the CPU backend standing in and N = 2."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SRACK_TEST_STEPS_OUT=str(tmp_path / "steps"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "bench_cpu_rank.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--voices", "4", "--samples", "600", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and len(out["per_rank_ms_per_step"]) == 2
    assert abs(out["value"] - 2 * 4 * 600 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert sorted(os.listdir(tmp_path)) == ["steps.0", "steps.1"] and open(str(tmp_path / "steps") + ".1").read() == "3"


def test_bench_single_rank_needs_no_rendezvous(tmp_path):
    out = _run_bench_cpu(tmp_path, 1)
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1


def test_launcher_reports_a_failing_rank():
    import bench
    rc = bench.launch_ranks(2, [sys.executable, "-c", "import os, sys, time; time.sleep(0.3 if os.environ['RANK'] == '0' else 30); sys.exit(7 if os.environ['RANK'] == '0' else 0)"], timeout=60)
    assert rc == 7  # and the sleeping rank was stopped rather than waited for


@pytest.mark.gpu
def test_bench_with_more_ranks_than_gpus_fails_cleanly():
    """`bench.py --gpus N` on a box with fewer devices: the rank without a device exits with an error, the launcher stops the others
    (they would wait in the rendezvous for ever) and reports the failure — no hang, no JSON line."""
    import subprocess
    import srack_pkg
    S = srack_pkg.load()
    n = S.device_count() + 1
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--voices", "4096",
                        "--samples", "2048", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
