"""The N > 1 path on real devices (-m gpu): `bench.py --gpus N` through the HIP backend and the product's RCCL communicator,
CHECKED, not only timed.  The cases switch themselves on with the number of visible devices: on a 1-GPU box only the one-rank case
runs (communicator of one, `--force-dist`); with 2 / 4 / 8 devices the 2 / 4 / 8-rank cases run too.

What is checked is SURVEY 8(d)'s list for config 5 (voices sharded by global voice index, RCCL sum of the [2][T] partial mixes):
  * the communicator spans N ranks (`ranks_seen`), every rank rendered on its own device (distinct PCI bus ids, device == local rank);
  * first and last voice of every shard (and a few in between) against the CPU oracle, |d| <= 1e-5 * max(|ref|, 1);
  * every rank's partial mix against the f64 sum of ALL its frames, and rank 0's reduced mix against the f64 sum over all ranks,
    |d| <= 1e-5 * sum |terms|.
bench.py leaves these in $SRACK_BENCH_DUMP from a fresh render of the same shard through the same entry points (HipBackend.dump).
The reference has no counterpart: one audio thread, one instance (src/main.rs:59-63).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import srack_pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V_PER_RANK, T = 8192, 18000  # 128 waves per rank through the flagship kernel; six launches and a ragged tile; P1's gate LFO (1.72 Hz) first
                             # rises at sample 13 964: the render holds silence, an attack and most of a decay


@pytest.fixture(scope="module")
def S():
    S = srack_pkg.load()
    assert S.device_count() > 0, "no GPU visible: the render path has no CPU fallback"
    return S


def run_bench(tmp_path, n, extra=()):
    env = dict(os.environ, SRACK_BENCH_DUMP=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)  # bench.py is its own launcher here
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--voices", str(V_PER_RANK),
           "--samples", str(T), "--no-cpu", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(n)], r.stderr


def check_shards(S, oracle, out, dumps, n):
    assert out["n_gpus"] == n and out["ranks_seen"] == n and len(out["per_rank_ms_per_step"]) == n
    assert out["value"] == pytest.approx(n * V_PER_RANK * T / (out["ms_per_step"] * 1e-3), rel=1e-9)
    assert out["scaling"] == "weak" and out["config"]["voices_per_gpu"] == V_PER_RANK
    # one device per rank, and it is the one the launcher assigned
    assert len({str(d["pci_bus_id"]) for d in dumps}) == n, [str(d["pci_bus_id"]) for d in dumps]
    for k, d in enumerate(dumps):
        assert int(d["rank"]) == k and int(d["world"]) == n and int(d["hip_device"]) == int(d["local_rank"]) == k and int(d["ranks_seen"]) == n
        assert "kernel=render_voice_chain_track" in str(d["info"])  # the kernel the metric is quoted on
    total, total_abs = 0.0, 0.0
    for k, d in enumerate(dumps):
        idx = d["voices"]
        assert idx[0] == 0 and idx[-1] == V_PER_RANK - 1  # first and last voice of the shard
        det, cut = S.p1_voice_params(V_PER_RANK, first_voice=k * V_PER_RANK)  # GLOBAL voice index: shard k is voices [k V, (k + 1) V)
        o = oracle.OraclePatch(48000, 1024, 2)
        ids = S.build_p1(o)
        ref, _ = o.render_batch(len(idx), T, [(ids["osc_a"], S.OSC_VAL, det[idx]), (ids["vcf"], S.VCF_FREQ, cut[idx])], mix=False, threads=4)
        got = d["frames"][0].astype(np.float64)
        assert np.abs(ref[0]).max() > 0.05, "the oracle's render is silent"
        err = np.abs(got - ref[0]) / np.maximum(np.abs(ref[0]), 1.0)
        assert err.max() <= 1e-5, f"rank {k}: voice {idx[np.unravel_index(err.argmax(), err.shape)[1]]} off by {err.max():.3e}"
        # this rank's partial mix = the sum of ALL its voices' frames (both channels carry plane 0 in P1)
        s, sa = d["frames_sum_f64"], d["frames_abs_sum_f64"]
        for c, plane in enumerate(d["planes"]):
            assert plane >= 0
            assert (np.abs(d["partial_mix"][c] - s[plane]) <= 1e-5 * np.maximum(sa[plane], 1.0)).all(), f"rank {k} channel {c}"
        total, total_abs = total + s, total_abs + sa
    red = dumps[0]["reduced_mix"]
    for c, plane in enumerate(dumps[0]["planes"]):
        assert (np.abs(red[c] - total[plane]) <= 1e-5 * np.maximum(total_abs[plane], 1.0)).all(), f"reduced mix, channel {c}"
    if n > 1:  # and it is more than rank 0's own share
        assert np.abs(red[0] - dumps[0]["partial_mix"][0]).max() > 1e-3


@pytest.mark.parametrize("n", [2, 4, 8])
def test_voices_sharded_over_n_gpus_checked(S, oracle, tmp_path, n):
    """config 5's structure at test size: N ranks x 8192 voices, one RCCL reduce of the mix per step."""
    if S.device_count() < n:
        pytest.skip(f"needs {n} GPUs, {S.device_count()} visible")
    out, dumps, err = run_bench(tmp_path, n)
    check_shards(S, oracle, out, dumps, n)
    assert "ms per step by rank" in err  # a straggler would show in the driver's log


def test_one_rank_through_the_communicator(S, oracle, tmp_path):
    """The same checks with a communicator of one (runs on any box): the dump hook, the reduce, the shard-0 draw."""
    out, dumps, _ = run_bench(tmp_path, 1, ["--force-dist"])
    check_shards(S, oracle, out, dumps, 1)
    assert "srack_dist_reduce_mix" in out["config"]["workload"]


def test_shards_are_the_same_voices_whatever_the_rank_count(S):
    """Voice v of rank r at V voices per rank is global voice r V + v: its parameters do not depend on how many ranks there are."""
    a_det, a_cut = S.p1_voice_params(2 * V_PER_RANK, first_voice=0)
    b_det, b_cut = S.p1_voice_params(V_PER_RANK, first_voice=V_PER_RANK)
    assert (a_det[V_PER_RANK:] == b_det).all() and (a_cut[V_PER_RANK:] == b_cut).all()
