"""One rank of bench.py's N > 1 protocol WITHOUT a GPU (test infrastructure, started by tests/test_dist.py).

bench.py's launcher, rendezvous, control plane (gloo), timing protocol and JSON line run unchanged; only the backend is
replaced: the rank renders its voice shard with the CPU oracle (the checker standing in for the GPU) and sums the partial
mixes over torch.distributed/gloo where the product uses its RCCL communicator.  Rank 0 leaves the reduced mix in
$SRACK_TEST_MIX_OUT for the test to compare with a single-process render of all voices.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class OracleBackend:
    name = "oracle-cpu (test)"

    def __init__(self, args, world, rank, local_rank, cp):
        import torch
        import srack_pkg
        from oracle import oracle as O
        self.torch, self.cp, self.args, self.rank, self.world = torch, cp, args, rank, world
        self.S = W = srack_pkg.load_workloads()
        V = args.voices
        self.g = O.OraclePatch(48000, 1024, 2)
        ids = W.build_p1(self.g, adsr="finite", lfo_val=-2.0)
        det, cut = W.p1_voice_params(V, first_voice=rank * V)   # the draw bench.py's HipBackend makes
        self.ov = [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)]
        self.what = f"test: patch P1 on the CPU oracle, {V} voices per rank"
        self.n_planes, self.buffer_size = 1, 1024
        self.comm = "gloo" if world > 1 else None
        self.ranks_seen = world
        self.mix = None
        self.steps = 0

    def step(self):
        _, mix = self.g.render_batch(self.args.voices, self.args.samples, self.ov, frames=False, mix=True, threads=1)
        t = self.torch.from_numpy(mix.astype(np.float32))
        if self.cp.dist is not None:
            self.cp.dist.reduce(t, dst=0, op=self.cp.dist.ReduceOp.SUM)
        self.mix = t.numpy()
        self.steps += 1

    def sync(self):
        pass

    def arm_kernel_timer(self):
        pass

    def kernel_ms(self):
        return 0.0, 0

    def info(self):
        return "oracle"

    def close(self):
        if self.rank == 0 and os.environ.get("SRACK_TEST_MIX_OUT"):
            np.save(os.environ["SRACK_TEST_MIX_OUT"], self.mix)
        if os.environ.get("SRACK_TEST_STEPS_OUT"):
            with open(os.environ["SRACK_TEST_STEPS_OUT"] + f".{self.rank}", "w") as f:
                f.write(str(self.steps))


if __name__ == "__main__":
    sys.exit(bench.main(backend_cls=OracleBackend, self_cmd=[sys.executable, os.path.abspath(__file__)]))
