"""tests/test_gpu_tick.py::test_two_patches_on_two_streams, in a process of its own (torch must start HIP before the library loads:
INTEGRATION.md section 3).  Two patches are rendered block by block on two non-default streams, their calls interleaved and nothing
synchronised until the end — each patch runs its own tick session on its own stream — then one of them changes stream mid-way.  The
frames and mixes must equal, bit for bit, what the same patches render alone on the null stream.

    python tests/stream_driver.py <out.npz>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

torch.cuda.init()
import srack_pkg  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import tick_driver  # noqa: E402

L, N = 512, 9
SPEC = [("p1", 0), ("p3", 32)]


def build(S, scenario):
    V = tick_driver.voices_of(scenario)
    p = S.Patch(48000, 1024, 2)
    _, over = tick_driver.make(S, p, scenario, V)
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    return p, V


def main():
    S = srack_pkg.load()
    dev = torch.device("cuda", 0)
    out = {}
    # alone, on the null stream
    for k, (scenario, flags) in enumerate(SPEC):
        p, V = build(S, scenario)
        n_planes, _ = p.planes()
        fr = torch.empty((N, n_planes, L, V), dtype=torch.float32, device=dev)
        mx = torch.empty((N, 2, L), dtype=torch.float32, device=dev)
        for i in range(N):
            p.render_raw(L, fr[i].data_ptr(), mx[i].data_ptr(), flags, None)
        torch.cuda.synchronize()
        out[f"alone_fr{k}"], out[f"alone_mx{k}"] = fr.cpu().numpy(), mx.cpu().numpy()
    # together: interleaved calls on two streams, no synchronisation in between; patch 0 moves to a third stream after five calls
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    patches, bufs = [], []
    for scenario, flags in SPEC:
        p, V = build(S, scenario)
        n_planes, _ = p.planes()
        patches.append(p)
        bufs.append((torch.empty((N, n_planes, L, V), dtype=torch.float32, device=dev), torch.empty((N, 2, L), dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()
    for i in range(N):
        for k, (scenario, flags) in enumerate(SPEC):
            st = streams[2] if (k == 0 and i >= 5) else streams[k]
            patches[k].render_raw(L, bufs[k][0][i].data_ptr(), bufs[k][1][i].data_ptr(), flags, st.cuda_stream)
    torch.cuda.synchronize()
    for k in range(len(SPEC)):
        out[f"both_fr{k}"], out[f"both_mx{k}"] = bufs[k][0].cpu().numpy(), bufs[k][1].cpu().numpy()
        out[f"info{k}"] = np.array(patches[k].info())
    # a stream that goes away: patch "p1" ticks on a raw non-blocking HIP stream, the host DESTROYS that stream, then (a) reads state back and
    # carries on on the null stream, (b) carries on on a second raw stream straight away.  The library may not touch a stream after the call
    # that passed it (the session's copy-back waits on an event of its own): both continue bit for bit like the patch alone.
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    for tag, read_back in (("gone_read", True), ("gone_move", False)):
        p, V = build(S, "p1")
        n_planes, _ = p.planes()
        fr = torch.empty((N, n_planes, L, V), dtype=torch.float32, device=dev)
        mx = torch.empty((N, 2, L), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        a, b = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(a), 1) == 0 and hip.hipStreamCreateWithFlags(C.byref(b), 1) == 0  # hipStreamNonBlocking
        for i in range(4):
            p.render_raw(L, fr[i].data_ptr(), mx[i].data_ptr(), 0, a.value)
        assert hip.hipStreamDestroy(a) == 0   # (work in flight completes; the handle is dead)
        if read_back:
            out[tag + "_pos"] = p.get_voice_field(1, S.OSC_POS)   # the gate LFO: a module of the control program
            for i in range(4, N):
                p.render_raw(L, fr[i].data_ptr(), mx[i].data_ptr(), 0, None)
        else:
            for i in range(4, N):
                p.render_raw(L, fr[i].data_ptr(), mx[i].data_ptr(), 0, b.value)
        torch.cuda.synchronize()
        assert hip.hipStreamDestroy(b) == 0
        out[tag + "_fr"], out[tag + "_mx"] = fr.cpu().numpy(), mx.cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
