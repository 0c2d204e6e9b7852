"""Is it the COPY?  Round 5's soaks, sixteen processes on one device: one render in ~2 500 came back with a stretch of ZEROS — the same sample
range in different patches of one worker, sometimes in two renders in a row — and never when the device had one process.  This takes the
render out of the picture: one patch is rendered ONCE into device buffers, then the same bytes are copied back N times through the path the
tests use (srack_device_to_host: hipMemcpyAsync into pageable memory + a stream sync) into fresh zero-filled arrays, and every copy is compared
with the first.  usage: copy_probe.py <copies> [workers]   (workers > 1: that many processes side by side, each its own patch and buffers)"""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def worker(n_copies, wid):
    import srack_pkg
    S = srack_pkg.load()
    V, T = 16, 48000
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p3(p)
    p.configure_voices(V)
    n_planes, _ = p.planes()
    nbytes = n_planes * T * V * 4
    d = C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d), nbytes) == 0
    p.render_raw(T, d, None, 0, None)
    assert S.lib.srack_device_sync(None) == 0
    first = np.zeros((n_planes, T, V), np.float32)
    assert S.lib.srack_device_to_host(first.ctypes.data_as(C.c_void_p), d, nbytes, None) == 0
    bad = 0
    t0 = time.time()
    for k in range(n_copies):
        buf = np.zeros((n_planes, T, V), np.float32)
        assert S.lib.srack_device_to_host(buf.ctypes.data_as(C.c_void_p), d, nbytes, None) == 0
        if not np.array_equal(buf, first):
            diff = np.argwhere(buf != first)
            zeros = bool((buf[buf != first] == 0).all())
            bad += 1
            print(f"worker {wid} copy {k}: {len(diff)} words differ, planes {sorted(set(diff[:, 0]))}, t {diff[:, 1].min()}..{diff[:, 1].max()}, "
                  f"the copy holds zeros there: {zeros}; byte offset of the first {((diff[0][0] * T + diff[0][1]) * V + diff[0][2]) * 4}", flush=True)
            if not np.array_equal(first, first):  # (never)
                pass
    print(f"worker {wid}: {n_copies} copies of {nbytes} bytes, {bad} differ from the first, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 3:
        worker(int(sys.argv[1]), int(sys.argv[3]))
    else:
        n, w = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n), "x", str(i)]) for i in range(w)]
        # (the soaks' other half: oracle threads keeping the host cores busy)
        for pr in procs:
            pr.wait()
