"""srack_device_to_host: GB/s of reading rendered frames back into fresh pageable memory (numpy), and that the bytes arrive.
usage: readback_bench.py [GB ...]    (default: 0.05 1 8; the headline's frames for one second of audio are 50.3 GB)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg

S = srack_pkg.load()
sizes = [float(x) for x in sys.argv[1:]] or [0.05, 1.0, 8.0]
for gb in sizes:
    n = int(gb * 1e9) // 4 // 4096 * 4096
    avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    if 4 * n * 1.2 > avail:
        print(f"{gb} GB: skipped (MemAvailable {avail / 1e9:.0f} GB)")
        continue
    d = C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d), 4 * n) == 0, S.lib.srack_last_error()
    # a pattern that depends on the position: render something cheap into it — P1 frames of as many voices as fit
    V = 4096
    T = n // V
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=-2.0)
    p.configure_voices(V)
    det, cut = S.p1_voice_params(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    done = 0
    while done < T:  # (a render call is at most 2^31 frames bytes; 65536-sample calls)
        k = min(65536, T - done)
        p.render_raw(k, d.value + 4 * done * V, None, 0, None)
        done += k
    assert S.lib.srack_device_sync(None) == 0
    for fresh in (True, False):
        host = np.empty(T * V, np.float32) if fresh else host
        t0 = time.perf_counter()
        assert S.lib.srack_device_to_host(host.ctypes.data_as(C.c_void_p), d, 4 * T * V, None) == 0, S.lib.srack_last_error()
        dt = time.perf_counter() - t0
        print(f"{4 * T * V / 1e9:.2f} GB into {'fresh' if fresh else 'touched'} pageable memory: {dt:.3f} s = {4 * T * V / dt / 1e9:.1f} GB/s", flush=True)
    # the bytes: a second read-back of a sample of rows through small calls must agree with the big one
    h2 = host.reshape(T, V)
    bad = 0
    for t in list(range(0, T, max(1, T // 64)))[:64] + [T - 1]:
        row = np.zeros(V, np.float32)
        assert S.lib.srack_device_to_host(row.ctypes.data_as(C.c_void_p), d.value + 4 * t * V, 4 * V, None) == 0
        bad += not np.array_equal(row.view(np.uint32), h2[t].view(np.uint32))
    print(f"   rows re-read one by one: {bad} differ; max |frame| {float(np.abs(h2[:: max(1, T // 512)]).max()):.3f}")
    assert bad == 0
    del host, h2
    assert S.lib.srack_device_free(d) == 0
