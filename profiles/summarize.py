"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 rocpd SQLite output) into small committed files under profiles/.

    python profiles/summarize.py r03 [--no-assert]

Every pass of profiles/run_profile.sh runs the same `bench.py --steps K --warmup W` command, whose one JSON line is in that pass's log.
The dispatch list of a pass therefore ends in W + K identical step blocks; ONLY THE K TIMED STEPS are summarised (the warm-up step
first-touches 50 GB of frames and runs at another clock; round 2 averaged it in and its per-kernel averages did not add up with the
step times they were meant to explain).  Dispatches are grouped by (kernel name, grid size): a specialised kernel's control-only
launches share the voice launches' name but not their grid, and the HIP events of `srack_render_kernel_ms` bracket voice launches only.

Writes
  profiles/<tag>_kernel_stats.csv   per (kernel, grid): launches per step, median / mean / min / max duration over the timed steps, sum per step
  profiles/<tag>_summary.json       the same + PMC counters per launch (timed steps only) + derived HBM bytes + the reconciliation block:
        trace.sum_dominant_ms_per_step  <=  trace.gpu_span_ms_per_step  ~  traced bench line's ms_per_step        (asserted)
        trace.dominant_median_ms        ~  traced bench line's roofline.kernel_ms (HIP events, same run)           (asserted to 2 %)
        frac_from_profile = algorithmic bytes per launch / dominant mean duration / 8 TB/s   vs  the line's frac_kernel
    and, once profiles/<tag>_bench.json (the UN-profiled line of the same box) exists, the profiler's inflation of the step.
WRITE_SIZE / FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) — the raw
values are kept and the corrected figure is given separately.  A profiled run clocks lower than an un-profiled one (same guide, "DVFS
give-back": 1.89-1.95 vs 2.02 GHz): durations are only comparable within one run, which is why the traced run's own line is the yardstick.
"""
import csv
import glob
import json
import os
import sqlite3
import statistics
import sys

HBM_PEAK = 8.0e12


def find_db(src, sub):
    hits = glob.glob(os.path.join(src, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


def bench_line(path):
    """the ONE JSON line bench.py printed into a pass's log"""
    try:
        for ln in open(path, errors="replace"):
            ln = ln.strip()
            if ln.startswith('{"metric"'):
                return json.loads(ln)
    except OSError:
        pass
    return None


def dispatches(con):
    """[(name, grid class, start_ns, end_ns, dispatch_id)] in start order.  The grid class of a launch is the largest grid its kernel is
    ever launched with ("voice launches": the co-scheduled control blocks add one or a few workgroups to some of them, which is not
    another kind of launch) or, for a launch of less than half that, its own grid (a specialised kernel's control-only launches)."""
    q = ("select S.display_name, K.grid_size_x * K.grid_size_y * K.grid_size_z, K.start, K.end, K.dispatch_id from rocpd_kernel_dispatch K "
         "join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid order by K.start, K.dispatch_id")
    rows = [(str(n), int(g), int(s), int(e), int(d)) for n, g, s, e, d in con.execute(q)]
    biggest = {}
    for n, g, _, _, _ in rows:
        biggest[n] = max(biggest.get(n, 0), g)
    return [(n, biggest[n] if 2 * g > biggest[n] else g, s, e, d) for n, g, s, e, d in rows]


def step_period(keys, n_steps):
    """The smallest P such that the last n_steps * P entries of `keys` are n_steps repetitions of one block; 0 if there is none."""
    n = len(keys)
    for p in range(1, n // n_steps + 1):
        tail = keys[n - n_steps * p:]
        block = tail[:p]
        if all(tail[j * p:(j + 1) * p] == block for j in range(1, n_steps)):
            return p
    return 0


def timed_blocks(rows, warmup, steps):
    """-> (period, [block of rows per timed step], rows of the last warm-up block or [])"""
    keys = [(r[0], r[1]) for r in rows]
    p = step_period(keys, warmup + steps)
    n = len(rows)
    if p:
        blocks = [rows[n - (steps - j) * p: n - (steps - j - 1) * p] for j in range(steps)]
        prev = rows[n - (steps + 1) * p: n - steps * p] if warmup > 0 else []
        return p, blocks, prev
    # Dispatches of two streams may interleave differently from step to step: no exact period.  Per (kernel, grid) the count per step is
    # still fixed: a key launched c x (W + K) times contributes its dispatches [(W + j) c, (W + j + 1) c) to timed step j.
    by_key = {}
    for r in rows:
        by_key.setdefault((r[0], r[1]), []).append(r)
    blocks, prev, per_step = [[] for _ in range(steps)], [], 0
    for k, v in by_key.items():
        if len(v) % (warmup + steps) or len(v) < warmup + steps:
            continue  # set-up work
        c = len(v) // (warmup + steps)
        per_step += c
        for j in range(steps):
            blocks[j] += v[(warmup + j) * c:(warmup + j + 1) * c]
        if warmup > 0:
            prev += v[(warmup - 1) * c: warmup * c]
    if per_step == 0:
        return 0, [], []
    return per_step, blocks, prev


def short(name):
    return name if len(name) <= 120 else name[:117] + "..."


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r03"
    do_assert = "--no-assert" not in sys.argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(root, "profiles")
    out = {"tag": tag, "method": "timed steps only (the last K step blocks of the dispatch list); groups = (kernel name, grid size)",
           "trace": {}, "kernels": {}, "counters_per_launch": {}, "derived": {}, "checks": []}
    problems = []

    # ---- pass 1: kernel trace ------------------------------------------------------------------------------------------------------------
    line = bench_line(os.path.join(src, "trace.log"))
    con = find_db(src, "trace")
    dominant = None
    if con and line:
        json.dump(line, open(os.path.join(dst, f"{tag}_bench_traced.json"), "w"))
        W, K = int(line["warmup"]), int(line["steps"])
        rows = dispatches(con)
        period, blocks, prev = timed_blocks(rows, W, K)
        tr = out["trace"]
        tr.update({"steps": K, "warmup": W, "dispatches_total": len(rows), "dispatches_per_step": period})
        if period:
            per_key = {}
            for b in blocks:
                for name, grid, s, e, _ in b:
                    per_key.setdefault((name, grid), []).append((e - s) / 1e6)
            sums = {k: sum(v) / K for k, v in per_key.items()}
            dominant = max(sums, key=sums.get)
            for (name, grid), v in sorted(per_key.items(), key=lambda kv: -sums[kv[0]]):
                out["kernels"][f"{short(name)} [grid {grid}]"] = {
                    "launches_per_step": len(v) / K, "median_ms": statistics.median(v), "mean_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v),
                    "sum_ms_per_step": sums[(name, grid)]}
            # GPU-side span of a timed step: from the end of the previous step's last dispatch (step 1: its own first start) to the end of this one's
            ends = [max(r[3] for r in b) for b in blocks]
            first = min(r[2] for r in blocks[0])  # (the fence between warm-up and timed steps leaves the GPU idle: not part of step 1)
            spans = [(ends[0] - first) / 1e6] + [(ends[j] - ends[j - 1]) / 1e6 for j in range(1, K)]
            dom = per_key[dominant]
            tr.update({
                "dominant": f"{short(dominant[0])} [grid {dominant[1]}]", "dominant_launches_per_step": len(dom) / K,
                "dominant_median_ms": statistics.median(dom), "dominant_mean_ms": sum(dom) / len(dom),
                "sum_dominant_ms_per_step": sums[dominant], "sum_all_kernels_ms_per_step": sum(sums.values()),
                "gpu_span_ms_per_step": sum(spans) / K, "gpu_span_ms_per_step_min": min(spans), "gpu_span_ms_per_step_max": max(spans),
                "bench_traced_ms_per_step": line["ms_per_step"], "bench_traced_kernel_ms": line["roofline"]["kernel_ms"],
                "bench_traced_launches_per_step": line["roofline"]["launches_per_step"],
                "bench_traced_frac_kernel": line["roofline"].get("hbm", line["roofline"]).get("frac_kernel"),
            })
            bpl = line["roofline"].get("algorithmic_bytes_per_launch")
            if bpl:
                tr["frac_from_profile"] = bpl / (tr["dominant_mean_ms"] * 1e-3) / HBM_PEAK
                tr["frac_from_profile_step"] = line["roofline"]["algorithmic_bytes_per_step"] / (tr["gpu_span_ms_per_step"] * 1e-3) / HBM_PEAK

            def check(what, ok, detail):
                out["checks"].append({"check": what, "ok": bool(ok), "detail": detail})
                if not ok:
                    problems.append(f"{what}: {detail}")

            check("sum of the dominant kernel per step <= GPU span of a step", tr["sum_dominant_ms_per_step"] <= tr["gpu_span_ms_per_step"] * 1.001,
                  f'{tr["sum_dominant_ms_per_step"]:.3f} ms vs {tr["gpu_span_ms_per_step"]:.3f} ms')
            check("GPU span of a step <= the traced run's own ms_per_step (host clock, incl. the final sync)",
                  tr["gpu_span_ms_per_step"] <= line["ms_per_step"] * 1.005, f'{tr["gpu_span_ms_per_step"]:.3f} ms vs {line["ms_per_step"]:.3f} ms')
            if line["roofline"]["kernel_ms"] > 0 and abs(tr["dominant_launches_per_step"] - line["roofline"]["launches_per_step"]) < 0.5:
                rel = tr["dominant_mean_ms"] / line["roofline"]["kernel_ms"] - 1.0
                tr["trace_vs_hip_events"] = rel
                # (an event pair brackets the dispatch, launch latency included: a few microseconds, which is more than 2 % of a 0.2 ms kernel)
                gap_us = (line["roofline"]["kernel_ms"] - tr["dominant_mean_ms"]) * 1e3
                check("rocprof mean duration of the dominant kernel vs HIP events of the same run, within 2 % (or 8 us of launch latency)",
                      abs(rel) <= 0.02 or 0.0 <= gap_us <= 8.0, f'{tr["dominant_mean_ms"]:.4f} ms vs {line["roofline"]["kernel_ms"]:.4f} ms ({rel:+.2%}, {gap_us:+.1f} us)')
            with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Kernel [grid]", "LaunchesPerStep", "MedianMs", "MeanMs", "MinMs", "MaxMs", "SumMsPerStep"])
                for name, v in out["kernels"].items():
                    w.writerow([name, v["launches_per_step"], f'{v["median_ms"]:.6f}', f'{v["mean_ms"]:.6f}', f'{v["min_ms"]:.6f}', f'{v["max_ms"]:.6f}', f'{v["sum_ms_per_step"]:.6f}'])
                w.writerow([f"# timed steps only: {K} steps after {W} warm-up; GPU span per step {tr['gpu_span_ms_per_step']:.4f} ms; "
                            f"traced bench line {line['ms_per_step']:.4f} ms/step, kernel_ms (HIP events) {line['roofline']['kernel_ms']:.4f}"])
        else:
            problems.append("no periodic step structure found in the dispatch list")
    elif con is None:
        problems.append("no trace database")
    else:
        problems.append("no bench line in trace.log")

    # ---- PMC passes: counters per launch over the timed steps ------------------------------------------------------------------------------
    for sub in ("pmc_sq", "pmc_wr", "pmc_rd", "pmc_mem"):
        con = find_db(src, sub)
        pl = bench_line(os.path.join(src, sub + ".log"))
        if not con or not pl:
            continue
        rows = dispatches(con)
        _, blocks, _ = timed_blocks(rows, int(pl["warmup"]), int(pl["steps"]))
        keep = {r[4]: (r[0], r[1]) for b in blocks for r in b} if blocks else {r[4]: (r[0], r[1]) for r in rows}
        acc = {}
        for did, counter, value in con.execute("select dispatch_id, counter_name, value from counters_collection"):
            k = keep.get(int(did))
            if k is None:
                continue
            acc.setdefault(k, {}).setdefault(str(counter), []).append(float(value))
        for (name, grid), cs in acc.items():
            key = f"{short(name)} [grid {grid}]"
            for counter, vals in cs.items():
                out["counters_per_launch"].setdefault(key, {})[counter] = {"launches": len(vals), "avg": sum(vals) / len(vals)}

    for name, c in out["counters_per_launch"].items():
        d = {}
        if "WRITE_SIZE" in c:
            d["hbm_write_bytes"] = c["WRITE_SIZE"]["avg"] * 1024
        if "FETCH_SIZE" in c:
            d["hbm_read_bytes_raw"] = c["FETCH_SIZE"]["avg"] * 1024
            d["hbm_read_bytes_gfx950_corrected"] = c["FETCH_SIZE"]["avg"] * 1024 * 2
        if "hbm_write_bytes" in d and "hbm_read_bytes_gfx950_corrected" in d:
            d["hbm_traffic_bytes"] = d["hbm_write_bytes"] + d["hbm_read_bytes_gfx950_corrected"]
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]["avg"] > 0:
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"]["avg"] / c["SQ_WAVES"]["avg"]
            if "SQ_INSTS_SALU" in c:
                d["salu_insts_per_wave"] = c["SQ_INSTS_SALU"]["avg"] / c["SQ_WAVES"]["avg"]
        if "GRBM_GUI_ACTIVE" in c and name in out["kernels"]:
            d["note_clock"] = "GRBM_GUI_ACTIVE / duration of the same pass would be the clock; durations here are from the trace pass"
        if d:
            out["derived"][name] = d
    if dominant and line:
        key = f"{short(dominant[0])} [grid {dominant[1]}]"
        tb = out["derived"].get(key, {}).get("hbm_traffic_bytes")
        bpl = line["roofline"].get("algorithmic_bytes_per_launch")
        if tb and bpl:
            out["trace"]["traffic_over_algorithmic"] = tb / bpl

    # ---- the un-profiled line of the same box, when run_profile.sh has written it --------------------------------------------------------------
    try:
        plain = json.load(open(os.path.join(dst, f"{tag}_bench.json")))
        if line and plain.get("config", {}).get("name") == line.get("config", {}).get("name"):
            out["trace"]["bench_unprofiled_ms_per_step"] = plain["ms_per_step"]
            out["trace"]["bench_unprofiled_kernel_ms"] = plain["roofline"]["kernel_ms"]
            out["trace"]["profiler_inflation_of_the_step"] = line["ms_per_step"] / plain["ms_per_step"] - 1.0
    except (OSError, ValueError, KeyError):
        pass

    json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps(out["trace"], indent=1))
    print(json.dumps(out["kernels"], indent=1))
    print(json.dumps(out["derived"], indent=1))
    for c in out["checks"]:
        print(("ok   " if c["ok"] else "FAIL ") + c["check"] + ": " + c["detail"])
    if problems and do_assert:
        print("summarize: " + "; ".join(problems), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
