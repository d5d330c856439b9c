#!/usr/bin/env python3
"""CPU baseline runner -- TEST INFRASTRUCTURE (oracle/), executed only by bench.py's `cpu_baseline` leg (as a subprocess, so
that the worker processes are forked from an interpreter that never loaded the HIP runtime) and by hand.

Times the CPU oracle (a port of the reference algorithm, `kind: "port"`) on the host cores the way the reference uses a CPU:
its custom ops are single-threaded, and the one multi-threaded piece, the nanoflann k-NN, runs ONE CLOUD PER THREAD
(`#pragma omp parallel for` over the batch, libs/nearest_neighbors/knn_.cxx:108).  So:

  end to end   oracle/generator.py on synthetic 256-point patches: (a) one process, one thread; (b) one patch per process over
               W = min(cores, patches) single-threaded processes -- the batch-parallel figure.  (Round 2 threaded the short
               inner loops of every layer instead and was SLOWER with 256 hardware threads than with one.)
  per op       oracle/dispu_oracle.c at bounded shapes with 1 thread and with min(cores, clouds) threads, one cloud each
               (the OpenMP regions are capped at their cloud count: oracle/dispu_oracle.c:orc_nt).

Prints ONE JSON object on the last line of stdout.    python oracle/cpu_bench.py [--seconds S] [--no-ops]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

# one thread per process for everything numpy links (OpenBLAS / MKL pools would otherwise start `cores` threads in each of the
# `cores` worker processes); the oracle's own OpenMP regions take their thread count from orc_set_threads()
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "OMP_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NPOINT, UP = 256, 4
_P = None
_X = None


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, avail


def _one_patch(i):
    from oracle import generator as OG
    c, f = OG.generator_forward(_P, _X[i:i + 1])
    return float(f.sum())


def end_to_end(seconds):
    global _P, _X
    import numpy as np  # noqa: F401
    from dispu_amd import synth                      # numpy-only helper (no torch / HIP import)
    from dispu_amd.params import init_params
    from oracle import generator as OG
    from oracle import oracle as O
    model, avail = cpu_info()
    _P = init_params(seed=1234)
    O.set_threads(1)                                  # inherited by the forked workers: every process is single-threaded
    workers = max(1, min(avail, 256))
    _X = synth.patches(max(2 * workers, 4), NPOINT, seed=1000)
    OG.generator_forward(_P, _X[:1])                  # page in
    t = time.perf_counter()
    n1 = 0
    while n1 < 2 or (time.perf_counter() - t < min(3.0, seconds / 4) and n1 < 8):
        _one_patch(n1)
        n1 += 1
    t1 = (time.perf_counter() - t) / n1
    pts = NPOINT * UP
    out = {"points_per_s_1_thread": pts / t1, "s_per_patch_1_thread": t1}
    if workers > 1:
        ctx = mp.get_context("fork")
        with ctx.Pool(workers) as pool:
            pool.map(_one_patch, range(workers), chunksize=1)                       # every worker warm
            # rounds of one patch per process until the time budget is used (a round can be much slower than t1: 256 processes
            # share the memory system)
            t = time.perf_counter()
            done = 0
            while True:
                pool.map(_one_patch, [(done + i) % _X.shape[0] for i in range(workers)], chunksize=1)
                done += workers
                dt = time.perf_counter() - t
                if dt > seconds * 0.6 or done >= 8 * workers:
                    break
        out.update({"points_per_s_all_cores": done * pts / dt, "processes": workers, "patches_timed": done, "seconds_timed": dt})
    else:
        out.update({"points_per_s_all_cores": pts / t1, "processes": 1, "patches_timed": n1, "seconds_timed": t1 * n1})
    out.update({"cpu_model": model, "host_cores": avail})
    return out


def per_op(budget_s):
    import numpy as np
    from oracle import oracle as O
    _, avail = cpu_info()
    rng = np.random.default_rng(7)
    R = lambda *s: rng.random(s, dtype=np.float32)
    x1k, y1k = R(32, 1024, 3), R(32, 1024, 3)
    idx = rng.integers(0, 1024, (32, 1024, 16)).astype(np.int32)
    feats = rng.standard_normal((32, 1024, 128)).astype(np.float32)
    xs, ys = R(4, 1024, 3), R(4, 1024, 3)
    cases = [
        ("farthest_point_sample", (32, 1024, 384), 32, lambda: O.farthest_point_sample(384, x1k)),
        ("knn_xyz (self query)", (32, 1024, 16), 32, lambda: O.knn_batch(x1k, x1k, 16)),
        ("query_ball_point", (32, 1024, 1024, 20), 32, lambda: O.query_ball_point(0.07, 20, x1k, x1k)),
        ("group_point", (32, 1024, 1024, 16, 128), 32, lambda: O.group_point(feats, idx)),
        ("three_nn", (32, 1024, 256), 32, lambda: O.three_nn(x1k, y1k[:, :256])),
        ("nn_distance (both directions)", (32, 1024, 1024), 32, lambda: O.nn_distance(x1k, y1k)),
        ("approx_match", (4, 1024, 1024), 4, lambda: O.approx_match(xs, ys)),
    ]
    rows, t_start = [], time.perf_counter()
    for name, shape, clouds, fn in cases:
        threads = max(1, min(avail, clouds))
        r = {"op": name, "shape": list(shape), "threads_all_cores": threads}
        for label, c in (("ms_1_thread", 1), ("ms_all_cores", threads)):
            if time.perf_counter() - t_start > budget_s:
                r[label] = None
                continue
            O.set_threads(c)
            t = time.perf_counter()
            fn()
            warm = time.perf_counter() - t
            best = None
            for _ in range(3 if warm < 0.3 else 1):
                t = time.perf_counter()
                fn()
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            r[label] = round(best * 1e3, 3)
        rows.append(r)
    O.set_threads(1)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=14.0, help="target CPU wall time of the end-to-end sample")
    ap.add_argument("--no-ops", action="store_true")
    args = ap.parse_args()
    # no OMP_PROC_BIND / OMP_PLACES: libgomp would pin the master thread to one place and every forked worker inherits that mask
    out = end_to_end(args.seconds)
    if not args.no_ops:
        out["ops"] = per_op(budget_s=max(10.0, args.seconds))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
