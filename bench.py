#!/usr/bin/env python3
"""Headline benchmark: upsampled points/s of the Dis-PU generator forward, 256 -> 1024 (4x), fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic patches that already live in HBM:
BASELINE.json configs[1] (B = 32 patches of 256 points per GPU, generator forward -> 32 x 1024 points),
and for N > 1 configs[2]: the patch batch is sharded 32 per rank (weak scaling, no data-path collective)
followed by ONE RCCL all-gather that reassembles the upsampled clouds on every rank.
Rank 0 prints one JSON line (contract in the task brief) including `roofline` for the dominant kernel
(HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver (before torch loads HIP)

PATCHES_PER_GPU = 32
NPOINT = 256
UP = 4
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def linear_flops(name):
    # "linear<BM,BN,transb>[MxKxN]" -> algorithmic flops of that launch: 2*M*K*N
    dims = name[name.index("[") + 1:name.index("]")].split("x")
    m, k, n = (int(v) for v in dims)
    return 2.0 * m * k * n


def step_macs_per_patch():
    """Multiply-accumulates of ONE 256 -> 1024 generator forward AS EXECUTED here (the reference
    graph as written is 1.808 G MAC per patch; two exact algebraic savings -- duplicate_up's 482-wide
    conv per source point, PointShuffle2's conv0 per source point -- leave 1.39 G).  Shapes: Common/ops.py:1437-1486
    (dense blocks), :1152-1199, :1089-1110, :1012-1087, :302-346."""
    n, m, k, g = NPOINT, NPOINT * UP, 16, 24
    mac = n * 3 * 24                                                     # layer0
    width = 24
    for d in range(1, 5):
        c = 24 if d == 1 else 48
        if d > 1:
            mac += n * width * 48                                        # layer{d}_prep
        mac += n * n * c                                                 # feature k-NN, GEMM form
        mac += n * k * g * (2 * c + (g + c) + (2 * g + c))               # l0, l1, l2 over the [n, 16] edge tensor
        width += 3 * g + c
    assert width == 480
    mac += n * 480 * 256 + m * 2 * 256                                   # duplicate_up conv1 (per source point) + grid part
    mac += m * 256 * 128                                                 # conv2
    mac += m * (128 * 256 + 256 * 64 + 64 * 3)                           # coarse regressor
    mac += m * m * 3                                                     # xyz k-NN distances
    mac += m * 128 * 320                                                 # K|V, Q, conv0 feature part
    mac += 2 * m * m * 64                                                # attention logits + PV
    mac += m * 64 * 256                                                  # conv_back_project
    mac += m * 134 * 256                                                 # skip
    mac += m * 6 * 128                                                   # conv0 xyz part
    mac += m * k * (128 * 128 + 3 * 16 + 16 * 128)                       # conv1, weight_net, feature x weight
    mac += m * 2048 * 256                                                # after_conv
    mac += m * (256 * 256 + 256 * 256 + 256 * 64 + 64 * 3)               # aggregation + fine regressor
    return mac


REFERENCE_FLOPS_PER_PATCH = 3.616e9      # the graph as the reference writes it


def cpu_baseline(params, target_seconds=12.0, with_ops=True):
    """The CPU oracle (oracle/generator.py + oracle/dispu_oracle.c, a port of the reference algorithm; OpenMP over rows /
    clouds) on the host cores: end to end with 1 thread and with every core, and per op."""
    import numpy as np
    from dispu_amd import synth
    from oracle import generator as OG
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ops_bench
    model, avail = ops_bench.cpu_info()
    x = synth.patches(256, NPOINT, seed=1000)
    OG.generator_forward(params, x[:1])                       # warm-up / page-in
    per = {}
    for c in sorted({avail, min(avail, 64), min(avail, 16), 1}, reverse=True):
        O.set_threads(c)
        t = time.perf_counter()
        OG.generator_forward(params, x[:2])
        per[c] = (time.perf_counter() - t) / 2
    # the oracle's OpenMP loops are short, so every core is not always the fastest setting: `value` is the best one,
    # the 1-thread and all-core figures are reported next to it
    cores = min(per, key=per.get)
    O.set_threads(cores)
    n = int(max(2, min(256, target_seconds / max(per[cores], 1e-3))))
    t = time.perf_counter()
    OG.generator_forward(params, x[:n])
    dt = time.perf_counter() - t
    pts = NPOINT * UP
    out = {"value": n * pts / dt, "unit": "points/s", "cores": cores, "kind": "port", "cpu_model": model, "host_cores": avail,
           "points_per_s_1_thread": pts / per[1], "points_per_s_all_cores": pts / per[avail],
           "points_per_s_by_threads": {str(c): round(pts / v, 1) for c, v in sorted(per.items())},
           "sample": "%d patches of %d points (same synthetic workload), oracle/generator.py with %d OpenMP threads "
                     "(fastest of %s threads on a host with %d cores: %s), %.1f s"
                     % (n, NPOINT, cores, sorted(per), avail, model, dt)}
    if with_ops:
        out["ops"] = ops_bench.cpu_ops()["ops"]
    return out


def pmc_traffic(kern):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_traffic.sh -> profiles/pmc_traffic_latest.json): (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950
    correction of MI355X_MICROARCH.md's HBM section.  Counters cannot be read from inside the process, so this is
    the figure of the last profiled build; None when the file or the kernel is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_latest.json")
    try:
        with open(path) as f:
            rows = json.load(f)
    except (OSError, ValueError):
        return None, None
    row = rows.get("linear_mfma_kernel" + kern[len("linear"):])
    if not row or "hbm_bytes_corrected" not in row:
        return None, None
    return row["hbm_bytes_corrected"], "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the forward in a hipGraph")
    ap.add_argument("--split-bf16", action="store_true",
                    help="EXPLORATORY, not the headline: after_conv's products as 3-way split-bf16 MFMAs (fp32-accurate, fp32 accumulate)")
    ap.add_argument("--no-ops", action="store_true", help="skip the per-op roofline table (roofline.ops, cpu_baseline.ops)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    from dispu_amd.params import init_params

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # DISPU_BENCH_BACKEND=gloo lets the N > 1 code path be smoke-tested on a box with fewer GPUs than ranks (ranks then
    # share devices and the gather is staged through the host); the driver's real runs use nccl == RCCL, one GPU per rank.
    backend = os.environ.get("DISPU_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d visible GPUs" % (local, ndev))
    local_dev = local % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    params = init_params(seed=1234)                           # Xavier-uniform, zero biases (reference init)
    gen = Generator(params=params, device=dev)
    gen.return_views = True                                   # results stay in the workspace: no copy kernels in the step
    gen.split_bf16 = bool(args.split_bf16)
    x = torch.from_numpy(synth.patches(PATCHES_PER_GPU, NPOINT, seed=1000 * 2 + rank)).to(dev)   # 1000*config + rank
    gathered = torch.empty((world * PATCHES_PER_GPU, NPOINT * UP, 3), dtype=torch.float32, device=dev) if world > 1 else None

    from dispu_amd import parallel

    def step_eager():
        _, fine = gen(x)
        if world > 1:
            parallel.all_gather_clouds(fine, n_items=world * PATCHES_PER_GPU, out=gathered)
        return fine

    step_eager()
    torch.cuda.synchronize()
    launch = "eager"
    graph = None
    if not args.eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                gen(x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gen(x)                                         # generator only; the collective stays outside the graph
            launch = "hipgraph"
        except Exception as e:                                 # noqa: BLE001
            graph = None
            launch = "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()

    fine_buf = gen._ws[(PATCHES_PER_GPU, NPOINT)]["fine"]

    def step():
        if graph is not None:
            graph.replay()
            if world > 1:
                parallel.all_gather_clouds(fine_buf, n_items=world * PATCHES_PER_GPU, out=gathered)
        else:
            step_eager()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: HIP events around every launch, on the launch stream (eager pass)
    roof = None
    if rank == 0:
        reps = 5
        acc = {}
        for _ in range(reps):
            gen.profile = []
            gen(x)
            torch.cuda.synchronize()
            for name, e0, e1 in gen.profile:
                a = acc.setdefault(name, [0.0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3
                a[1] += 1
        gen.profile = None
        # group the instrumented launches by the kernel name rocprofv3 reports (one template instantiation each)
        by_kernel = {}
        for name, (t, c) in acc.items():
            kern = name.split("[")[0]
            g = by_kernel.setdefault(kern, [0.0, 0, 0.0])
            g[0] += t / reps
            g[1] += c / reps
            if name.startswith("linear<"):
                g[2] += linear_flops(name) * c / reps
        t_all = sum(g[0] for g in by_kernel.values())
        dom = max(by_kernel.items(), key=lambda kv: kv[1][0])              # dominant kernel = most time per step
        kern, (t_k, n_k, fl_k) = dom
        if not kern.startswith("linear<"):                                  # roofline is quoted on the MFMA GEMM
            kern, (t_k, n_k, fl_k) = max(((k, v) for k, v in by_kernel.items() if k.startswith("linear<")),
                                          key=lambda kv: kv[1][0])
        achieved = fl_k / t_k / 1e12
        traffic, traffic_src = pmc_traffic(kern)
        roof = {"bound": "mfma", "kernel": "dispu::linear_mfma_kernel" + kern[len("linear"):],
                "launches_per_step": round(n_k), "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": t_k / n_k * 1e6, "flops_per_launch": fl_k / n_k, "share_of_step": t_k / t_all,
                "per_kernel_us_per_step": {k: round(v[0] * 1e6, 1) for k, v in
                                           sorted(by_kernel.items(), key=lambda kv: -kv[1][0])[:8]}}

    if rank == 0:
        # step-level fraction of the fp32 MFMA peak: every flop the step executes (dense contractions are 97 % of them)
        ms = dt / args.steps * 1e3
        flops_step = 2.0 * step_macs_per_patch() * PATCHES_PER_GPU
        roof["step_flops"] = flops_step
        roof["step_frac"] = flops_step / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
        roof["step_frac_reference_graph"] = REFERENCE_FLOPS_PER_PATCH * PATCHES_PER_GPU / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
        roof["step_note"] = ("step_frac = flops executed per step (2 x %.3f G MAC per patch x %d patches) / ms_per_step / %.1f "
                             "TFLOP/s; step_frac_reference_graph prices the same step at the reference graph's %.3f GFLOP "
                             "per patch (work removed by exact algebra counted as done)"
                             % (step_macs_per_patch() / 1e9, PATCHES_PER_GPU, FP32_MFMA_PEAK_TFLOPS, REFERENCE_FLOPS_PER_PATCH / 1e9))
        if world == 1 and not args.no_ops:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ops_bench
            try:
                roof["ops"] = ops_bench.gpu_ops(dev, quick=True)
            except Exception as e:                             # noqa: BLE001 -- the headline line must survive a failing side table
                roof["ops"] = None
                roof["ops_error"] = "%s: %s" % (type(e).__name__, e)
            roof["ops_peaks"] = {"hbm_B_per_s": ops_bench.HBM_PEAK, "valu_lane_ops_per_s": ops_bench.VALU_PEAK,
                                 "exp_per_s": ops_bench.EXP_PEAK}
        pts = world * PATCHES_PER_GPU * NPOINT * UP
        out = {"metric": "upsampled points/sec (256->1024, 4x)", "value": pts * args.steps / dt, "unit": "points/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if not args.split_bf16 else "f32 storage; after_conv products 3-way split-bf16 (24-bit), fp32 accumulate [exploratory]",
               "data": "synthetic",
               "config": {"workload": "BASELINE configs[%d]: %d patches x %d points per GPU, generator forward 256->1024 (4x), "
                                      "fp32%s" % (1 if world == 1 else 2, PATCHES_PER_GPU, NPOINT,
                                                  "" if world == 1 else ", + %s all-gather of the upsampled clouds" % ("RCCL" if backend == "nccl" else backend)),
                          "patches_per_gpu": PATCHES_PER_GPU, "global_patches": world * PATCHES_PER_GPU,
                          "points_out_per_step": pts, "launch": launch, "weights": "xavier-uniform seed 1234, zero bias",
                          "parallelism": "patch-sharded x%d" % world},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(params, with_ops=not args.no_ops)
            except Exception as e:                             # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": 0, "kind": "port", "sample": "failed: %s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
