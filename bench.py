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
SETTLE_STEPS = 100                    # untimed setup replays before the W warm-up steps (device clocks; see main())
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

# launch label (Generator._call) -> the kernel name rocprofv3 reports for it
ROCPROF_NAME = {"ps_local": "dispu::ps_local_ws_kernel", "mlp_chain[coarse]": "dispu::mlp_chain_kernel<256, 128, 256, 64>",
                "mlp_chain[fine]": "dispu::mlp_chain_kernel<256, 256, 256, 64>", "attention_project": "dispu::flash_attention_kernel<true>",
                "edge_dense_conv": "dispu::edge_dense_conv_mfma_kernel", "knn_feat": "dispu::knn_feat_wave_kernel",
                "knn_xyz": "dispu::knn_xyz_wave_kernel", "skip_max": "dispu::ps_skip_max16_kernel"}


def rocprof_name(kern):
    """launch label -> the kernel name rocprofv3 reports for it"""
    if kern.startswith("linear<"):
        return "dispu::linear_mfma_kernel" + kern[len("linear"):]
    if kern.startswith("linear_skinny<"):
        return "dispu::linear_skinny_kernel" + kern[len("linear_skinny"):]
    return ROCPROF_NAME.get(kern, kern)


def launch_flops(name, B):
    """ALGORITHMIC flops of one launch of the MFMA kernels of the step, by launch label (shapes: SURVEY.md Appendix A,
    Common/ops.py:1012-1087, :302-346, :1089-1110, :1856-1915).  None: not an MFMA kernel (k-NN selection, gathers, heads)."""
    n, m, k = NPOINT, NPOINT * UP, 16
    if name.startswith(("linear<", "linear_skinny<")):
        return linear_flops(name)
    if name == "ps_local":                       # conv1 128 -> 128 over 16 neighbours, weight net 3 -> 16, feature x weight 16 x 16 x 128
        return 2.0 * B * m * k * (128 * 128 + 3 * 16 + 16 * 128)
    if name == "mlp_chain[coarse]":
        return 2.0 * B * m * (256 * 128 + 128 * 256 + 256 * 64 + 64 * 3)
    if name == "mlp_chain[fine]":
        return 2.0 * B * m * (256 * 256 + 256 * 256 + 256 * 64 + 64 * 3)
    if name == "attention_project":              # logits + PV + conv_back_project
        return B * (4.0 * m * m * 64 + 2.0 * m * 64 * 256)
    return None


def cpu_baseline(target_seconds=14.0, with_ops=True):
    """The CPU oracle (oracle/generator.py + oracle/dispu_oracle.c, a port of the reference algorithm) timed on the host cores
    by oracle/cpu_bench.py in a SUBPROCESS (its worker processes must be forked from an interpreter without the HIP runtime):
    end to end with one thread and batch-parallel (one patch per single-threaded process over min(cores, 256) processes, the way
    the reference parallelises its only threaded CPU op, knn_.cxx:108), and per op with 1 thread / one cloud per thread."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--seconds", str(target_seconds)]
    if not with_ops:
        cmd.append("--no-ops")
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES"):
        env.pop(k, None)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    if r.returncode != 0:
        raise RuntimeError("oracle/cpu_bench.py failed: %s" % r.stderr.decode(errors="replace")[-400:])
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    best_all = d["points_per_s_all_cores"] >= d["points_per_s_1_thread"]
    out = {"value": max(d["points_per_s_all_cores"], d["points_per_s_1_thread"]), "unit": "points/s",
           "cores": d["processes"] if best_all else 1, "kind": "port", "cpu_model": d["cpu_model"], "host_cores": d["host_cores"],
           "points_per_s_1_thread": d["points_per_s_1_thread"], "points_per_s_all_cores": d["points_per_s_all_cores"],
           "sample": "%d patches of %d points (the bench's synthetic workload) through oracle/generator.py, one patch per "
                     "single-threaded process over %d processes, %.1f s; 1-thread figure from %.2f s per patch; host: %d cores, %s"
                     % (d["patches_timed"], NPOINT, d["processes"], d["seconds_timed"], d["s_per_patch_1_thread"],
                        d["host_cores"], d["cpu_model"])}
    if "ops" in d:
        out["ops"] = d["ops"]
    return out


def train_step_table(dev, steps=20, warmup=12):
    """Side table `roofline.train_step` (BASELINE configs[4]'s per-GPU share: 8 patches per GPU, full train step = training-mode forward,
    pu_loss, backward, gradient all-reduce (a no-op on one rank), Adam): ms per step (eager launches), fp32 and bf16, plus the
    B = 32 step; `mfma_frac` prices 3 x the forward's executed flops against the fp32 MFMA peak (a lower bound on the work: the backward
    recomputes the dense blocks and conv1).  Every row is tools/train_bench.py in a FRESH process: inside this one, after the headline /
    per-op benches and the Trainers of the earlier rows, the same step measured 10 - 15 % slower than on its own (2.2 vs 1.9 ms)."""
    import subprocess
    out = {}
    for dtype, B in (("f32", 8), ("bf16", 8), ("f32", 32)):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--batch", str(B), "--steps", str(steps), "--warmup", str(warmup),
               "--dtype", dtype]
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        key = "%s_b%d_eager" % (dtype, B)
        if r.returncode != 0:
            out[key] = {"error": r.stderr.decode(errors="replace")[-300:]}
            continue
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        # the MEDIAN of the five K-step loops train_bench.py times (after `warmup` untimed steps): a single 20-step loop of this
        # ~30 ms region scatters by +-4 % (first loop 1.706 vs median 1.644 ms in round 5), min / max ride along
        rep = d["ms_per_step_repeats"]
        ms = rep["median"]
        flops = 3.0 * 2.0 * step_macs_per_patch() * B
        out[key] = {"ms_per_step": round(ms, 4), "ms_min": round(rep["min"], 4), "ms_max": round(rep["max"], 4),
                    "ms_first_loop": round(d["ms_per_step"], 4), "patches_per_s": round(B / ms * 1e3, 1),
                    "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                    "forward_ms": d.get("forward_ms"), "loss_ms": d.get("loss_ms"), "backward_ms": d.get("backward_ms")}
    out["note"] = ("full train step (forward in training mode + pu_loss + backward + Adam) on one GPU, each row tools/train_bench.py in its own "
                   "process; ms_per_step = median of 5 loops of %d steps after %d warm-up steps (ms_min / ms_max / ms_first_loop beside it);" % (steps, warmup) + "  mfma_frac = 3 x forward flops / time / %.1f TFLOP/s fp32 MFMA peak; bf16 = bf16 products AND bf16 storage of the "
                   "local cell's pair tensors" % FP32_MFMA_PEAK_TFLOPS)
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
    # the row of exactly THIS kernel instantiation or nothing: no fall-back to a sibling instantiation of an older build
    row = rows.get("linear_mfma_kernel" + kern[len("linear"):])
    if not row or "hbm_bytes_corrected" not in row:
        return None, None
    meta = rows.get("_meta", {})
    src = "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("dispu_build", os.path.join(ROOT, "dis-pu_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        cur = mod.source_hash()
    except Exception:                                          # noqa: BLE001
        cur = None
    if meta.get("csrc_sha1") and cur:
        src += "; taken at commit %s, kernel sources %s this tree's" % (str(meta.get("git_head"))[:10],
                                                                       "==" if meta["csrc_sha1"] == cur else "!=")
    else:
        src += "; build of the counter passes not recorded"
    return row["hbm_bytes_corrected"], src + ")"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the forward in a hipGraph")
    ap.add_argument("--graph-only", action="store_true", help="always replay the hipGraph (default: setup picks the faster of replay and eager launches)")
    ap.add_argument("--split-bf16", action="store_true",
                    help="EXPLORATORY, not the headline: after_conv's products as 3-way split-bf16 MFMAs (fp32-accurate, fp32 accumulate)")
    ap.add_argument("--one-stream", action="store_true", help="Generator.branches = False: the non-local cell on the launch stream (profiling passes)")
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
    # DISPU_BENCH_COLLECTIVES=1 with --gpus 1: a ONE-rank process group, so that the N > 1 code (comm lane, events, the collective calls
    # themselves -- a 1-rank RCCL communicator really enqueues them) runs on the single GPU a test box has; the side tables are skipped
    force1 = world == 1 and os.environ.get("DISPU_BENCH_COLLECTIVES", "0") == "1"
    comm = world > 1 or force1
    solo = world == 1 and not force1
    if comm:
        kw = {}
        if force1 and "MASTER_ADDR" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            kw = dict(init_method="tcp://127.0.0.1:%d" % sk.getsockname()[1], rank=0, world_size=1)
            sk.close()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)

    params = init_params(seed=1234)                           # Xavier-uniform, zero biases (reference init)
    gen = Generator(params=params, device=dev)
    if args.one_stream:
        gen.branches = False
    gen.return_views = True                                   # results stay in the workspace: no copy kernels in the step
    gen.split_bf16 = bool(args.split_bf16)
    x = torch.from_numpy(synth.patches(PATCHES_PER_GPU, NPOINT, seed=1000 * 2 + rank)).to(dev)   # 1000*config + rank
    from dispu_amd import parallel

    # N > 1: the all-gather of step i runs on a comm lane (side HIP stream under RCCL) while step i + 1 computes: two result buffers,
    # the fine head writes its clouds into the slot's own [32, 1024, 3] buffer (Generator.fine_out), nothing waits on the compute
    # stream.  DISPU_BENCH_GATHER=sync keeps round 4's exposed gather (A/B), =off drops the collective (the compute-only reference the
    # two-rank dry run compares against).
    gather_mode = os.environ.get("DISPU_BENCH_GATHER", "overlap") if comm else "off"
    pipe = parallel.GatherPipeline((PATCHES_PER_GPU, NPOINT * UP, 3), dev) if gather_mode == "overlap" else None
    gathered = torch.empty((world * PATCHES_PER_GPU, NPOINT * UP, 3), dtype=torch.float32, device=dev) if gather_mode == "sync" else None

    def step_eager():
        if pipe is not None:
            slot, gen.fine_out = pipe.acquire()
        _, fine = gen(x)
        if pipe is not None:
            pipe.launch(slot)
        elif gathered is not None:
            parallel.all_gather_clouds(fine, n_items=world * PATCHES_PER_GPU, out=gathered)
        return fine

    def drain():
        if pipe is not None:
            pipe.drain()                                       # every gather launched so far is complete before the clock stops

    step_eager()
    step_eager()
    drain()
    torch.cuda.synchronize()
    launch = "eager"
    graphs = None
    if not args.eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                gen(x)
            torch.cuda.current_stream().wait_stream(side)
            graphs = []
            for buf in (pipe.local if pipe is not None else [None]):   # one graph per result slot (the output pointer is baked in)
                gen.fine_out = buf
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    gen(x)                                     # generator only; the collective stays outside the graph
                graphs.append(g)
            launch = "hipgraph"
        except Exception as e:                                 # noqa: BLE001
            graphs = None
            launch = "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()
    gen.fine_out = None

    fine_buf = gen._ws[(PATCHES_PER_GPU, NPOINT)]["fine"]

    def step():
        if graphs is None:
            step_eager()
        elif pipe is not None:
            slot, _ = pipe.acquire()
            graphs[slot].replay()
            pipe.launch(slot)
        else:
            graphs[0].replay()
            if gathered is not None:
                parallel.all_gather_clouds(fine_buf, n_items=world * PATCHES_PER_GPU, out=gathered)

    # setup, before the contract's W warm-up steps: the replays that bring the device to its sustained clocks (a cold device runs the
    # first ~20 steps 2 % slower than every later loop: ms_per_step_repeats of round 4); reported as "settle_steps"
    for _ in range(SETTLE_STEPS):
        step()
    drain()
    torch.cuda.synchronize()
    # still setup: which way of launching the step is faster on this box?  A hipGraph replay costs the host nothing, but this runtime's
    # graph executor overlaps the step's two streams less than eager submission does; the eager step needs ~17 launches of
    # host time per 0.94 ms.  40 steps each, the better one runs the warm-up and the timed steps; config.launch says which.
    calib = None
    if graphs is not None and not args.graph_only:
        def _time(fn, n=40):
            torch.cuda.synchronize()
            if comm:
                dist.barrier()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            drain()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        def _eager_step():
            step_eager()
        for _ in range(10):
            _eager_step()
        t_graph, t_eager = _time(step), _time(_eager_step)
        pick = torch.tensor([1.0 if t_eager < 0.995 * t_graph else 0.0], device=dev)
        if comm:                                          # every rank must take the same path
            dist.all_reduce(pick, op=dist.ReduceOp.MIN)
        calib = {"hipgraph_ms": t_graph * 1e3, "eager_ms": t_eager * 1e3}
        if float(pick.item()) > 0.5:
            graphs = None
            launch = "eager (two streams; picked over hipGraph replay in setup: %.4f vs %.4f ms per step)" % (t_eager * 1e3, t_graph * 1e3)
        else:
            launch = "hipgraph (picked over eager launches in setup: %.4f vs %.4f ms per step)" % (t_graph * 1e3, t_eager * 1e3)
    for _ in range(args.warmup):
        step()
    drain()
    if comm:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                    # the K-th step's gather is inside the timed region too
    torch.cuda.synchronize()
    if comm:
        dist.barrier()
    dt = time.perf_counter() - t0
    if comm:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # variance of the figure above: the same K-step loop four more times (20 steps are ~20 ms); `value` / `ms_per_step` stay the FIRST
    # loop's (the contract's timed region), min / median / max of all five ride along
    loops = [dt]
    for _ in range(4):
        if comm:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        torch.cuda.synchronize()
        if comm:
            dist.barrier()
        d1 = time.perf_counter() - t1
        if comm:
            tt = torch.tensor([d1], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d1 = float(tt.item())
        loops.append(d1)

    # ---- the same step with after_conv on the bf16 matrix pipe (operands split exactly into three bf16 terms, six partial products per
    # k, fp32 accumulate: csrc/linear_bf16x3.hip).  fp32-accurate (tests/test_headline_gpu.py) but NOT the fmaf chain of the strict path,
    # so it rides along as a second figure; `value` above stays the strict-fp32 step.
    alt = None
    if rank == 0 and solo and not args.split_bf16 and not args.eager:
        try:
            g2 = Generator(params=params, device=dev)
            g2.return_views = True
            g2.split_bf16 = True
            g2(x)
            torch.cuda.synchronize()
            side2 = torch.cuda.Stream()
            side2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side2):
                g2(x)
            torch.cuda.current_stream().wait_stream(side2)
            gr2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr2):
                g2(x)
            for _ in range(SETTLE_STEPS // 2):
                gr2.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    gr2.replay()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) / args.steps)
            ts.sort()
            te = []
            for _ in range(10):
                g2(x)
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    g2(x)
                torch.cuda.synchronize()
                te.append((time.perf_counter() - t1) / args.steps)
            te.sort()
            alt_launch = "hipgraph"
            if te[1] < ts[1]:
                ts, alt_launch = te, "eager"
            alt = {"ms_per_step": ts[1] * 1e3, "value": PATCHES_PER_GPU * NPOINT * UP / ts[1], "unit": "points/s", "launch": alt_launch,
                   "dtype": "f32 storage and accumulate; after_conv's products as six bf16 MFMAs per k over exact 3-term splits of both operands",
                   "note": "opt-in (Generator.split_bf16 / bench.py --split-bf16); median of 3 loops of K steps; coarse bit-exact, fine <= 1e-5 vs the oracle"}
            del gr2, g2
        except Exception as e:                                 # noqa: BLE001
            alt = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- serving-loop figure: two 32-patch batches in flight (two generators = two workspaces, alternating streams, eager): the next
    # batch's latency-bound feature extractor fills the tail of the previous one.  Strict fp32, same kernels, same results per batch;
    # NOT `value` (the contract's step is one batch from its first kernel to its last) -- reported beside it.
    alt2 = None
    if rank == 0 and solo and not args.split_bf16 and not args.eager:
        try:
            gens = [gen, Generator(params=params, device=dev)]
            gens[1].return_views = True
            strs = [torch.cuda.Stream(), torch.cuda.Stream()]
            xs2 = [x, x.clone()]

            def run_pairs(n):
                for i in range(n):
                    j = i & 1
                    with torch.cuda.stream(strs[j]):
                        gens[j](xs2[j])
            for s_ in strs:
                s_.wait_stream(torch.cuda.current_stream())
            run_pairs(20)
            torch.cuda.synchronize()
            ts2 = []
            for _ in range(3):
                t1 = time.perf_counter()
                run_pairs(2 * args.steps)
                torch.cuda.synchronize()
                ts2.append((time.perf_counter() - t1) / (2 * args.steps))
            ts2.sort()
            alt2 = {"ms_per_step": ts2[1] * 1e3, "value": PATCHES_PER_GPU * NPOINT * UP / ts2[1], "unit": "points/s",
                    "note": "two 32-patch batches in flight on alternating streams and workspaces (eager), median of 3 loops of 2K steps; "
                            "per-batch results identical to the single-batch step; a serving-loop option, not the contract's step"}
            del gens, strs, xs2
        except Exception as e:                                 # noqa: BLE001
            alt2 = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- roofline of the dominant kernel: HIP events around every launch, on the launch stream (eager pass)
    roof = None
    if rank == 0:
        reps = 5
        acc = {}
        # every kernel ALONE on the device: the step's second stream (Generator.branches: the non-local cell beside the local cell) is
        # folded back into the launch stream for this pass, so a kernel's time is its own and not that of two kernels sharing the CUs
        # (same kernels, same data; tools/prof_bench.sh runs its rocprofv3 pass the same way, --one-stream)
        two_streams, gen.branches = gen.branches, False
        for _ in range(reps):
            gen.profile = []
            gen(x)
            torch.cuda.synchronize()
            for name, e0, e1 in gen.profile:
                a = acc.setdefault(name, [0.0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3
                a[1] += 1
        gen.profile = None
        gen.branches = two_streams
        # group the instrumented launches by the kernel name rocprofv3 reports (one template instantiation each)
        by_kernel = {}
        edge_c = [24, 48, 48, 48]                                           # dense block d reads C = 24 / 48 / 48 / 48 channels
        edge_seen = 0
        for name, (t, c) in acc.items():
            kern = name.split("[")[0] if name.startswith(("linear<", "linear_skinny<")) else name
            g = by_kernel.setdefault(kern, [0.0, 0, 0.0, True])
            g[0] += t / reps
            g[1] += c / reps
            fl = launch_flops(name, PATCHES_PER_GPU)
            if name == "edge_dense_conv":                                   # four launches share the label: l0 / l1 / l2 over [n, 16] pairs
                fl = sum(2.0 * PATCHES_PER_GPU * NPOINT * 16 * 24 * (4 * C + 72) for C in edge_c) / 4.0
            if fl is None:
                g[3] = False
            else:
                g[2] += fl * c / reps
        t_all = sum(g[0] for g in by_kernel.values())
        # dominant kernel = the launch label with the most time per step, whatever it is (round-2 code filtered on `linear<`)
        kern, (t_k, n_k, fl_k, is_mfma) = max(by_kernel.items(), key=lambda kv: kv[1][0])
        kname = rocprof_name(kern)
        achieved = fl_k / t_k / 1e12 if is_mfma else None
        traffic, traffic_src = pmc_traffic(kern) if kern.startswith("linear<") else (None, None)
        roof = {"bound": "mfma" if is_mfma else "valu", "kernel": kname,
                "launches_per_step": round(n_k), "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS if achieved is not None else None, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": t_k / n_k * 1e6, "flops_per_launch": fl_k / n_k, "share_of_step": t_k / t_all,
                "per_kernel_us_per_step": {k: round(v[0] * 1e6, 1) for k, v in
                                           sorted(by_kernel.items(), key=lambda kv: -kv[1][0])[:8]}}
        # the decomposition of step_frac: every launch label of the step with its time, algorithmic flops and MFMA fraction
        roof["kernels"] = [
            {"kernel": rocprof_name(k),
             "launches_per_step": round(v[1]), "us_per_step": round(v[0] * 1e6, 2), "avg_launch_us": round(v[0] / max(v[1], 1) * 1e6, 2),
             "gflop_per_step": round(v[2] / 1e9, 3) if v[3] else None,
             "mfma_frac": round(v[2] / v[0] / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if v[3] else None}
            for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0])]
        roof["kernels_note"] = ("HIP-event time of every launch of one eager step, grouped by kernel instantiation; mfma_frac = algorithmic flops "
                                "/ time / %.1f TFLOP/s (fp32 MFMA peak); null = not an MFMA kernel (k-NN selection, gathers, 3-wide heads)"
                                % FP32_MFMA_PEAK_TFLOPS)

    if rank == 0 and solo:
        # what the matrix pipe of THIS chip sustains for a GEMM-like instruction mix with real data (tools/micro/mfma_power.hip, built by
        # __graft_entry__.build(): 8 accumulator tiles per wave, 2 A + 4 B LDS fragment reads per 8 MFMAs, random operands, no global
        # traffic, no barriers, no loader waves): the practical ceiling next to the datasheet `peak` -- power management, not the kernel
        mb = os.path.join(ROOT, "tools", "micro", "bin", "mfma_power")
        if os.path.exists(mb):
            try:
                import subprocess
                torch.cuda.synchronize()
                r_ = subprocess.run([mb], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
                js = [l for l in r_.stdout.decode().splitlines() if l.startswith("{")]
                if js:
                    sp = json.loads(js[-1])
                    roof["sustained_peak"] = {"tflops": sp["mfma_lds_reads_random_data_tflops"], "mfma_only_tflops": sp["mfma_only_tflops"],
                                              "lds_reads_constant_data_tflops": sp["mfma_lds_reads_constant_data_tflops"],
                                              "frac_of_sustained": (roof["achieved"] / sp["mfma_lds_reads_random_data_tflops"]) if roof.get("achieved") else None,
                                              "note": "tools/micro/mfma_power.hip on this device: bare v_mfma_f32_32x32x2_f32 loop with the GEMM's LDS "
                                                      "fragment reads and random operands, 10 launches back to back; `peak` stays the datasheet figure"}
            except Exception as e:                             # noqa: BLE001
                roof["sustained_peak"] = {"error": "%s: %s" % (type(e).__name__, e)}
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
        side = {}
        if solo and not args.no_ops:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ops_bench
            try:
                side["ops"] = ops_bench.gpu_ops(dev, quick=True)
            except Exception as e:                             # noqa: BLE001 -- the headline line must survive a failing side table
                side["ops"] = None
                side["ops_error"] = "%s: %s" % (type(e).__name__, e)
            try:
                side["train_step"] = train_step_table(dev)
            except Exception as e:                             # noqa: BLE001
                side["train_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
            side["ops_peaks"] = {"hbm_B_per_s": ops_bench.HBM_PEAK, "valu_lane_ops_per_s_datasheet": ops_bench.VALU_PEAK_DATASHEET,
                                 "exp_per_s_datasheet": ops_bench.EXP_PEAK_DATASHEET, "valu_measured": ops_bench.VALU_PEAK,
                                 "exp_measured": ops_bench.EXP_PEAK,
                                 "note": "`frac` of a VALU / exp row is against the DATASHEET figure (256 CU x 4 SIMD x 32 lanes x 2.4 GHz); "
                                         "frac_measured_peak against the measured issue rate (tools/micro/valu_rate.hip)"}
        pts = world * PATCHES_PER_GPU * NPOINT * UP
        out = {"metric": "upsampled points/sec (256->1024, 4x)", "value": pts * args.steps / dt, "unit": "points/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if not args.split_bf16 else "f32 storage; after_conv products 3-way split-bf16 (24-bit), fp32 accumulate [exploratory]",
               "data": "synthetic",
               "config": {"workload": "BASELINE configs[%d]: %d patches x %d points per GPU, generator forward 256->1024 (4x), "
                                      "fp32%s" % (1 if world == 1 else 2, PATCHES_PER_GPU, NPOINT,
                                                  "" if not comm else ", + %s all-gather of the upsampled clouds" % ("RCCL" if backend == "nccl" else backend)),
                          "patches_per_gpu": PATCHES_PER_GPU, "global_patches": world * PATCHES_PER_GPU,
                          "points_out_per_step": pts, "launch": launch, "launch_calibration_ms": calib,
                          "collective": (None if not comm else
                                         {"overlap": "all-gather of step i on a comm lane while step i + 1 computes (two result slots, "
                                                     "parallel.GatherPipeline); the K-th gather completes inside the timed region",
                                          "sync": "all-gather on the compute stream after every step (round-4 behaviour)",
                                          "off": "none (compute-only reference)"}[gather_mode]),
                          "weights": "xavier-uniform seed 1234, zero bias",
                          "parallelism": "patch-sharded x%d" % world},
               "roofline": roof}
        if alt is not None:
            out["alt_split_bf16"] = alt
        if alt2 is not None:
            out["alt_two_in_flight"] = alt2
        srt = sorted(loops)
        out["ms_per_step_repeats"] = {"n": len(loops), "min": srt[0] / args.steps * 1e3, "median": srt[len(srt) // 2] / args.steps * 1e3,
                                      "max": srt[-1] / args.steps * 1e3, "note": "the timed K-step loop run 5 times; value / ms_per_step = the first"}
        if solo and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(with_ops=not args.no_ops)
            except Exception as e:                             # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": 0, "kind": "port", "sample": "failed: %s: %s" % (type(e).__name__, e)}
        # The ONE JSON line stays small enough for the driver to keep it whole (round 3's ~40 KB line lost its nested tables): scalars + the
        # top-8 kernels here; the full per-kernel list, the per-op roofline table, the train-step table and the per-op CPU baseline go to a
        # side file next to the rocprof summaries.
        side["kernels"] = roof.pop("kernels")
        side["kernels_note"] = roof.pop("kernels_note")
        roof["kernels_top8"] = side["kernels"][:8]
        if "cpu_baseline" in out and "ops" in out["cpu_baseline"]:
            side["cpu_baseline_ops"] = out["cpu_baseline"].pop("ops")
        if "train_step" in side and isinstance(side["train_step"], dict):
            roof["train_step_ms"] = {k: v.get("ms_per_step") for k, v in side["train_step"].items() if isinstance(v, dict)}
            roof["train_step_ms_min_median_max"] = {k: [v.get("ms_min"), v.get("ms_per_step"), v.get("ms_max")]
                                                    for k, v in side["train_step"].items() if isinstance(v, dict) and "ms_min" in v}
        side["headline"] = {k: out[k] for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus")}
        # only the full run writes the side file: the profiled / counter passes (--no-ops, timings inflated by the profiler) must not
        # overwrite it
        for d in ((os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")) if (solo and not args.no_ops) else ()):
            try:
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "bench_side_tables.json"), "w") as f:
                    json.dump(side, f, indent=1)
            except OSError:
                pass
        if solo and not args.no_ops:
            roof["side_tables"] = "profiles/bench_side_tables.json (keys: kernels, ops, train_step, ops_peaks, cpu_baseline_ops)"
        print(json.dumps(out))
    if pipe is not None:
        pipe.close()
    if comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
