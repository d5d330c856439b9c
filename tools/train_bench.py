#!/usr/bin/env python3
"""Training-step benchmark (BASELINE configs[4] shape: full train step, data-parallel over the ranks, RCCL gradient
all-reduce; --dtype f32 (the reference's arithmetic) or bf16 (mixed precision, an extension: the reference is fp32-only)).  Not the driver's headline bench (that is bench.py, configs[1]).

    python tools/train_bench.py --batch 8 --steps 20                 # one GPU, 8 patches (config 5's per-GPU share)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py --batch 8

Prints one JSON line on rank 0: ms/step (max over ranks), patches/s (all ranks), and the per-phase split measured
with HIP events on rank 0 (forward / loss+its gradient / backward / all-reduce+Adam)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts: RCCL needs dmabuf IPC on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="patches per GPU")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32: the reference's arithmetic; bf16: bf16 products / fp32 accumulation and storage (csrc/linear_bf16.hip)")
    ap.add_argument("--dw-streams", type=int, default=0, help="side streams for the weight-gradient products (0: the Trainer's default)")
    ap.add_argument("--no-tail-on-chain", action="store_true", help="A/B: the last block's weight gradients on a side stream again")
    ap.add_argument("--no-bf16-stream", action="store_true", help="A/B: dtype bf16 without the streaming bf16 kernel (Trainer.bf16_stream = False)")
    ap.add_argument("--bf16-min-macs", type=float, default=0.0, help="Trainer.bf16_min_macs (A/B)")
    ap.add_argument("--no-bf16-tn", action="store_true", help="A/B: Trainer.bf16_tn = False")
    ap.add_argument("--no-group-reduce", action="store_true", help="A/B: every weight-gradient product launches its own split reduction (Trainer.group_reduce = False)")
    ap.add_argument("--tape", action="store_true", help="forward + loss + backward re-issued from a launch tape (Trainer.train_step_taped)")
    args = ap.parse_args()
    from dispu_amd import synth
    from dispu_amd.train import Trainer
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # weights: Xavier-uniform, seed 1234 (identical on every rank) -- generated without the oracle package
    from dispu_amd.params import init_params
    P = init_params(1234)
    tr = Trainer(params=P, device=dev, dtype=args.dtype)
    if args.dw_streams > 0:
        tr.dw_streams = args.dw_streams
    if args.bf16_min_macs > 0:
        tr.bf16_min_macs = args.bf16_min_macs
    if args.no_bf16_tn:
        tr.bf16_tn = False
    if args.no_bf16_stream:
        tr.bf16_stream = False
    if args.no_tail_on_chain:
        tr.tail_on_chain = False
    if args.no_group_reduce:
        tr.group_reduce = False
    x, gt = synth.patch_with_gt(args.batch, 256, 1024, seed=5000 + rank)
    x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    radius = torch.ones(args.batch, device=dev)

    step_fn = tr.train_step_taped if args.tape else tr.train_step
    for _ in range(args.warmup):
        step_fn(x, gt, radius)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = step_fn(x, gt, radius)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the same loop four more times: one 20-step loop is ~35 ms and single runs scatter by several per cent (host jitter of the
    # eager launch sequence); ms_per_step stays the FIRST loop's, min / median / max ride along
    loops = [dt]
    for _ in range(4):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_fn(x, gt, radius)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        d1 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([d1], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d1 = float(tt.item())
        loops.append(d1)
    srt = sorted(loops)

    phases = {}
    if rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        acc = np.zeros(4)
        for _ in range(5):
            tr.zero_grad()
            ev[0].record()
            tr.forward(x)
            ev[1].record()
            tr.loss_backward(gt, radius)
            ev[2].record()
            tr.backward()
            ev[3].record()
            tr.adam(tr.all_reduce_grads())
            ev[4].record()
            torch.cuda.synchronize()
            acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
        phases = dict(zip(("forward_ms", "loss_ms", "backward_ms", "allreduce_adam_ms"), (acc / 5).round(3).tolist()))
        print(json.dumps({"metric": "training patches/sec (256->1024 generator, full step)", "value": world * args.batch * args.steps / dt,
                          "unit": "patches/s", "n_gpus": world, "patches_per_gpu": args.batch, "steps": args.steps,
                          "ms_per_step": dt / args.steps * 1e3,
                          "ms_per_step_repeats": {"n": 5, "min": srt[0] / args.steps * 1e3, "median": srt[2] / args.steps * 1e3,
                                                  "max": srt[4] / args.steps * 1e3}, "launch": "tape" if args.tape else "eager",
                          "dtype": "f32" if args.dtype == "f32" else "bf16 products, f32 accumulate / storage", "pu_loss": float(terms["pu_loss"]), **phases}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
