#!/usr/bin/env python3
"""Where do the ~26 us go that the pipelined all-gather adds to the 32-patch step under a ONE-rank RCCL group?  (tests/test_distributed_gpu.py:
test_bench_one_rank_rccl_dry_run: 0.921 -> 0.948 ms.)  Same loop, one ingredient at a time: one graph; two alternating graphs (one per result
slot); + an event recorded after every replay and waited for by an idle side stream; + the collective itself on that side stream; the collective
replaced by a plain device copy of the same bytes."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import synth  # noqa: E402
from dispu_amd.generator import Generator  # noqa: E402
from dispu_amd.params import init_params  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
    gen = Generator(params=init_params(1234), device=dev)
    gen.return_views = True
    x = torch.from_numpy(synth.patches(32, 256, seed=2000)).to(dev)
    bufs = [torch.empty((32, 1024, 3), device=dev) for _ in range(2)]
    outs = [torch.empty((32, 1024, 3), device=dev) for _ in range(2)]
    for _ in range(3):
        gen(x)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gen(x)
    torch.cuda.current_stream().wait_stream(side)
    graphs = []
    for b in bufs:
        gen.fine_out = b
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gen(x)
        graphs.append(g)
    gen.fine_out = None
    lane = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    done = [None, None]

    def step(i, mode):
        s = i & 1 if mode != "one" else 0
        if done[s] is not None and mode not in ("one", "two") and not (mode.endswith("-q") and done[s].query()):
            cur.wait_event(done[s])
        graphs[s].replay()
        if mode in ("one", "two"):
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(lane):
            lane.wait_event(ev)
            if mode.startswith("gather"):
                dist.all_gather_into_tensor(outs[s], bufs[s])
            elif mode == "copy":
                outs[s].copy_(bufs[s])
            d = torch.cuda.Event()
            d.record(lane)
            done[s] = d

    def run(mode, n=200):
        for i in range(20):
            step(i, mode)
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            t = time.perf_counter()
            for i in range(n):
                step(i, mode)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t) / n * 1e3)
        return sorted(res)[2]

    for _ in range(2):
        for mode in ("one", "two", "event", "event-q", "copy", "gather", "gather-q"):
            print("%-9s %.4f ms per step" % (mode, run(mode)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
