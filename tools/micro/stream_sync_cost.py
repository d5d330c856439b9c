"""Cost of a cross-stream dependency on MI355X (run on the GPU box): a chain of small kernels that alternates between two streams
through events, against the same chain on one stream; and the same with more streams alive (hardware-queue sharing)."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)


def chain(n, streams, hop):
    """n small kernels; every `hop`-th launch moves to the next stream (event record + wait)."""
    cur = 0
    ev = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[cur]):
            x.add_(1.0)
            if hop and (i + 1) % hop == 0:
                ev[i].record(streams[cur])
                cur = (cur + 1) % len(streams)
                streams[cur].wait_event(ev[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for extra in (0, 4):
    pool = [torch.cuda.Stream() for _ in range(2 + extra)]
    s2 = pool[:2]
    for hop in (0, 1, 4):
        for _ in range(2):
            t = chain(2000, s2, hop)
        print("streams alive %d, hop every %d launches: %.2f us per launch" % (2 + extra, hop, t))
    if extra:
        for _ in range(2):
            t = chain(2000, pool, 1)
        print("round robin over %d streams, hop every launch: %.2f us per launch" % (len(pool), t))

# GPU-side view: the same chains captured in a hipGraph (no host in the loop)
for hop in (0, 1, 4):
    g = torch.cuda.CUDAGraph()
    pool = [torch.cuda.Stream() for _ in range(2)]
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            cur = None
            strs = [cap, pool[0]]
            ci = 0
            for i in range(400):
                with torch.cuda.stream(strs[ci]):
                    x.add_(1.0)
                    if hop and (i + 1) % hop == 0:
                        e = torch.cuda.Event()
                        e.record(strs[ci])
                        ci = 1 - ci
                        strs[ci].wait_event(e)
            if ci == 1:
                e = torch.cuda.Event(); e.record(strs[1]); cap.wait_event(e)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("hipGraph, hop every %d launches: %.2f us per launch" % (hop, (time.perf_counter() - t0) / 4000 * 1e6))
