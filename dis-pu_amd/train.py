"""Dis-PU training step on MI355X: generator forward in training mode, the reference's loss, hand-written backward,
gradient all-reduce and Adam.

Counterpart of DisPU/model.py: `Model.build_model` (:68-87: coarse/fine = G(input); pu_loss = 1000 CD(coarse) +
weight_fine(epoch) * 1000 CD(fine) + repulsion_w * repulsion(fine)), `setup_optimizer` (:158-178: staircase LR decay
over epochs, Adam beta1 = opts.beta on every generator variable) and the per-batch body of `train` (:215-232).
The reference obtains gradients from TF1 autodiff; here every backward op is an explicit C-ABI launch
(include/dispu_hip.h, "training step").  torch supplies device memory, the stream and torch.distributed only.

Execution differences from the reference (not results):
  * parameters, gradients and both Adam moments live in four flat fp32 buffers (1.05 M floats each): ONE RCCL
    all-reduce and ONE Adam launch per step;
  * concat / tile tensors are column slices of wide buffers (the dense blocks' edge tensor is one
    [B*N*16, 72+2C] matrix whose slices are the inputs and outputs of l0/l1/l2), their gradients likewise;
  * duplicate_up's 482-wide conv is evaluated per source point (as in inference) and differentiated in that form.
Training-mode forward values equal the inference path's except for BatchNorm (batch statistics here).
"""
import contextlib
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._util import f32, req
from .generator import BN_EPS, DENSE_BLOCKS, GROWTH, K_NEIGH, _Opts, gen_grid

BN_DECAY = 0.95      # DisPU/generator.py:39 bn_decay
BN = "refine/PointShuffle/weight_net/wconv0/bn/"
BN_CH = 16            # its channels (the weight net's 3 -> 16 conv, Common/ops.py:181-191)


class TrainOpts(_Opts):
    """the training-side defaults of DisPU/configs.py"""
    base_lr_g = 0.001
    beta = 0.9
    lr_decay = True
    decay_step = 30
    lr_decay_rate = 0.7
    lr_clip = 1e-6
    use_repulse = True
    repulsion_w = 1.0


# Streams are shared by every Trainer of a process (per device): HIP maps streams onto a few hardware queues, and a process that
# builds Trainer after Trainer (bench.py's side table, the test suite) would otherwise keep adding streams to them.  Trainers of one
# process run one after the other on the host thread, so sharing is safe: events order the work.  (Which of torch's pooled streams
# the step gets does not matter: skipping 0 - 7 of them first leaves the step at 1.945 ms.)
_STREAM_POOL = {}


def _pool_stream(device, kind, i):
    key = (torch.device(device).index or 0, kind, i)
    st = _STREAM_POOL.get(key)
    if st is None:
        with torch.cuda.device(device):
            st = _STREAM_POOL[key] = torch.cuda.Stream(device=device)
    return st


def weight_fine(epoch):
    """model.py:52-54 piecewise_constant(epoch, [10, 20, 30], [0.01, 0.1, 0.5, 1.0])."""
    return 0.01 if epoch <= 10 else 0.1 if epoch <= 20 else 0.5 if epoch <= 30 else 1.0


def learning_rate(opts, epoch):
    """model.py:160-170."""
    lr = opts.base_lr_g
    if opts.lr_decay:
        lr = max(opts.base_lr_g * opts.lr_decay_rate ** math.floor(epoch / opts.decay_step), opts.lr_clip)
    return lr


def _p(t, off=0):
    """Device pointer of element `off` of t's storage (in t's own element size: a bf16-stored activation advances 2 bytes per column)."""
    return _lib.C.c_void_p(t.data_ptr() + t.element_size() * off) if t is not None else _lib.C.c_void_p(0)


class Trainer(object):
    """Trainer(opts, params).train_step(input[B,N,3], gt[B,4N,3], radius[B]) -> dict of loss terms.

    `params`: the same name -> array mapping Generator.load_params takes (oracle/generator.py:layer_shapes names)."""

    def __init__(self, opts=None, params=None, device=None, process_group=None, dtype="f32", comm_thread=None, collectives_at_world_1=False):
        """dtype "f32": the reference's arithmetic (every product on the fp32 matrix pipe, bit-equal to an fmaf chain).
        dtype "bf16" (BASELINE configs[4]): mixed precision -- master weights, activations and gradients stay fp32 in HBM,
        the dense products (forward, dX, dW) round their operands to bf16 and run on v_mfma_f32_32x32x16_bf16 with fp32
        accumulation (csrc/linear_bf16.hip); the 3-wide coordinate heads, the loss, BatchNorm, softmax and Adam stay fp32."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self.bf16 = dtype == "bf16"
        self.opts = opts if opts is not None else TrainOpts()
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type == "cuda" and self.device.index is None:       # torch.device("cuda") != torch.device("cuda:0")
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.up_ratio = int(self.opts.up_ratio)
        if self.up_ratio != 4:
            raise NotImplementedError("the shipped generator graph is built for up_ratio 4")
        self.pg = process_group
        self.epoch = 0
        self.global_step = 0
        self._ws = {}
        self._scratch = {}                    # per stream: scratch of the split reductions / column sums
        self._bn_scratch = None
        self._stash_ready = False
        self._tapes = {}
        self.dw_streams = 2                   # side streams the weight-gradient products are spread over (round-robin)
        self._ar = None                       # parallel.BucketedAllReduce over flat_g (data parallel only, see _reducer)
        self._ar_armed = False                # True inside train_step(): backward() may start a bucket's all-reduce as soon as it is complete
        # a ONE-rank process group normally means "no collectives".  True keeps them (a 1-rank RCCL communicator executes the real
        # stream / event ordering of the data-parallel step: the dry run of tests/test_distributed_gpu.py on the one GPU a test box has)
        self.collectives_at_world_1 = bool(collectives_at_world_1)
        # None: the transport's default (RCCL: the launching thread enqueues the collectives).  True: a background thread does -- the step
        # is launch-bound at 8 patches per GPU and a collective call costs ~30 us of host time (parallel._Lane)
        self.comm_thread = comm_thread
        # weight-gradient products (dW = X^T dZ) are off the backward chain: only Adam reads them.  They run on a second
        # HIP stream next to the dX products that ARE the chain (both read dZ; at 8 patches neither fills 256 CUs alone).
        self.overlap_dw = True
        # dense blocks: forward = the fused inference kernel, backward = one recomputing kernel per block (csrc/edge_bwd.hip);
        # 0 = round 2's path through materialised edge tensors (A/B tests)
        self.fused_dense = True
        self.use_wt = True
        # dtype "bf16": the big activation / gradient tensors of the local cell are STORED as bf16 (0: fp32 storage, bf16 products only)
        self.bf16_storage = True          # dX products through per-step W^T copies (A/B switch)
        self._aux = []
        self._cur = "main"
        self._sides = []
        self._side_rr = -1
        self._fork_ev = None
        self._join_ev = None
        self._sched = 1
        self.fused_heads_bwd = True
        # non-local cell: flash-style attention forward (+ per-row log-sum-exp) and a recomputing backward (csrc/attention_train.hip);
        # 0 = round 3's path through a materialised [B, M, M] probability tensor (A/B tests, the parity twin)
        self.flash_attn = True
        self.bf16_min_macs = 1.5e9

        self._side_busy = []
        self._group = None                  # (event, side streams that already wait for it) inside _fork_group()
        # Side work (weight-gradient products, the non-local / skip branches) is QUEUED ON THE HOST AFTER the chain's kernels: every
        # launch costs ~10 us of Python / ctypes, and a chain kernel that is submitted behind a dozen side launches leaves the GPU's
        # main queue idle for that long (round 4, profiles/r04_a_train_timeline.txt: 145 us before the fused local cell, 137 us before
        # its backward).  Side work records its fork event where it belongs and is submitted later, at points where the main queue
        # holds enough work (Trainer._flush); 0 restores the in-place submission (A/B)
        # (default "auto": on up to 16 patches per step -- at 32 the chip is saturated by the chain's own kernels, side work submitted
        # later only lengthens the tail: 4.58 -> 4.63 ms; "1" / "0" force it)
        self.fused_stem = True     # one launch per dense block in the forward pass
        self.bf16_tn = True        # dtype="bf16": weight-gradient products of fp32-stored operands on the bf16 TN kernel too (False: they stay on the fp32 one)
        self.bf16_stream = True    # dtype="bf16": the large dense products on the streaming bf16 kernel (csrc/linear_bf16_stream.hip) where its shape rules hold
        self._packs = {}
        # the split reductions of the weight-gradient products are not launched one by one: every product describes the reduction it left
        # undone (dispu_tn_defer) and a stream's descriptors run as ONE launch when that stream is joined (_join, _bucket_point): ~20
        # launches of 4 - 25 us per step become 3.  False: every product reduces itself (A/B; bit-identical)
        self.group_reduce = True
        # the local cell's backward (feature x weight gradient, conv1's dX, conv0's gather) as ONE recomputing launch (csrc/ps_local_bwd.hip,
        # fp32 storage only): h1 / dz0 / wv / the inverted neighbour graph never exist in HBM.  Correct and tested, but 40 us slower per 8-patch
        # step than the five-launch path (1.595 vs 1.555 ms): OFF by default
        self.fused_local_bwd = False
        self._rg = {}                      # stream pointer -> [host descriptor array, entries pending, stream c_void_p]
        self._rg_dev = {}                  # descriptor table bytes -> device copy (content-addressed: tapes keep pointing at theirs)
        self.tail_on_chain = True  # the first dense block's weight gradients (the LAST work of the backward) stay on the chain's stream: no cross-stream hop in front of Adam
        self.prep_late = True     # zeroing / W^T copies for the backward behind the non-local branch's own kernels (0: in front of them, round 3)
        self.prep_on_side = False   # ... or on a weight-gradient stream during the forward (measured slower at 8 patches in fp32: 1.79 vs 1.71 ms)   # backward's zeroing / W^T copies on a dW stream during the forward
        self._defer_mode = "auto"
        self.defer_side = self._defer_mode != "0"
        self._deferred = []
        self._pending = set()               # branches whose submission is still in _deferred
        self._cap_events = []               # events created while a hipGraph is being captured (see _ev)
        self._aux_done = {}                 # branch i -> the completion event recorded at its last exit
        self.P = None
        if params is not None:
            self.load_params(params)

    # --------------------------------------------------------------------------------------------- parameters ----
    def _invalidate_recordings(self):
        """Launch tapes hold RAW device pointers (flat parameter / gradient / moment buffers, W^T copies,
        scratch, workspaces).  Whenever one of those buffers is re-allocated every recording is dropped, so the next
        train_step_taped records afresh instead of replaying launches onto freed memory.  The bf16 images of the weights
        (_stream) are keyed by the weight's address: dropped with the tapes that point at them."""
        self._tapes.clear()
        if getattr(self, "_packs", None):
            torch.cuda.synchronize(self.device)
            self._packs.clear()
        if getattr(self, "_rg_dev", None):               # device copies of reduction tables: only tapes (dropped above) and the step being
            torch.cuda.synchronize(self.device)          # recorded point at them; a moved scratch buffer makes new table contents anyway
            self._rg_dev.clear()

    def load_params(self, params):
        dev = self.device
        self._invalidate_recordings()
        self._ws.clear()
        if self._ar is not None:              # the reducer holds views of the old gradient buffer
            self._ar.close()
            self._ar = None
        names = [k for k in params if k.endswith(("/weights", "/biases", "/gamma", "/beta"))]
        sizes = [int(np.asarray(params[k]).size) for k in names]
        # every tensor starts on a 16-byte boundary inside the flat buffers
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 3) & ~3
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        # gradients and, behind them, the 2 x 16 BatchNorm moving statistics in ONE buffer: data parallel, both are reduced over the
        # replicas by the same two bucket collectives (a collective call costs ~30 us of host time whatever its size -- two more for
        # 128 bytes of statistics were 4 % of the 8-patch step, tests/test_distributed_gpu.py:test_rccl_branch_on_one_rank)
        self._red = torch.zeros(total + 2 * BN_CH, dtype=torch.float32, device=dev)
        self.flat_g = self._red[:total]
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.P, self.G, self.names = OrderedDict(), OrderedDict(), names
        for k, o, s in zip(names, offs, sizes):
            shp = np.asarray(params[k]).shape
            self.P[k] = self.flat_p[o:o + s].view(shp)
            self.G[k] = self.flat_g[o:o + s].view(shp)
            self.P[k].copy_(torch.from_numpy(np.ascontiguousarray(params[k], np.float32)))
        # W^T copies for the dX products (refreshed once per step by ONE launch, csrc/train_fused.hip:transpose_batched_kernel):
        # every 2-D weight whose transposed rows stay 16-byte aligned (K % 4 == 0)
        self.flat_pT = torch.zeros_like(self.flat_p)
        self.PT, desc = OrderedDict(), []
        for k, o, s in zip(names, offs, sizes):
            shp = np.asarray(params[k]).shape
            if k.endswith("/weights") and len(shp) == 2 and shp[0] % 4 == 0 and shp[1] % 4 == 0 and shp[0] >= 16:
                self.PT[k] = self.flat_pT[o:o + s].view(shp[1], shp[0])
                desc += [o, shp[0], shp[1]]
        self._t_desc = torch.tensor(desc, dtype=torch.int32, device=dev)
        self._stats = self._red[total:]
        self.moving_mean, self.moving_var = self._stats[:BN_CH], self._stats[BN_CH:]
        self.moving_mean.copy_(torch.from_numpy(np.ascontiguousarray(params[BN + "moving_mean"], np.float32)))
        self.moving_var.copy_(torch.from_numpy(np.ascontiguousarray(params[BN + "moving_variance"], np.float32)))
        self.grid = torch.from_numpy(gen_grid(self.up_ratio)).to(dev)
        self.adam_t = 0
        # data parallel: the reducer (and, under gloo, its lane's process group -- dist.new_group is a collective) is built HERE, where
        # every rank passes in the same order, not lazily inside whichever of train_step / all_reduce_grads a rank reaches first
        self._reducer()

    def params(self):
        """current parameters as the name -> numpy mapping Generator.load_params / the oracle take."""
        out = OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.P.items())
        out[BN + "moving_mean"] = self.moving_mean.cpu().numpy().copy()
        out[BN + "moving_variance"] = self.moving_var.cpu().numpy().copy()
        return out

    def grads(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.G.items())

    # ---------------------------------------------------------------------------------------------- workspace ----
    def _workspace(self, B, N):
        key = (B, N)
        if key in self._ws:
            return self._ws[key]
        dev, f32, i32 = self.device, torch.float32, torch.int32
        M = N * self.up_ratio
        rn, rm, k = B * N, B * M, K_NEIGH
        E = lambda *shape, dtype=f32: torch.empty(shape, dtype=dtype, device=dev)
        Z = lambda *shape: torch.zeros(shape, dtype=f32, device=dev)
        pt = torch.bfloat16 if (self.bf16 and self.bf16_storage) else f32      # storage type of the pair tensors / dF'
        ws = dict(
            feat=E(rn, 480), dfeat=E(rn, 480),
            prep=[None, None] + [E(rn, 48) for _ in range(2, DENSE_BLOCKS + 1)],
            # zero-filled once per step, off the chain (atomics accumulate into them): the dense blocks' input gradients (one per block:
            # the dW products on the side streams read them) and the skip branch's share of d(up128)
            zeroed=E((DENSE_BLOCKS - 1) * rn * 48 + rm * 128),
            kidx=[None] + [E(rn, k + 1, dtype=i32) for _ in range(DENSE_BLOCKS)],
            h256=E(rn, 256), dh256=E(rn, 256), gcode=E(rm, 2),
            up256=E(rm, 256), dup256=E(rm, 256), up128=E(rm, 128), dup128=E(rm, 128),
            c256=E(rm, 256), dc256=E(rm, 256), c64=E(rm, 64), dc64=E(rm, 64), coarse=E(B, M, 3), dcoarse=E(B, M, 3),
            psidx=E(rm, k, dtype=i32), inv_off=E(B, M + 1, dtype=i32), inv=E(B, M * k, dtype=i32),
            gmax=Z(rm, 144), dgmax=E(rm, 136), skip=E(rm, 256), dskip=E(rm, 256),
            # local cell: conv0 per source point (G, A), the pair tensors h0 / h1 / wv are RECOMPUTED for the backward pass
            gm=E(rm, 128), am=E(rm, 128), dG=E(rm, 128), dAneg=E(rm, 128),
            # (dtype "bf16": stored as bf16 -- the GEMMs that read them round their operands to bf16 anyway)
            h0=E(rm * k, 128, dtype=pt), h1=E(rm * k, 128, dtype=pt), dz1=E(rm * k, 128, dtype=pt), dz0=E(rm * k, 128, dtype=pt),
            wv=E(rm * k, 16), dwv=E(rm * k, 16),
            bn_stats=E(48), bn_scale=E(16), bn_shift=E(16), bn_sums=E(32),
            hp=E(rm, 2048), dhp=E(rm, 2048, dtype=pt), aft=E(rm, 256), daft=E(rm, 256),
            kv=E(rm, 128), dkv=E(rm, 128), q=E(rm, 64), dq=E(rm, 64), att=E(rm, 64), datt=E(rm, 64), lse=E(rm), dvec=E(rm),
            S=None if self.flash_attn else E(B, M, M), dS=None if self.flash_attn else E(B, M, M),
            nl=E(rm, 256), dnl=E(rm, 256), sum=E(rm, 256), dsum=E(rm, 256), agg=E(rm, 256), dagg=E(rm, 256),
            f256=E(rm, 256), df256=E(rm, 256), f64=E(rm, 64), df64=E(rm, 64), z=E(rm, 3), dz=E(rm, 3), fine=E(B, M, 3), dfine=E(B, M, 3),
            # loss
            cd=[dict(d_gt=E(B, M), i_gt=E(B, M, dtype=i32), d_pred=E(B, M), i_pred=E(B, M, dtype=i32), g_gt=E(B, M), g_pred=E(B, M),
                     dgt_unused=E(B, M, 3), rowmean=E(B), rowmax=E(B)) for _ in range(2)],      # one set per Chamfer term
            ball=E(B, M, 20, dtype=i32), ball_cnt=E(B, M, dtype=i32), rep=E(B, M), rowmean=E(B), rowmax=E(B),
            loss_vals=Z(8), r07=torch.full((B,), 0.07, dtype=f32, device=dev), zeros=Z(1))
        ws["dprep"] = [None, None] + [ws["zeroed"][i * rn * 48:(i + 1) * rn * 48].view(rn, 48) for i in range(DENSE_BLOCKS - 1)]
        ws["dup128s"] = ws["zeroed"][(DENSE_BLOCKS - 1) * rn * 48:].view(rm, 128)
        # grid code of duplicate_up: row (cloud*up + r)*N + i carries grid[r]
        ws["gcode"].view(B, self.up_ratio, N, 2).copy_(self.grid.view(1, self.up_ratio, 1, 2).expand(B, self.up_ratio, N, 2))
        self._ws[key] = ws
        return ws

    def _edge_buffers(self, B, N):
        """edge tensors [B*N*16, 72 + 2C] and their gradients, one pair per dense block: only the UNFUSED dense-block path
        (Trainer.fused_dense = False, kept for A/B tests) materialises them."""
        key = ("edge", B, N)
        if key not in self._ws:
            rows = B * N * K_NEIGH
            mk = lambda: [None, torch.empty((rows, 72 + 48), dtype=torch.float32, device=self.device)] + \
                [torch.empty((rows, 72 + 96), dtype=torch.float32, device=self.device) for _ in range(2, DENSE_BLOCKS + 1)]
            self._ws[key] = (mk(), mk())
        return self._ws[key]

    def _scratch_floats(self, n, key=None):
        """scratch of the launches queued on ONE stream (they run in order, so they can share it): a dW stream's (key "dw<i>"), or the
        current stream's (main or a branch)."""
        key = key if key else self._cur
        cur = self._scratch.get(key)
        if cur is None or cur.numel() < n:
            if cur is not None:
                torch.cuda.synchronize(self.device)          # a launch on that stream may still be using the old buffer
                self._invalidate_recordings()                # ... and a tape / graph recorded at a smaller shape points into it
            cur = self._scratch[key] = torch.empty(max(int(n), 1 << 20), dtype=torch.float32, device=self.device)
        return cur

    # ---- side stream(s) for the weight-gradient products (Trainer.dw_streams of them, used round-robin; each has its own scratch)
    def _side_next(self):
        if not self._sides:
            n = max(1, int(self.dw_streams))
            self._sides = [_pool_stream(self.device, "dw", j) for j in range(n)]
            self._fork_ev = torch.cuda.Event()
            self._join_evs = [torch.cuda.Event() for _ in range(n)]
            self._side_busy = [False] * n
        self._side_rr = (self._side_rr + 1) % len(self._sides)
        return self._side_rr

    def _defer(self, fn, prio=1):
        """run the side-stream submission `fn` now, or at a later _flush() when deferral is on (only while side streams are in use).
        prio 0: branches the chain will wait for (non-local / skip / recompute); 1: weight gradients, read by Adam only."""
        if self.defer_side and self.overlap_dw:
            self._deferred.append((prio, fn))
        else:
            fn()

    def _flush(self, n=None, prio=1):
        """submit the deferred side launches of priority <= prio, in their order: all of them, or the first n."""
        i = 0
        while i < len(self._deferred) and (n is None or n > 0):
            if self._deferred[i][0] <= prio:
                self._deferred.pop(i)[1]()                # (may append: a branch defers its own weight gradients)
                if n is not None:
                    n -= 1
            else:
                i += 1
        if not any(p == 0 for p, _ in self._deferred):
            self._pending.clear()

    def _defer_branch(self, i, body, after=None):
        """`with self._branch(i, after): body()` -- submitted now, or at the next _flush() / _merge(i) when deferral is on.  The branch
        is ordered after `after`, or after THIS point of the current stream (the event is recorded now, whenever the body is submitted)."""
        if not (self.defer_side and self.overlap_dw):
            with self._branch(i, after):
                body()
            return
        ev = after if after is not None else self._fork_point()

        def run():
            with self._branch(i, ev):
                body()
        self._pending.add(i)
        self._deferred.append((0, run))

    # ---- stream / event operations (recorded on the launch tape as raw HIP calls when one is being taken, see train_step_taped)
    def _rec(self, ev, stream):
        ev.record(stream)
        t = _lib.taping()
        if t is not None:
            t.keep += [ev, stream]
            t.calls.append((_lib.lib().dispu_event_record, (ctypes.c_void_p(ev.cuda_event), ctypes.c_void_p(stream.cuda_stream)), "event_record"))

    def _wait(self, stream, ev):
        stream.wait_event(ev)
        t = _lib.taping()
        if t is not None:
            t.keep += [ev, stream]
            t.calls.append((_lib.lib().dispu_stream_wait_event, (ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(ev.cuda_event)),
                            "stream_wait_event"))

    def _zero(self, t):
        """t.zero_() on the stream launches currently go to, as a memset through the library (a plain C call: tape-able, and cheaper
        than the torch op)."""
        L = _lib.tape_lib()
        st = self.st if getattr(self, "st", None) is not None else _lib.stream_ptr(self.device)
        _lib.check(L.dispu_memset_async(_lib.C.c_void_p(t.data_ptr()), 0, t.numel() * t.element_size(), st), "memset")

    def _ev(self, cached):
        """the event to record at this point: the cached object in eager mode; a FRESH one while a hipGraph is being captured.
        Re-recording one event object several times inside a capture loses stream order on this runtime (round 4: the fused coarse-head
        backward and the dup_sum_grad launch after it, separated by five re-records of one fork event, ran unordered in the replayed
        graph -- stale d(up256), every gradient of the feature extractor wrong from the second replay on; eager order was never
        affected).  Fresh events are kept alive until the capture ends (train_step_graphed clears the list)."""
        if torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            self._cap_events.append(ev)
            return ev
        return cached

    def _fork(self):
        """-> (stream pointer, scratch key) for a dW product that may start once everything queued on the main stream so far is done."""
        i = self._side_next()
        if self._group is not None:
            # inside _fork_group(): every product of the group hangs off ONE event; a side stream waits for it once
            ev, waited = self._group
            if i not in waited:
                self._wait(self._sides[i], ev)
                waited.add(i)
        else:
            main = torch.cuda.current_stream(self.device)
            ev = self._ev(self._fork_ev)
            self._rec(ev, main)
            self._wait(self._sides[i], ev)
        self._side_busy[i] = True
        return ctypes.c_void_p(self._sides[i].cuda_stream), "dw%d" % i

    @contextlib.contextmanager
    def _fork_group(self):
        """Several dW products that all depend on the SAME point of the current stream (the four weight gradients of a fused head
        chain): one event, one wait per side stream.  Besides saving events this is what keeps the step correct under hipGraph replay:
        captured as record / wait / record / wait ... with nothing launched on the main stream in between (redundant edges from one
        node to a chain of side-stream nodes), the replayed graph ran the NEXT main-stream kernel without waiting for its predecessor
        (round 4, tools/debug/graph_vs_eager.py: stale d(up256) -> every gradient of the feature extractor wrong from the second
        replay on; eager launches were never affected)."""
        if not self.overlap_dw:
            yield
            return
        self._side_next()                                  # make sure the side streams exist
        self._side_rr -= 1
        self._group = (self._fork_point(), set())
        try:
            yield
        finally:
            self._group = None

    def _fork_point(self):
        """an event at the current position of the main stream, for a dW product queued later (see _lin_bwd)."""
        ev = torch.cuda.Event()
        if torch.cuda.is_current_stream_capturing():
            self._cap_events.append(ev)
        self._rec(ev, torch.cuda.current_stream(self.device))
        return ev

    def _fork_after(self, ev):
        i = self._side_next()
        self._wait(self._sides[i], ev)
        self._side_busy[i] = True
        return ctypes.c_void_p(self._sides[i].cuda_stream), "dw%d" % i

    # ---- grouped split reductions (dispu_tn_defer / dispu_tn_reduce_grouped) ----
    _RG_MAX = 64

    def _rg_slot(self, st, out_ptr, bias_ptr):
        """-> (address of the next free descriptor of stream `st`, its group, out pointer, bias pointer).  A product whose destination
        another pending reduction also accumulates into flushes that group first (two descriptors of one launch must not alias)."""
        for g in self._rg.values():
            if g[1] and any(o == out_ptr or (bias_ptr and b == bias_ptr) for o, b in g[3]):
                self._rg_flush_group(g)
        g = self._rg.get(st.value)
        if g is None:
            g = self._rg[st.value] = [(_lib.TnReduceDesc * self._RG_MAX)(), 0, ctypes.c_void_p(st.value), []]
        if g[1] >= self._RG_MAX:
            self._rg_flush_group(g)
        return (ctypes.c_void_p(ctypes.addressof(g[0]) + g[1] * ctypes.sizeof(_lib.TnReduceDesc)), g, out_ptr, bias_ptr)

    def _rg_commit(self, slot):
        _, g, out_ptr, bias_ptr = slot
        if g[0][g[1]].splits > 0:                        # the product left a reduction behind (0: it wrote its result itself)
            g[1] += 1
            g[3].append((out_ptr, bias_ptr))

    def _rg_flush_group(self, g):
        n = g[1]
        if not n:
            return
        raw = ctypes.string_at(ctypes.addressof(g[0]), n * ctypes.sizeof(_lib.TnReduceDesc))
        dev = self._rg_dev.get(raw)
        if dev is None:                                  # first step with this table (steady state: the same pointers every step)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("a new reduction table inside a hipGraph capture: run one eager step with this batch shape first")
            dev = self._rg_dev[raw] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        _lib.check(_lib.tape_lib().dispu_tn_reduce_grouped(n, ctypes.c_void_p(ctypes.addressof(g[0])), ctypes.c_void_p(dev.data_ptr()), g[2]),
                   "dispu_tn_reduce_grouped")
        g[1] = 0
        g[3] = []

    def _rg_flush(self):
        for g in self._rg.values():
            self._rg_flush_group(g)

    def _join(self):
        """the current stream waits for every dW product queued so far (before a buffer they read is overwritten, before Adam)."""
        self._flush()
        self._rg_flush()                                 # each stream's pending split reductions: one launch per stream, behind its products
        for i, busy in enumerate(self._side_busy if self._sides else []):
            if busy:
                ev = self._ev(self._join_evs[i])
                self._rec(ev, self._sides[i])
                self._wait(torch.cuda.current_stream(self.device), ev)
                self._side_busy[i] = False

    @contextlib.contextmanager
    def _branch(self, i, after=None):
        """Launches inside run on auxiliary stream i, after everything queued on the current stream so far (or after the event
        `after`, recorded earlier on it); `_merge(i)` makes
        the current stream wait for them.  At 8 patches per GPU a chain of 10 us kernels leaves most of the 256 CUs idle:
        independent sub-graphs (non-local cell | skip + local cell; the two Chamfer terms) run side by side."""
        if not self.overlap_dw:
            yield
            return
        while len(self._aux) <= i:
            self._aux.append((_pool_stream(self.device, "aux", len(self._aux)), torch.cuda.Event(), torch.cuda.Event()))
        aux, ev_fork, ev_done = self._aux[i]
        if after is None:
            after = self._ev(ev_fork)
            self._rec(after, torch.cuda.current_stream(self.device))
        self._wait(aux, after)
        old_st, old_cur = self.st, self._cur
        with torch.cuda.stream(aux):
            self.st, self._cur = ctypes.c_void_p(aux.cuda_stream), "aux%d" % i
            try:
                yield
            finally:
                self.st, self._cur = old_st, old_cur
                # recorded HERE, not in _merge: streams share hardware queues (GPU_MAX_HW_QUEUES = 4), and a marker queued at merge
                # time lands behind whatever the other streams of that queue were given in between (measured: the local cell's
                # backward started 0.33 ms late, behind dW products it does not depend on)
                if self._sched & 1:
                    done = self._ev(ev_done)
                    self._rec(done, aux)
                    self._aux_done[i] = done

    def _merge(self, i):
        if i in self._pending:
            self._flush(prio=0)                          # a branch whose submission is still deferred cannot be waited for
        if self.overlap_dw and (i < len(self._aux) or i in self._aux_done):
            aux, _, ev_done = self._aux[i]
            if not (self._sched & 1):
                done = self._ev(ev_done)
                self._rec(done, aux)
                self._aux_done[i] = done
            self._wait(torch.cuda.current_stream(self.device), self._aux_done.get(i, ev_done))

    # ----------------------------------------------------------------------------------------------- helpers ----
    def _dl(self, batch, M, K, N, *rest):
        """dispu_linear, or its bf16-product twin when the trainer runs mixed precision (narrow 3-wide layers stay fp32)."""
        L = _lib.tape_lib()
        fn = L.dispu_linear_bf16 if self._use_bf16(batch, M, K, N) else L.dispu_linear
        return fn(batch, M, K, N, *rest)

    def _use_bf16(self, batch, M, K, N):
        """bf16 products for this GEMM?  Mixed precision is per product: the narrow 3-wide layers always stay fp32, and so do products
        below `bf16_min_macs` multiply-adds -- at 8192 rows the K <= 256 products are launch / prologue-bound and the bf16 kernel (fp32
        operands rounded on their way into LDS) is no faster than the fp32 one, often slower (24 - 29 us vs 15 - 24 us)."""
        return self.bf16 and K > 4 and N > 4 and float(batch) * M * K * N >= self.bf16_min_macs

    def _stream(self, M, K, N, X, xoff, W, woff, transpose, bias, act, Y, yoff):
        """Y[:, yoff:yoff+N] = act(X[:, xoff:xoff+K] . B + bias) on the streaming bf16 kernel; B = W[woff..] as [K][N] (transpose = 1: the
        forward product) or W^T with W[woff..] as [N][K] (transpose = 0: dX = dZ . W^T).  Returns False -- nothing launched -- when the
        kernel's shape rules do not hold (the caller keeps dispu_linear_bf16).  The weight is packed to bf16 [N][K] right here, on the
        same stream (weights change every step; 0.5 M elements at most)."""
        if not (self.bf16 and self.bf16_stream) or M % 128 or K % 32 or N % 128:
            return False
        xb, yb = X.dtype == torch.bfloat16, Y.dtype == torch.bfloat16
        es = 2 if xb else 4
        if (X.stride(0) * es) % 16 or (xoff * es) % 16 or X.stride(1) != 1 or Y.stride(1) != 1 or W.stride(1) != 1:
            return False
        L = _lib.tape_lib()
        key = (W.data_ptr() + 4 * woff, K, N, transpose)
        bt = self._packs.get(key)
        if bt is None:
            bt = self._packs[key] = torch.empty((N, K), dtype=torch.bfloat16, device=self.device)
        if transpose:
            _lib.check(L.dispu_bf16_pack(K, N, _p(W, woff), W.stride(0), 1, _p(bt), self.st), "dispu_bf16_pack")
        else:
            _lib.check(L.dispu_bf16_pack(N, K, _p(W, woff), W.stride(0), 0, _p(bt), self.st), "dispu_bf16_pack")
        tiles = (M // 128) * (N // (256 if N % 256 == 0 else 128))
        nsp = 1
        if tiles < 192 and not yb:
            while nsp < 8 and tiles * nsp < 256 and K % (64 * nsp) == 0 and K // (2 * nsp) >= 256:
                nsp *= 2
        if nsp > 1:
            parts = self._scratch_floats(nsp * M * N)
            _lib.check(L.dispu_linear_bf16_stream(M, K, N, _p(X, xoff), X.stride(0), int(xb), _p(bt), K, None, 0, _p(parts), N, 0, nsp, M * N, self.st),
                       "dispu_linear_bf16_stream")
            _lib.check(L.dispu_linear_splitk_finish(M, N, nsp, _p(parts), M * N, _p(bias), act, _p(Y, yoff), Y.stride(0), self.st), "splitk_finish")
        else:
            _lib.check(L.dispu_linear_bf16_stream(M, K, N, _p(X, xoff), X.stride(0), int(xb), _p(bt), K, _p(bias), act, _p(Y, yoff), Y.stride(0),
                                                  int(yb), 1, 0, self.st), "dispu_linear_bf16_stream")
        return True

    def _lin(self, X, xoff, K, wname, act, Y, yoff, N, M=None, bias=True, W=None, woff=0):
        """Y[:, yoff:yoff+N] = act(X[:, xoff:xoff+K] . W + b)"""
        L = _lib.tape_lib()
        M = X.shape[0] if M is None else M
        W = self.P[wname + "/weights"] if W is None else W
        b = self.P[wname + "/biases"] if bias else None
        sto = (1 if X.dtype == torch.bfloat16 else 0) | (4 if Y.dtype == torch.bfloat16 else 0)
        if (sto or self._use_bf16(1, M, K, N)) and self._stream(M, K, N, X, xoff, W, woff, 1, b, act, Y, yoff):
            return
        if sto:
            assert xoff == 0 and yoff == 0
            _lib.check(L.dispu_linear_bf16s(1, M, K, N, _p(X), X.stride(0), 0, _p(W, woff), W.stride(0), 0, 0, _p(b), act, _p(Y), Y.stride(0), 0,
                                            None, 0, 0, sto, self.st), "dispu_linear_bf16s")
            return
        _lib.check(self._dl(1, M, K, N, _p(X, xoff), X.stride(0), 0, _p(W, woff), W.stride(0), 0, 0, _p(b), act,
                                  _p(Y, yoff), Y.stride(0), 0, None, 0, 0, None, 0, 0, self.st), "dispu_linear")

    def _tn(self, batch, M, K, N, X, xoff, ldx, sx, Zt, zoff, ldz, sz, out, ooff, ldo, so, accumulate, dbias=None, side=False, after=None,
            wgrad=False):
        """out (+)= X^T . Zt.  side=True: on a side stream (the caller guarantees nothing overwrites X / Zt before _join), ordered after
        `after` / the enclosing _fork_group's event / this point of the current stream; the launch itself may be deferred (_defer).
        wgrad=True: `out` is a weight gradient -- nothing reads it before _join(), so its split reduction may wait for the stream's
        grouped launch (Trainer.group_reduce)."""
        L = _lib.tape_lib()
        side = side and self.overlap_dw
        sto = (1 if X.dtype == torch.bfloat16 else 0) | (2 if Zt.dtype == torch.bfloat16 else 0)
        bf = (self.bf16_tn and self._use_bf16(batch, M, K, N)) or bool(sto)
        need = (L.dispu_linear_tn_bf16_scratch_floats if bf else L.dispu_linear_tn_scratch_floats)(batch, M, K, N)
        # the streaming TN kernel where its shape rules hold (after_conv's 2048 x 256: 37 vs 59 us at 8192 rows; the pair tensors' 128 x 128
        # over 131072 bf16-stored rows: 22 vs 47 us)
        tn_stream = 0
        if (bf and self.bf16_stream and batch == 1 and sto in (0, 3) and K % 128 == 0 and N % 128 == 0 and xoff == 0 and zoff == 0):
            tn_stream = L.dispu_linear_tn_bf16_stream_scratch_floats(M, K, N)
            need = max(need, tn_stream)

        def launch(st, key):
            slot = None
            if self.group_reduce and wgrad and batch == 1 and (key is not None or self._cur == "main"):
                # a scratch buffer of this product's own (it must survive until the grouped reduction) and a descriptor slot of its stream
                key = ("tn", out.data_ptr() + 4 * ooff, K, N)
                slot = self._rg_slot(st, out.data_ptr() + 4 * ooff, dbias.data_ptr() if dbias is not None else 0)
            sc = self._scratch_floats(need, key)
            if slot is not None:
                _lib.check(L.dispu_tn_defer(slot[0]), "dispu_tn_defer")
            try:
                launch_product(st, sc)
            finally:
                if slot is not None:
                    self._rg_commit(slot)

        def launch_product(st, sc):
            if tn_stream:
                _lib.check(L.dispu_linear_tn_bf16_stream(M, K, N, _p(X), ldx, _p(Zt), ldz, sto, _p(out, ooff), ldo, accumulate, _p(dbias), _p(sc),
                                                         sc.numel(), st), "dispu_linear_tn_bf16_stream")
                return
            if sto:
                assert bf and xoff == 0 and zoff == 0
                _lib.check(L.dispu_linear_tn_bf16s(batch, M, K, N, _p(X), ldx, sx, _p(Zt), ldz, sz, _p(out, ooff), ldo, so, accumulate, _p(dbias), _p(sc),
                                                   sc.numel(), sto, st), "dispu_linear_tn_bf16s")
                return
            fn = L.dispu_linear_tn_bf16 if bf else L.dispu_linear_tn
            _lib.check(fn(batch, M, K, N, _p(X, xoff), ldx, sx, _p(Zt, zoff), ldz, sz, _p(out, ooff), ldo, so, accumulate, _p(dbias), _p(sc),
                          sc.numel(), st), "dispu_linear_tn_bf16" if bf else "dispu_linear_tn")

        if not side:
            launch(self.st, None)
            return
        grp = self._group
        ev = after if after is not None else (None if grp is not None else self._fork_point())

        def submit():
            if ev is not None:
                st, key = self._fork_after(ev)
            else:                                          # member of a _fork_group: one wait per side stream for the group's event
                i = self._side_next()
                gev, waited = grp
                if i not in waited:
                    self._wait(self._sides[i], gev)
                    waited.add(i)
                self._side_busy[i] = True
                st, key = ctypes.c_void_p(self._sides[i].cuda_stream), "dw%d" % i
            launch(st, key)

        self._defer(submit)

    def _act_bias_grad(self, M, N, dY, dyoff, Y, yoff, act, dZ, dzoff, dbias):
        L = _lib.tape_lib()
        need = L.dispu_act_bias_grad_scratch_floats(M, N)
        sc = self._scratch_floats(need)
        _lib.check(L.dispu_act_bias_grad(M, N, _p(dY, dyoff), dY.stride(0), _p(Y, yoff) if Y is not None else None,
                                         Y.stride(0) if Y is not None else 0, act, _p(dZ, dzoff) if dZ is not None else None,
                                         dZ.stride(0) if dZ is not None else 0, _p(dbias), 1, _p(sc), sc.numel(), self.st),
                   "dispu_act_bias_grad")

    def _dx(self, M, N, K, dY, dyoff, W, woff, dX, dxoff, acc, mask=None, WT=None):
        """dX[:, dxoff:dxoff+K] (+)= dY[:, dyoff:dyoff+N] . W^T, then zeroed where mask <= 0: mask = (tensor, column offset, columns) is the
        ReLU output that fed this layer -- the relu_grad of the layer below rides in the GEMM epilogue (no separate pass over dX).
        WT: the step's transposed copy of W ([N, K] row-major): the product then runs untransposed (the forward GEMM's fast path)."""
        L = _lib.tape_lib()
        bf = self._use_bf16(1, M, N, K)
        r1 = _p(dX, dxoff) if acc else None
        ldr = dX.stride(0) if acc else 0
        if WT is not None:
            wp, ldw, tb = _p(WT), WT.stride(0), 0
        else:
            wp, ldw, tb = _p(W, woff), W.stride(0), 1
        sto = (1 if dY.dtype == torch.bfloat16 else 0) | (4 if dX.dtype == torch.bfloat16 else 0)
        if (sto or bf) and mask is None and not acc and self._stream(M, N, K, dY, dyoff, W, woff, 0, None, 0, dX, dxoff):
            return
        if sto:
            assert self.bf16 and mask is None and not acc and dyoff == 0 and dxoff == 0
            _lib.check(L.dispu_linear_bf16s(1, M, N, K, _p(dY), dY.stride(0), 0, wp, ldw, 0, tb, None, 0, _p(dX), dX.stride(0), 0, None, 0, 0,
                                            sto, self.st), "dispu_linear_bf16s(dX)")
            return
        if mask is None:
            fn = L.dispu_linear_bf16 if bf else L.dispu_linear
            _lib.check(fn(1, M, N, K, _p(dY, dyoff), dY.stride(0), 0, wp, ldw, 0, tb, None, 0, _p(dX, dxoff), dX.stride(0), 0,
                          r1, ldr, 0, None, 0, 0, self.st), "dispu_linear(dX)")
        else:
            mt, moff, mcols = mask
            fn = L.dispu_linear_bf16_masked if bf else L.dispu_linear_masked
            _lib.check(fn(1, M, N, K, _p(dY, dyoff), dY.stride(0), 0, wp, ldw, 0, tb, None, 0, _p(dX, dxoff), dX.stride(0), 0,
                          r1, ldr, 0, _p(mt, moff), mt.stride(0), int(mcols), self.st), "dispu_linear_masked(dX)")

    def _lin_bwd(self, X, xoff, K, wname, N, dY, dyoff, dX=None, dxoff=0, acc_dx=False, M=None, bias=True, W=None, dW=None, woff=0,
                 mask=None, db=None, side=True):
        """backward of _lin given dZ = dY[:, dyoff:dyoff+N] ALREADY multiplied by the layer's own relu' (its producer did that):
        db += colsum dZ and dW += X^T dZ on the second stream, dX[:, dxoff:dxoff+K] (+)= dZ . W^T with `mask` (see _dx)."""
        M = X.shape[0] if M is None else M
        W = self.P[wname + "/weights"] if W is None else W
        dW = self.G[wname + "/weights"] if dW is None else dW
        if db is None and bias:
            db = self.G[wname + "/biases"]
        # the dX product IS the backward chain: it is queued first; the dW product (read by Adam only) follows on the second stream,
        # ordered after dZ by an event recorded BEFORE the dX launch (dX never writes what the dW product reads)
        ev = self._fork_point() if (self.overlap_dw and dX is not None) else None
        if dX is not None:
            WT = self.PT.get(wname + "/weights") if (wname is not None and woff == 0 and self.use_wt and K == W.shape[0]) else None
            self._dx(M, N, K, dY, dyoff, W, woff, dX, dxoff, acc_dx, mask, WT)
        self._tn(1, M, K, N, X, xoff, X.stride(0), 0, dY, dyoff, dY.stride(0), 0, dW, woff, dW.stride(0), 0, 1, dbias=db, side=side, after=ev,
                 wgrad=True)

    # ----------------------------------------------------------------------------------------------- forward ----
    def forward(self, inputs):
        """training-mode forward (BatchNorm batch statistics, moving averages updated); keeps every activation."""
        if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 3
                and inputs.shape[2] == 3):
            raise ValueError("Trainer expects a float32 [B,N,3] tensor on a ROCm device")
        x = inputs.contiguous()
        B, N, _ = x.shape
        M, k = N * self.up_ratio, K_NEIGH
        rn, rm = B * N, B * M
        # shape limits of the fused training kernels, refused here instead of as a bare hipErrorInvalidValue mid-step:
        # dispu_mlp_chain_stash / dispu_mlp_chain_grad tile 64 rows; dispu_knn_invert sorts one cloud's graph in LDS
        req(N > K_NEIGH, "a patch needs more than %d points (k-NN graph of the dense blocks), got N = %d" % (K_NEIGH, N))
        req(rm % 64 == 0, "the fused head chains work on 64-row tiles: B * %d * N must be a multiple of 64, got B = %d, N = %d "
                          "(pad the batch)" % (self.up_ratio, B, N))
        req(M <= 4096, "the local cell's backward inverts a cloud's k-NN graph in LDS: %d * N <= 4096, got N = %d" % (self.up_ratio, N))
        req(M % 32 == 0, "the non-local cell's attention kernels work on 32-point tiles: %d * N must be a multiple of 32, got N = %d"
                         % (self.up_ratio, N))
        ws = self._workspace(B, N)
        L = _lib.tape_lib()
        self.st = _lib.stream_ptr(x.device)
        if self._defer_mode == "auto":
            # not while capturing: in a hipGraph only the dependencies count, and the executor orders the deferred nodes worse
            # (graphed 8-patch step 1.94 ms with the deferral, 1.88 ms without)
            self.defer_side = rm <= 16 * 1024 and not torch.cuda.is_current_stream_capturing()
        self._shape = (B, N)
        self._x = x
        if self.overlap_dw and self.prep_on_side:
            # what the backward needs and nothing in the forward touches (zeroed accumulators, the W^T copies: ~45 us of kernels) goes to a
            # weight-gradient stream, idle during the forward, right now; loss_backward() joins it.  (It used to ride at the head of the
            # non-local branch "because that has slack" -- since the flash attention the main stream waits 75 us for exactly that branch.)
            st_side, key = self._fork()
            self._side_rr -= 1                                       # the weight gradients keep their round-robin assignment
            main_st, self.st = self.st, st_side
            self._backward_prep(ws)
            self.st = main_st
            self._prep_side = int(key[2:])
        P = self.P
        feat = ws["feat"]
        fe = "generator/feature_extraction_coarse/"
        stem0 = self.fused_dense and self.fused_stem and N <= 256 and N % 2 == 0
        if not stem0:      # (fused stem: the first block's launch evaluates layer0 while it stages its cloud)
            _lib.check(L.dispu_linear_small_k(rn, 3, 24, _p(x), 3, _p(P[fe + "layer0/weights"]), _p(P[fe + "layer0/biases"]), 0,
                                              _p(feat, 456), 480, self.st), "layer0")
        col = 456
        self._blocks = []
        prep_done = False                      # the previous block's launch already ran this block's bottleneck conv (fused stem)
        for d in range(1, DENSE_BLOCKS + 1):
            if d == 1:
                F, foff, C = feat, 456, 24
            else:
                if not prep_done:
                    self._lin(feat, col, 480 - col, fe + "layer%d_prep" % d, 1, ws["prep"][d], 0, 48)
                F, foff, C = ws["prep"][d], 0, 48
            ldf = F.stride(0)
            kidx = ws["kidx"][d]
            sc = fe + "layer%d" % d
            width = 3 * GROWTH + C
            in_col = col
            col -= width
            stem = self.fused_dense and self.fused_stem and N <= 256 and N % 2 == 0
            if not stem:
                _lib.check(L.dispu_knn_feat_strided(B, N, N, C, k + 1, _p(F, foff), ldf, _p(F, foff), ldf, None, _p(kidx), self.st), "knn_feat")
            if stem:
                # search + edge features + dense_conv in one launch (csrc/edge.hip, KNN variant); the neighbour table stays for the backward pass
                _lib.check(L.dispu_stem_block(rn, N, C, _p(F, foff), ldf, k + 1, 1, _p(P[sc + "/l0/weights"]), _p(P[sc + "/l0/biases"]),
                                              _p(P[sc + "/l1/weights"]), _p(P[sc + "/l1/biases"]), _p(P[sc + "/l2/weights"]),
                                              _p(P[sc + "/l2/biases"]), _p(feat, col), 480, _p(kidx),
                                              _p(P[fe + "layer%d_prep/weights" % (d + 1)]) if d < DENSE_BLOCKS else None,
                                              _p(P[fe + "layer%d_prep/biases" % (d + 1)]) if d < DENSE_BLOCKS else None, 480 - in_col,
                                              _p(ws["prep"][d + 1]) if d < DENSE_BLOCKS else None, 48, _p(x) if d == 1 else None,
                                              _p(P[fe + "layer0/weights"]) if d == 1 else None, _p(P[fe + "layer0/biases"]) if d == 1 else None,
                                              _p(feat, 456) if d == 1 else None, 480, self.st), "stem_block")
            elif self.fused_dense:
                # the inference kernel (csrc/edge.hip): edge features, three chained convs and the max in one launch; nothing is kept
                # for the backward pass, which recomputes the block on chip (csrc/edge_bwd.hip)
                _lib.check(L.dispu_edge_dense_conv(rn, N, C, _p(F, foff), ldf, _p(kidx), k + 1, 1, _p(P[sc + "/l0/weights"]),
                                                   _p(P[sc + "/l0/biases"]), _p(P[sc + "/l1/weights"]), _p(P[sc + "/l1/biases"]),
                                                   _p(P[sc + "/l2/weights"]), _p(P[sc + "/l2/biases"]), _p(feat, col), 480, self.st),
                           "edge_dense_conv")
            else:
                Eb = self._edge_buffers(B, N)[0][d]
                _lib.check(L.dispu_edge_feature(rn, N, k, C, _p(F, foff), ldf, _p(kidx), k + 1, 1, _p(Eb, 72), Eb.stride(0), self.st), "edge_feature")
                self._lin(Eb, 72, 2 * C, sc + "/l0", 1, Eb, 48, 24)
                self._lin(Eb, 48, 24 + C, sc + "/l1", 1, Eb, 24, 24)
                self._lin(Eb, 24, 48 + C, sc + "/l2", 0, Eb, 0, 24)
                _lib.check(L.dispu_max_k(rn, k, width, _p(Eb), Eb.stride(0), _p(feat, col), 480, self.st), "max_k")
            prep_done = stem and d < DENSE_BLOCKS
            self._blocks.append((d, C, col, in_col, width))
        assert col == 0

        # duplicate_up (per source point) + coarse regressor
        w1 = P["generator/upshuffle_0/conv1/weights"]
        self._lin(feat, 0, 480, None, 0, ws["h256"], 0, 256, bias=False, W=w1)
        _lib.check(L.dispu_dup_grid(B, N, 256, self.up_ratio, _p(ws["h256"]), 256, _p(w1, 480 * 256),
                                    _p(P["generator/upshuffle_0/conv1/biases"]), _p(self.grid), _p(ws["up256"]), 256, self.st), "dup_grid")
        # conv2 -> fc_layer0 -> fc_layer1 -> fc_layer2 in ONE launch (csrc/mlp_chain.hip), every intermediate stashed for the backward
        cs = "generator/coarse_coordinate_regressor/"
        coarse = ws["coarse"].view(rm, 3)
        c2 = "generator/upshuffle_0/conv2"
        _lib.check(L.dispu_mlp_chain_stash(rm, 256, 128, 256, 64, _p(ws["up256"]), 256, _p(P[c2 + "/weights"]), _p(P[c2 + "/biases"]),
                                           _p(P[cs + "fc_layer0/weights"]), _p(P[cs + "fc_layer0/biases"]),
                                           _p(P[cs + "fc_layer1/weights"]), _p(P[cs + "fc_layer1/biases"]),
                                           _p(P[cs + "fc_layer2/weights"]), _p(P[cs + "fc_layer2/biases"]),
                                           _p(ws["up128"]), 128, _p(ws["c256"]), 256, _p(ws["c64"]), 64, None, 0, 0, None, 0,
                                           _p(coarse), 3, self.st), "mlp_chain[coarse]")
        ps = "refine/PointShuffle/"
        up128 = ws["up128"]
        # The k-NN graph of the coarse cloud heads the step's critical path (knn -> BatchNorm statistics -> fused local cell ->
        # after_conv -> fine head): its two launches are queued BEFORE the ~8 launches of the non-local branch, which has slack
        # (round 4: queued after them, the graph waited ~70 us for the host to get there -- tools/trace_timeline.py)
        ww, bw = P[ps + "weight_net/wconv0/weights"], P[ps + "weight_net/wconv0/biases"]
        nb = L.dispu_ps_wnet_scratch_bytes(rm)
        if self._bn_scratch is None or self._bn_scratch.numel() * 8 < nb:
            self._bn_scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=self.device)
        with self._branch(2):
            _lib.check(L.dispu_knn_xyz(B, M, M, k, _p(coarse), _p(coarse), _p(ws["psidx"]), None, _lib.ARITH_PLAIN, self.st), "knn_xyz")
            _lib.check(L.dispu_ps_wnet_bn_stats(rm, M, k, 16, _p(ws["psidx"]), _p(coarse), _p(ww), _p(bw), _p(P[BN + "gamma"]), _p(P[BN + "beta"]),
                                                BN_EPS, BN_DECAY, _p(ws["bn_stats"]), _p(ws["bn_scale"]), _p(ws["bn_shift"]),
                                                _p(self.moving_mean), _p(self.moving_var), _p(self._bn_scratch), self._bn_scratch.numel() * 8,
                                                self.st), "ps_wnet_bn_stats")
        # PointShuffle2 (ops.py:1012-1087) in the inference path's form: no [B, M, 16, 134] grouped tensor.  The k-NN graph and the
        # weight net's BatchNorm statistics (from the neighbour offsets) on a branch, next to the per-point halves of conv0
        # (G = up128.Wf + xyz.(Wc + Wr) + b0, A = xyz.Wc; h0 = relu(G[j] - A[i])) that need neither
        w0 = P[ps + "conv0/weights"]
        self._lin(up128, 0, 128, None, 0, ws["gm"], 0, 128, bias=False, W=w0, woff=6 * 128)
        _lib.check(L.dispu_ps_prep(rm, 128, _p(coarse), _p(w0), _p(P[ps + "conv0/biases"]), _p(ws["gm"]), 128, _p(ws["am"]), 128, self.st), "ps_prep")
        # non-local cell (its attention kept for the backward as O + one log-sum-exp per query).  It reads up128 only: a branch next to
        # the grouping, the skip and the local cell; merged before add3.  Its ~8 launches are SUBMITTED after the local cell and
        # after_conv below (the chain), but ordered after this point of the stream.
        S = ws["S"]

        def nl_branch():
            if not (self.overlap_dw and self.prep_on_side) and not self.prep_late:
                self._backward_prep(ws)
            self._lin(up128, 0, 128, ps + "PointShuffle/conv_kv", 0, ws["kv"], 0, 128)
            self._lin(up128, 0, 128, ps + "PointShuffle/conv_query", 0, ws["q"], 0, 64)
            if self.flash_attn:
                # softmax(Q.K^T / 8).V on chip; what the backward needs is att and ONE float per query (csrc/attention_train.hip)
                _lib.check(L.dispu_attention_fwd_lse(B, M, M, 64, _p(ws["q"]), 64, _p(ws["kv"]), 128, _p(ws["kv"], 64), 128, 0.125,
                                                     _p(ws["att"]), 64, _p(ws["lse"]), self.st), "attention_fwd_lse")
            else:
                _lib.check(self._dl(B, M, 64, M, _p(ws["q"]), 64, M * 64, _p(ws["kv"]), 128, M * 128, 1, None, 0, _p(S), M, M * M,
                                          None, 0, 0, None, 0, 0, self.st), "scores")
                _lib.check(L.dispu_softmax_rows(rm, M, 0.125, _p(S), M, self.st), "softmax")
                _lib.check(self._dl(B, M, M, 64, _p(S), M, M * M, _p(ws["kv"], 64), 128, M * 128, 0, None, 0, _p(ws["att"]), 64,
                                          M * 64, None, 0, 0, None, 0, 0, self.st), "att.V")
            self._lin(ws["att"], 0, 64, ps + "PointShuffle/conv_back_project", 1, ws["nl"], 0, 256)
            if not (self.overlap_dw and self.prep_on_side) and self.prep_late:
                self._backward_prep(ws)

        self._defer_branch(0, nl_branch)
        self._merge(2)

        # skip (a second branch next to the local cell): gather-max straight from xyz / up128, then 134 -> 256
        def skip_branch():
            _lib.check(L.dispu_ps_skip_max(rm, M, k, 128, _p(ws["psidx"]), _p(coarse), _p(up128), 128, _p(ws["gmax"]), 144, self.st), "skip_max")
            self._lin(ws["gmax"], 0, 134, ps + "skip", 1, ws["skip"], 0, 256)

        self._defer_branch(1, skip_branch)
        # the fused cell (conv1, weight_net with the folded BatchNorm, feature x weight)
        _lib.check(L.dispu_ps_local(rm, M, k, 128, _p(ws["psidx"]), _p(coarse), _p(ws["gm"]), 128, _p(ws["am"]),
                                    _p(P[ps + "conv1/weights"]), _p(P[ps + "conv1/biases"]), _p(ww), _p(bw), _p(ws["bn_scale"]),
                                    _p(ws["bn_shift"]), _p(ws["hp"]), self.st), "ps_local")
        # after_conv, [rows x 2048] x [2048 x 256].  At <= 16 patches the rows give 32 - 128 tiles of 128 x 256 -- a fraction of the chip --
        # and the tile rule falls back to 512 L2-bound 64 x 64 tiles (139 us in the 8-patch step for a 55 us product): the contraction is
        # split into K-chunks run as one batched launch that fills the chip with 128 x 256 tiles, a second launch adds the chunks in
        # order, then bias and ReLU (reassociated; `fine` is tolerance-checked).  fp32 products only (bf16 products have their own tiles)
        nsp = 4 if rm >= 8192 else 8
        if rm <= 16384 and rm % 128 == 0 and ws["hp"].dtype == torch.float32 and not self._use_bf16(1, rm, 2048, 256):
            if ws.get("aft_parts") is None or ws["aft_parts"].numel() < nsp * rm * 256:
                ws["aft_parts"] = torch.empty((nsp * rm, 256), dtype=torch.float32, device=self.device)
            wa, ba = P[ps + "after_conv/weights"], P[ps + "after_conv/biases"]
            kc = 2048 // nsp
            _lib.check(L.dispu_linear(nsp, rm, kc, 256, _p(ws["hp"]), 2048, kc, _p(wa), 256, kc * 256, 0, None, 0, _p(ws["aft_parts"]), 256, rm * 256,
                                      None, 0, 0, None, 0, 0, self.st), "dispu_linear[split-K]")
            _lib.check(L.dispu_linear_splitk_finish(rm, 256, nsp, _p(ws["aft_parts"]), rm * 256, _p(ba), 1, _p(ws["aft"]), 256, self.st),
                       "dispu_linear_splitk_finish")
        else:
            self._lin(ws["hp"], 0, 2048, ps + "after_conv", 1, ws["aft"], 0, 256)
        self._merge(0)
        self._merge(1)
        _lib.check(L.dispu_add3(rm * 256, _p(ws["aft"]), _p(ws["skip"]), _p(ws["nl"]), _p(ws["sum"]), self.st), "add3")
        # aggregation -> fine regressor (fc_layer0, fc_layer1, fc_layer2) -> coarse + sigmoid(.) - 0.5 in one launch
        fs = "refine/fine_coordinate_regressor/"
        ag = ps + "aggregation"
        _lib.check(L.dispu_mlp_chain_stash(rm, 256, 256, 256, 64, _p(ws["sum"]), 256, _p(P[ag + "/weights"]), _p(P[ag + "/biases"]),
                                           _p(P[fs + "fc_layer0/weights"]), _p(P[fs + "fc_layer0/biases"]),
                                           _p(P[fs + "fc_layer1/weights"]), _p(P[fs + "fc_layer1/biases"]),
                                           _p(P[fs + "fc_layer2/weights"]), _p(P[fs + "fc_layer2/biases"]),
                                           _p(ws["agg"]), 256, _p(ws["f256"]), 256, _p(ws["f64"]), 64, _p(ws["z"]), 3, 1, _p(coarse), 3,
                                           _p(ws["fine"]), 3, self.st), "mlp_chain[fine]")
        self._stash_ready = False
        self._fresh = True               # the atomics accumulators (dprep, dup128s) were zero-filled for exactly ONE backward()
        return ws["coarse"], ws["fine"]

    def _recompute_pair_tensors(self):
        """h0 = relu(G[j] - A[i]), h1 = relu(h0.W1 + b1), wv = relu(BN(offsets.Ww + bw)) [B*M*16, .] and the inverted k-NN graph: what
        the local cell's backward reads.  The forward kernel (csrc/ps_local.hip) keeps them on chip; they are rebuilt here, on an
        auxiliary stream next to the loss and the fine head's backward.  dtype "f32": the same fmaf chains as the forward kernel,
        hence the same values and the same ReLU decisions.  dtype "bf16": conv1 is rebuilt with bf16 PRODUCTS (and h0 / h1 are stored as
        bf16) while the forward's fused cell ran fp32 products -- the masks (h1 > 0) and the h1 values entering dwv are those of the
        bf16 rebuild, i.e. they can differ from the forward's at pre-activations within bf16 rounding of zero; that is part of the
        mixed-precision approximation and sits inside the tolerances of tests/test_train_bf16_gpu.py (loss terms 2 %, gradient cosine
        >= 0.97), not a separate error source."""
        if self._stash_ready:
            return
        L = _lib.tape_lib()
        B, N = self._shape
        M, k = N * self.up_ratio, K_NEIGH
        rm = B * M
        ws = self._workspace(B, N)
        P = self.P
        ps = "refine/PointShuffle/"
        coarse = ws["coarse"].view(rm, 3)
        fused = self._fused_local_bwd_ok(ws)
        if not fused:
            _lib.check(L.dispu_knn_invert(B, M, k, _p(ws["psidx"]), _p(ws["inv_off"]), _p(ws["inv"]), self.st), "knn_invert")
        if ws["h0"].dtype == torch.bfloat16:
            _lib.check(L.dispu_ps_gather_sub_relu_bf16(rm, M, k, 128, _p(ws["psidx"]), _p(ws["gm"]), 128, _p(ws["am"]), 128, _p(ws["h0"]), 128,
                                                       self.st), "gather_sub_relu_bf16")
        else:
            _lib.check(L.dispu_ps_gather_sub_relu(rm, M, k, 128, _p(ws["psidx"]), _p(ws["gm"]), 128, _p(ws["am"]), 128, _p(ws["h0"]), 128, self.st),
                       "gather_sub_relu")
        if not fused:                                    # (fused backward: only h0 is needed in HBM -- the dW1 = h0^T . dz1 product reads it)
            self._lin(ws["h0"], 0, 128, ps + "conv1", 1, ws["h1"], 0, 128)
            _lib.check(L.dispu_ps_weight_net(rm, M, k, 16, _p(ws["psidx"]), _p(coarse), _p(P[ps + "weight_net/wconv0/weights"]),
                                             _p(P[ps + "weight_net/wconv0/biases"]), _p(ws["bn_scale"]), _p(ws["bn_shift"]), _p(ws["wv"]), self.st),
                       "weight_net")
        self._stash_ready = True

    def _fused_local_bwd_ok(self, ws):
        """the one-launch backward of the local cell: fp32 pair-tensor storage, whole 8-point groups, transposed weight copies at hand."""
        return (self.fused_local_bwd and ws["h0"].dtype == torch.float32 and ws["h0"].shape[0] % 64 == 0 and self.use_wt
                and ("refine/PointShuffle/conv1/weights" in self.PT))

    # -------------------------------------------------------------------------------------------------- loss ----
    def _chamfer(self, pred, gt, radius, coef, dpred, slot):
        """one Chamfer term (loss_utils.py:45-64 with nn_distance(gt, pred)): its un-scaled value into loss_vals[slot] and
        d(coef * CD)/d pred into dpred -- three launches (nn_distance, value, gradient) + one memset."""
        L = _lib.tape_lib()
        full = self._workspace(*self._shape)
        ws = full["cd"][slot]
        B, n_gt, n_pred = gt.shape[0], gt.shape[1], pred.shape[1]
        _lib.check(L.dispu_nn_distance(B, n_gt, _p(gt), n_pred, _p(pred), _p(ws["d_gt"]), _p(ws["i_gt"]), _p(ws["d_pred"]),
                                       _p(ws["i_pred"]), _lib.ARITH_CONTRACT, self.st), "nn_distance")
        _lib.check(L.dispu_chamfer_loss_grad(B, n_gt, _p(gt), n_pred, _p(pred), _p(ws["d_gt"]), _p(ws["i_gt"]), _p(ws["d_pred"]),
                                             _p(ws["i_pred"]), _p(radius), coef, _p(full["loss_vals"], slot), _p(dpred), self.st),
                   "chamfer_loss_grad")

    def _check_targets(self, gt, radius, B, M):
        """gt [B, 4N, 3] / radius [B] float32 on the device: the loss workspace (d_gt, i_gt, g_gt, ...) is sized [B, 4N]
        and the kernels take raw pointers, so anything else must be refused here (DisPU/model.py:47-49 placeholders)."""
        gt, radius = f32(gt, "gt"), f32(radius, "radius")
        req(gt.dim() == 3 and tuple(gt.shape) == (B, M, 3),
            "gt must have shape (%d, %d, 3) for this batch, got %s" % (B, M, tuple(gt.shape)))
        req(radius.dim() == 1 and radius.shape[0] == B, "radius must have shape (%d,), got %s" % (B, tuple(radius.shape)))
        req(gt.device == self.device and radius.device == self.device, "gt / radius must live on %s" % self.device)
        return gt, radius

    def loss_backward(self, gt, radius):
        """pu_loss of model.py:75-87 at the current epoch; fills dcoarse / dfine with its gradient."""
        L = _lib.tape_lib()
        B, N = self._shape
        M = N * self.up_ratio
        ws = self._workspace(B, N)
        gt, radius = self._check_targets(gt, radius, B, M)
        wf = weight_fine(self.epoch)
        # side work is submitted behind the fine term's launches (the chain): the pair tensors of the local cell's backward (needed much
        # later) and the coarse term
        self._defer_branch(2, self._recompute_pair_tensors)    # off the chain: needed by the local cell's backward only
        with self._branch(0):                                   # the coarse term next to the fine term and the repulsion term
            self._chamfer(ws["coarse"], gt, radius, 1000.0, ws["dcoarse"], 0)
        # fine term: nn_distance, then value + gradient (which zero-fills dfine); the repulsion term's ball query runs next to it and
        # adds its gradient once the Chamfer gradient is in place
        rep = None
        if self.opts.use_repulse:
            fine = ws["fine"]
            with self._branch(1):
                _lib.check(L.dispu_query_ball(B, M, M, _p(ws["r07"]), 20, _p(fine), _p(fine), _p(ws["ball"]), _p(ws["ball_cnt"]),
                                              _lib.ARITH_CONTRACT, self.st), "query_ball")   # as loss_utils.get_repulsion_loss
        self._chamfer(ws["fine"], gt, radius, 1000.0 * wf, ws["dfine"], 1)
        if self.opts.use_repulse:
            self._merge(1)
            _lib.check(L.dispu_repulsion_loss_grad(B * M, M, 20, 0.001, self.opts.repulsion_w / (B * M * 4.0), _p(fine), _p(ws["ball"]),
                                                   _p(ws["rep"]), _p(ws["dfine"]), self.st), "repulsion_loss_grad")
            rep = ws["rep"]
        self._merge(0)
        out = ws["loss_vals"]
        _lib.check(L.dispu_pu_loss_finalize(_p(out), _p(rep) if rep is not None else None, B * M, wf, float(self.opts.repulsion_w),
                                            _p(out, 2), self.st), "pu_loss_finalize")
        terms = self._terms(out, wf)
        self._flush(prio=0)                                     # the recompute branch: submitted behind the loss's own launches
        return terms

    @staticmethod
    def _terms(out, wf):
        vals = out[2:6].clone()                                 # device scalars that survive the next step
        return {"dis_coarse_cd": vals[0], "dis_fine_cd": vals[1], "repulsion_loss": vals[2], "pu_loss": vals[3], "weight_fine": wf}

    # ---------------------------------------------------------------------------------------------- backward ----
    def backward(self):
        """gradients of pu_loss w.r.t. every trainable variable, accumulated into the flat gradient buffer.
        ReLU gradients never run as separate passes: the dX product of a layer applies the mask of the layer below in its epilogue
        (`mask=`), so every dY arriving at _lin_bwd is already the pre-activation gradient dZ."""
        L = _lib.tape_lib()
        B, N = self._shape
        M, k = N * self.up_ratio, K_NEIGH
        rn, rm = B * N, B * M
        ws = self._workspace(B, N)
        P, G = self.P, self.G
        coarse = ws["coarse"].view(rm, 3)
        dcoarse, dfine = ws["dcoarse"].view(rm, 3), ws["dfine"].view(rm, 3)
        ps = "refine/PointShuffle/"
        fs = "refine/fine_coordinate_regressor/"
        if self.overlap_dw and getattr(self, "_prep_side", None) is not None:
            # the accumulators zeroed and the W^T copies made on a weight-gradient stream during the forward (see forward())
            i, self._prep_side = self._prep_side, None
            ev = self._ev(self._join_evs[i])
            self._rec(ev, self._sides[i])
            self._wait(torch.cuda.current_stream(self.device), ev)
        if not getattr(self, "_fresh", False):
            # a second backward() on the same forward (new targets / loss weights): the atomics accumulators still hold the previous,
            # already masked gradients -- clear them again, on this stream, before anything accumulates
            self._zero(ws["zeroed"])
        self._fresh = False
        if not self._stash_ready:                       # backward() without loss_backward(): rebuild the pair tensors here
            with self._branch(2):
                self._recompute_pair_tensors()

        if self._sched & 2:
            self._merge(2)                               # h0 / h1 / wv / the inverted graph: done by the time the loss is (see _branch)
        # fine = coarse + sigmoid(z) - 0.5
        _lib.check(L.dispu_sigmoid_offset_grad(rm * 3, _p(ws["z"]), _p(dfine), _p(ws["dz"]), _p(dcoarse), self.st), "sigmoid_grad")
        ag = ps + "aggregation"
        if self.fused_heads_bwd:
            # fc_layer2 -> fc_layer1 -> fc_layer0 -> aggregation backward and the three branch masks of
            # sum = relu(after) + relu(skip) + relu(nl) in ONE launch (csrc/mlp_chain_bwd.hip); the four dW products follow on the side streams
            PT = self.PT
            _lib.check(L.dispu_mlp_chain_grad(rm, 256, 256, 256, _p(ws["dz"]), 3, _p(P[fs + "fc_layer2/weights"]),
                                              _p(PT[fs + "fc_layer1/weights"]), _p(PT[fs + "fc_layer0/weights"]), _p(PT[ag + "/weights"]),
                                              _p(ws["f64"]), 64, _p(ws["f256"]), 256, _p(ws["agg"]), 256, None, 0, None, 0,
                                              _p(ws["df64"]), 64, _p(ws["df256"]), 256, _p(ws["dagg"]), 256,
                                              _p(ws["aft"]), _p(ws["skip"]), _p(ws["nl"]), 256, _p(ws["daft"]), _p(ws["dskip"]), _p(ws["dnl"]), 256,
                                              self.st), "mlp_chain_grad[fine]")
            with self._fork_group():
                self._lin_bwd(ws["f64"], 0, 64, fs + "fc_layer2", 3, ws["dz"], 0)
                self._lin_bwd(ws["f256"], 0, 256, fs + "fc_layer1", 64, ws["df64"], 0)
                self._lin_bwd(ws["agg"], 0, 256, fs + "fc_layer0", 256, ws["df256"], 0)
                self._lin_bwd(ws["sum"], 0, 256, ag, 256, ws["dagg"], 0)
        else:
            self._lin_bwd(ws["f64"], 0, 64, fs + "fc_layer2", 3, ws["dz"], 0, ws["df64"], mask=(ws["f64"], 0, 64))
            self._lin_bwd(ws["f256"], 0, 256, fs + "fc_layer1", 64, ws["df64"], 0, ws["df256"], mask=(ws["f256"], 0, 256))
            self._lin_bwd(ws["agg"], 0, 256, fs + "fc_layer0", 256, ws["df256"], 0, ws["dagg"], mask=(ws["agg"], 0, 256))
            self._lin_bwd(ws["sum"], 0, 256, ag, 256, ws["dagg"], 0, ws["dsum"])
            # sum = relu(after) + relu(skip) + relu(nl): the three branch gradients in one pass
            _lib.check(L.dispu_mask3(rm, 256, _p(ws["dsum"]), 256, _p(ws["aft"]), 256, _p(ws["skip"]), 256, _p(ws["nl"]), 256, _p(ws["daft"]),
                                     _p(ws["dskip"]), _p(ws["dnl"]), 256, self.st), "mask3")

        # local cell first: the host needs ~0.1 ms to queue the two branches below, the chain must not sit idle meanwhile; the
        # branches themselves only need the mask3 outputs, so they are ordered after THIS point of the stream, not after the product
        ev_br = self._fork_point() if self.overlap_dw else None
        self._lin_bwd(ws["hp"], 0, 2048, ps + "after_conv", 256, ws["daft"], 0, ws["dhp"])
        # non-local cell: reads dnl, writes datt / dS / dkv / dq / dup128 -- nothing the local cell or the skip branch touches, so
        # it runs as a branch next to them; merged before anything else accumulates into dup128
        dup128 = ws["dup128"]
        def nl_backward():
            self._lin_bwd(ws["att"], 0, 64, ps + "PointShuffle/conv_back_project", 256, ws["dnl"], 0, ws["datt"])
            S, dS, kv, dkv, q = ws["S"], ws["dS"], ws["kv"], ws["dkv"], ws["q"]
            if self.flash_attn:
                # dQ, dK, dV with the probabilities recomputed tile by tile from Q, K and the forward's log-sum-exp: two launches
                _lib.check(L.dispu_attention_bwd(B, M, M, 64, _p(q), 64, _p(kv), 128, _p(kv, 64), 128, 0.125, _p(ws["att"]), 64,
                                                 _p(ws["lse"]), _p(ws["datt"]), 64, _p(ws["dq"]), 64, _p(dkv), 128, _p(dkv, 64), 128,
                                                 _p(ws["dvec"]), self.st), "attention_bwd")
            else:
                # dP = dO . V^T
                _lib.check(self._dl(B, M, 64, M, _p(ws["datt"]), 64, M * 64, _p(kv, 64), 128, M * 128, 1, None, 0, _p(dS), M, M * M,
                                          None, 0, 0, None, 0, 0, self.st), "dP")
                # dV = P^T . dO  -> dkv[:, 64:128]
                self._tn(B, M, M, 64, S, 0, M, M * M, ws["datt"], 0, 64, M * 64, dkv, 64, 128, M * 128, 0)
                _lib.check(L.dispu_softmax_rows_grad(rm, M, 0.125, _p(S), M, _p(dS), M, self.st), "softmax_grad")
                # dQ = dS . K
                _lib.check(self._dl(B, M, M, 64, _p(dS), M, M * M, _p(kv), 128, M * 128, 0, None, 0, _p(ws["dq"]), 64, M * 64,
                                          None, 0, 0, None, 0, 0, self.st), "dQ")
                # dK = dS^T . Q -> dkv[:, 0:64]
                self._tn(B, M, M, 64, dS, 0, M, M * M, q, 0, 64, M * 64, dkv, 0, 128, M * 128, 0)
            self._lin_bwd(ws["up128"], 0, 128, ps + "PointShuffle/conv_kv", 128, dkv, 0, dup128)
            self._lin_bwd(ws["up128"], 0, 128, ps + "PointShuffle/conv_query", 64, ws["dq"], 0, dup128, 0, acc_dx=True)
        self._defer_branch(0, nl_backward, ev_br)
        # skip branch (a second branch): 134 -> 256 backward; its max gradient is scattered after the merges below
        split_skip = self.fused_heads_bwd and self.overlap_dw      # the skip branch's d(up128) in its own buffer, summed in the coarse chain
        def skip_backward():
            self._lin_bwd(ws["gmax"], 0, 134, ps + "skip", 256, ws["dskip"], 0, ws["dgmax"])
            if split_skip:
                _lib.check(L.dispu_ps_skip_max_grad(rm, M, k, 128, _p(ws["psidx"]), _p(coarse), _p(ws["up128"]), 128, _p(ws["gmax"]), 144,
                                                    _p(ws["dgmax"]), 136, _p(dcoarse), _p(ws["dup128s"]), 128, 1, self.st), "ps_skip_max_grad")
        self._defer_branch(1, skip_backward, ev_br)
        # after_conv's dX (0.1 - 0.15 ms on the GPU) is queued: submit the two branches the chain will wait for behind it; the weight
        # gradients stay deferred until the chain's next three kernels are queued too
        self._flush(prio=0)
        if not (self._sched & 2):
            self._merge(2)                               # h0 / h1 / wv / the inverted graph are in place
        fused_lb = self._fused_local_bwd_ok(ws)
        if fused_lb:
            # one launch: dwv, dz1 (for the side-stream dW1), dG (atomics into the zeroed buffer) and -dA; h1 / dz0 / wv stay on chip
            self._zero(ws["dG"])
            _lib.check(L.dispu_ps_local_grad(rm, M, _p(ws["psidx"]), _p(coarse), _p(ws["gm"]), 128, _p(ws["am"]), _p(P[ps + "conv1/weights"]),
                                             _p(P[ps + "conv1/biases"]), _p(self.PT[ps + "conv1/weights"]), _p(P[ps + "weight_net/wconv0/weights"]),
                                             _p(P[ps + "weight_net/wconv0/biases"]), _p(ws["bn_scale"]), _p(ws["bn_shift"]), _p(ws["dhp"]),
                                             _p(ws["dz1"]), _p(ws["dwv"]), _p(ws["dG"]), _p(ws["dAneg"]), self.st), "ps_local_grad")
        else:
            _lib.check(L.dispu_ps_point_matmul_grad_relu_s(rm, k, 128, 16, _p(ws["h1"]), 128, _p(ws["wv"]), _p(ws["dhp"]), 2048, _p(ws["dz1"]),
                                                           128, _p(ws["dwv"]), 1 if ws["h1"].dtype == torch.bfloat16 else 0, self.st),
                       "point_matmul_grad")
        # the weight net's backward (dwv -> BatchNorm -> 3 -> 16 conv -> atomics into dcoarse) next to the conv1 / conv0 gradients
        def wnet_backward():
            ww, bw = P[ps + "weight_net/wconv0/weights"], P[ps + "weight_net/wconv0/biases"]
            _lib.check(L.dispu_ps_wnet_grad(rm, M, k, 16, _p(ws["psidx"]), _p(coarse), _p(ww), _p(bw), _p(ws["bn_stats"]), _p(ws["bn_scale"]),
                                            _p(ws["bn_shift"]), _p(P[BN + "gamma"]), _p(ws["dwv"]), _p(G[ps + "weight_net/wconv0/weights"]),
                                            _p(G[ps + "weight_net/wconv0/biases"]), _p(G[BN + "gamma"]), _p(G[BN + "beta"]), _p(dcoarse),
                                            _p(ws["bn_sums"]), _p(self._bn_scratch), self._bn_scratch.numel() * 8, self.st), "ps_wnet_grad")
        self._defer_branch(2, wnet_backward)
        if fused_lb:
            self._lin_bwd(ws["h0"], 0, 128, ps + "conv1", 128, ws["dz1"], 0, None)            # dW1, db1 only: the dX product ran inside the fused launch
        else:
            self._lin_bwd(ws["h0"], 0, 128, ps + "conv1", 128, ws["dz1"], 0, ws["dz0"])      # dz0 holds dh0: conv0's relu' rides in the gather
            # conv0 in its per-source-point form: dh0 * (G[j] - A[i] > 0) -> dG (gather through the inverted graph), -dA; then [B*M, 128] products
            _lib.check(L.dispu_ps_conv0_gather_grad_s(rm, M, k, 128, _p(ws["psidx"]), _p(ws["inv_off"]), _p(ws["inv"]), _p(ws["dz0"]), 128,
                                                      1 if ws["dz0"].dtype == torch.bfloat16 else 0, _p(ws["gm"]), 128, _p(ws["am"]), 128,
                                                      _p(ws["dG"]), 128, _p(ws["dAneg"]), 128, self.st), "conv0_gather_grad")
        # the main queue now holds after_conv's dX, the feature x weight gradient, conv1's dX and the gather (~0.35 ms of kernels at 8
        # patches): time to submit the side work that piled up behind them (five weight gradients, the non-local and skip branches)
        self._flush()
        w0, dw0 = P[ps + "conv0/weights"], G[ps + "conv0/weights"]
        self._merge(0)                                   # dup128 holds the non-local cell's part from here on
        self._lin_bwd(ws["up128"], 0, 128, None, 128, ws["dG"], 0, dup128, 0, acc_dx=True, W=w0, dW=dw0, woff=6 * 128, bias=False,
                      db=G[ps + "conv0/biases"])
        _lib.check(L.dispu_ps_prep_grad(rm, 128, _p(coarse), _p(w0), _p(ws["dG"]), 128, _p(ws["dAneg"]), 128, _p(dcoarse), _p(dw0), self.st),
                   "ps_prep_grad")
        self._merge(1)
        if not split_skip:
            _lib.check(L.dispu_ps_skip_max_grad(rm, M, k, 128, _p(ws["psidx"]), _p(coarse), _p(ws["up128"]), 128, _p(ws["gmax"]), 144,
                                                _p(ws["dgmax"]), 136, _p(dcoarse), _p(dup128), 128, 1, self.st), "ps_skip_max_grad")
        self._merge(2)                                   # the weight net's share of dcoarse
        # the refine branch's weight-gradient products submitted so far (after_conv's 2048 x 256 with its ~100 MB of partials among them)
        # get their grouped reduction HERE, in the shadow of the coarse chain's backward; whatever is left at _join() is small.  (All of
        # them at _join(): 35 us of reductions between the last product and Adam.)
        self._rg_flush()
        self._bucket_point(0)                            # data parallel: every refine/* gradient is queued -> its all-reduce starts

        # coarse regressor
        cs = "generator/coarse_coordinate_regressor/"
        c2 = "generator/upshuffle_0/conv2"
        if self.fused_heads_bwd:
            # fc_layer2 -> fc_layer1 -> fc_layer0 (+ everything already accumulated in dup128, then conv2's relu') -> conv2 (duplicate_up's
            # relu') in one launch
            PT = self.PT
            _lib.check(L.dispu_mlp_chain_grad(rm, 256, 128, 256, _p(dcoarse), 3, _p(P[cs + "fc_layer2/weights"]),
                                              _p(PT[cs + "fc_layer1/weights"]), _p(PT[cs + "fc_layer0/weights"]), _p(PT[c2 + "/weights"]),
                                              _p(ws["c64"]), 64, _p(ws["c256"]), 256, _p(ws["up128"]), 128, _p(dup128), 128,
                                              _p(ws["dup128s"]) if split_skip else None, 128, _p(ws["dc64"]), 64, _p(ws["dc256"]), 256, _p(dup128), 128,
                                              _p(ws["up256"]), None, None, 256, _p(ws["dup256"]), None, None, 256, self.st),
                       "mlp_chain_grad[coarse]")
            # the chain goes on first (d(up256) summed over the four copies -> the 480-wide product below); the head's four weight
            # gradients are queued on the side streams after it
            _lib.check(L.dispu_dup_sum_grad(B, N, 256, self.up_ratio, _p(ws["dup256"]), 256, _p(ws["dh256"]), 256, self.st), "dup_sum_grad")
            with self._fork_group():
                self._lin_bwd(ws["c64"], 0, 64, cs + "fc_layer2", 3, dcoarse, 0)
                self._lin_bwd(ws["c256"], 0, 256, cs + "fc_layer1", 64, ws["dc64"], 0)
                self._lin_bwd(ws["up128"], 0, 128, cs + "fc_layer0", 256, ws["dc256"], 0)
                self._lin_bwd(ws["up256"], 0, 256, c2, 128, dup128, 0)
                w1, dw1 = P["generator/upshuffle_0/conv1/weights"], G["generator/upshuffle_0/conv1/weights"]
                self._tn(1, rm, 2, 256, ws["gcode"], 0, 2, 0, ws["dup256"], 0, 256, 0, dw1, 480 * 256, 256, 0, 1,
                         dbias=G["generator/upshuffle_0/conv1/biases"], side=True, wgrad=True)   # read by Adam only: off the chain like every other dW
        else:
            self._lin_bwd(ws["c64"], 0, 64, cs + "fc_layer2", 3, dcoarse, 0, ws["dc64"], mask=(ws["c64"], 0, 64))
            self._lin_bwd(ws["c256"], 0, 256, cs + "fc_layer1", 64, ws["dc64"], 0, ws["dc256"], mask=(ws["c256"], 0, 256))
            # the last product that accumulates into dup128 applies conv2's relu' (up128 = relu(conv2))
            self._lin_bwd(ws["up128"], 0, 128, cs + "fc_layer0", 256, ws["dc256"], 0, dup128, 0, acc_dx=True, mask=(ws["up128"], 0, 128))
            # duplicate_up
            self._lin_bwd(ws["up256"], 0, 256, c2, 128, dup128, 0, ws["dup256"], mask=(ws["up256"], 0, 256))
        w1, dw1 = P["generator/upshuffle_0/conv1/weights"], G["generator/upshuffle_0/conv1/weights"]
        if not self.fused_heads_bwd:
            self._tn(1, rm, 2, 256, ws["gcode"], 0, 2, 0, ws["dup256"], 0, 256, 0, dw1, 480 * 256, 256, 0, 1,
                     dbias=G["generator/upshuffle_0/conv1/biases"], side=True)     # read by Adam only: off the chain like every other dW
            _lib.check(L.dispu_dup_sum_grad(B, N, 256, self.up_ratio, _p(ws["dup256"]), 256, _p(ws["dh256"]), 256, self.st), "dup_sum_grad")
        feat, dfeat = ws["feat"], ws["dfeat"]
        self._lin_bwd(feat, 0, 480, None, 256, ws["dh256"], 0, dfeat, 0, bias=False, W=w1, dW=dw1)

        # dense blocks, last to first (every block has its own dE / dprep: the dW products on the second stream read them while
        # the chain moves on, no join inside the loop)
        fe = "generator/feature_extraction_coarse/"
        for (d, C, col, in_col, width) in reversed(self._blocks):
            sc = fe + "layer%d" % d
            if d == 1:
                dF, dfoff, F, foff = dfeat, 456, feat, 456
            else:
                dF, dfoff, F, foff = ws["dprep"][d], 0, ws["prep"][d], 0         # zero-filled in forward(), off the chain
            if self.fused_dense:
                # the block's backward on the chain; its weight-gradient partials (a scratch buffer per block) are summed on a side stream
                need = L.dispu_edge_dense_conv_grad_scratch_floats(rn, C)
                scr = self._scratch_floats(need, "edge%d" % d)
                _lib.check(L.dispu_edge_dense_conv_grad_partials(rn, N, C, _p(F, foff), F.stride(0), _p(ws["kidx"][d]), k + 1, 1,
                                                                 _p(P[sc + "/l0/weights"]), _p(P[sc + "/l0/biases"]), _p(P[sc + "/l1/weights"]),
                                                                 _p(P[sc + "/l1/biases"]), _p(P[sc + "/l2/weights"]), _p(P[sc + "/l2/biases"]),
                                                                 _p(dfeat, col), 480, _p(dF, dfoff), dF.stride(0), _p(scr), scr.numel(), self.st),
                           "edge_dense_conv_grad_partials")
                # the block's recomputing backward kernel (40 - 80 us) is queued: submit some of the side work deferred so far behind it
                # (the coarse head's weight gradients, the previous blocks' reductions and prep gradients) -- a few launches per block,
                # as many as the kernel's duration hides
                tail = self.tail_on_chain and d == 1 and self.overlap_dw
                self._flush(n=None if tail else 5)
                def reduce_partials(sc=sc, scr=scr, C=C, ev=(self._fork_point() if (self.overlap_dw and not tail) else None)):
                    st_r = self._fork_after(ev)[0] if ev is not None else self.st
                    _lib.check(L.dispu_edge_dense_conv_grad_reduce(rn, C, _p(scr), scr.numel(), _p(G[sc + "/l0/weights"]), _p(G[sc + "/l0/biases"]),
                                                                   _p(G[sc + "/l1/weights"]), _p(G[sc + "/l1/biases"]), _p(G[sc + "/l2/weights"]),
                                                                   _p(G[sc + "/l2/biases"]), st_r), "edge_dense_conv_grad_reduce")
                if tail:
                    # the last block of the backward: everything deferred so far is on the side streams by now; this block's own
                    # reduction (and layer0's weight gradient below) follow their producer on the chain's stream -- Adam waits for
                    # them either way, and a hop to a side stream and back costs more than the two launches take
                    reduce_partials()
                else:
                    self._defer(reduce_partials)
            else:
                Eb, dE = self._edge_buffers(B, N)[0][d], self._edge_buffers(B, N)[1][d]
                lde = dE.stride(0)
                # max gradient into the pooled columns [0, width), zeros into the neighbour half of the edge feature behind them
                _lib.check(L.dispu_max_k_grad_tail(rn, k, width, C, _p(Eb), Eb.stride(0), _p(feat, col), 480, _p(dfeat, col), 480, _p(dE), lde, self.st),
                           "max_k_grad")
                self._lin_bwd(Eb, 24, 48 + C, sc + "/l2", 24, dE, 0, dE, 24, acc_dx=True, mask=(Eb, 24, 24))     # columns 24:48 = l1: relu'
                self._lin_bwd(Eb, 48, 24 + C, sc + "/l1", 24, dE, 24, dE, 48, acc_dx=True, mask=(Eb, 48, 24))    # columns 48:72 = l0: relu'
                self._lin_bwd(Eb, 72, 2 * C, sc + "/l0", 24, dE, 48, dE, 72, acc_dx=True)
                _lib.check(L.dispu_edge_feature_grad(rn, N, k, C, _p(dE, 72), lde, _p(ws["kidx"][d]), k + 1, 1, _p(dF, dfoff), dF.stride(0), self.st),
                           "edge_feature_grad")
            if d > 1:
                self._act_bias_grad(rn, 48, dF, 0, ws["prep"][d], 0, 1, dF, 0, None)        # prep = relu(.): its mask (dF came from atomics)
                self._lin_bwd(feat, in_col, 480 - in_col, fe + "layer%d_prep" % d, 48, dF, 0, dfeat, in_col, acc_dx=True)
        # layer0 (no activation, input has no gradient)
        self._lin_bwd(self._x.view(rn, 3), 0, 3, fe + "layer0", 24, dfeat, 456, None, side=not (self.tail_on_chain and self.fused_dense))
        self._join()                     # every dW is in the flat gradient buffer from here on (all-reduce, Adam)

    # -------------------------------------------------------------------------------------------------- step ----
    def _backward_prep(self, ws):
        L = _lib.tape_lib()
        self._zero(ws["zeroed"])          # the dense blocks' input gradients, the skip branch's d(up128): accumulated with atomics
        if (self.use_wt or self.fused_heads_bwd) and self._t_desc.numel():       # W^T copies for the dX products
            _lib.check(L.dispu_transpose_batched(self._t_desc.numel() // 3, _p(self._t_desc), _p(self.flat_p), _p(self.flat_pT), self.st),
                       "transpose_batched")

    def zero_grad(self):
        self.st = _lib.stream_ptr(self.device)
        self._zero(self.flat_g)

    def _reducer(self):
        """data parallel (process group of > 1 rank): the flat gradient buffer as TWO all-reduce buckets in the order the backward
        pass completes them -- [refine/*] (3.1 MB: final once the local / non-local / skip cells are through, half a backward
        before the rest) and [generator/*] (1.0 MB: feature extractor, duplicate_up, coarse regressor) -- on a comm lane next to
        the compute streams (parallel.BucketedAllReduce).  None without a process group."""
        if self._ar is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.pg) == 1 and not self.collectives_at_world_1):
                return None
            from . import parallel
            first = next(k for k in self.names if k.startswith("refine/"))
            split = (self.P[first].data_ptr() - self.flat_p.data_ptr()) // 4
            assert all(k.startswith("refine/") == ((self.P[k].data_ptr() - self.flat_p.data_ptr()) // 4 >= split) for k in self.names)
            self._ar = parallel.BucketedAllReduce(self._red, [(split, self._red.numel()), (0, split)], self.pg, threaded=self.comm_thread)
        return self._ar

    def _bucket_point(self, i):
        """every gradient of bucket i has been QUEUED (main stream, or a weight-gradient stream): start its all-reduce behind them,
        while the backward pass goes on.  Eager steps only: a launch tape / hipGraph replays kernels, not collectives -- there every
        bucket is launched by all_reduce_grads() after the replay."""
        if not self._ar_armed:                           # backward() on its own never communicates: only train_step() arms the early launch
            return
        ar = self._reducer()
        if ar is None or _lib.taping() is not None or torch.cuda.is_current_stream_capturing() or ar.launched(i):
            return
        self._flush()                                    # deferred weight gradients of the bucket go to their streams first
        self._rg_flush()                                 # ... and their pending split reductions
        evs = []
        for s in [torch.cuda.current_stream(self.device)] + list(self._sides):
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        ar.launch(i, after=evs)

    def all_reduce_grads(self):
        """gradient all-reduce (RCCL over xGMI) of the 4.2 MB buffer, in the buckets of _reducer(): whatever backward() has not
        launched yet is launched here, then the current stream waits for all of it.  The 1/world average is folded into the Adam
        launch.  BN moving statistics (per-rank batch statistics) ride at the tail of the refine bucket and are averaged here."""
        ar = self._reducer()
        if ar is None:
            return 1
        world = ar.finish()
        if world > 1:
            self._stats.mul_(1.0 / world)                # summed with the refine bucket; the gradients' 1/world rides in the Adam launch
        return world

    def adam(self, world=1):
        """tf.train.AdamOptimizer(lr, beta1=opts.beta) (model.py:178)."""
        self.adam_t += 1
        b1, b2 = float(self.opts.beta), 0.999
        lr = learning_rate(self.opts, self.epoch)
        lr_t = lr * math.sqrt(1.0 - b2 ** self.adam_t) / (1.0 - b1 ** self.adam_t)
        _lib.check(_lib.tape_lib().dispu_adam(self.flat_p.numel(), _p(self.flat_p), _p(self.flat_g), _p(self.flat_m), _p(self.flat_v),
                                         lr_t, b1, b2, 1e-8, 1.0 / world, _lib.stream_ptr(self.device)), "dispu_adam")

    def train_step_taped(self, inputs, gt, radius):
        """train_step with forward + loss + backward re-issued from a LAUNCH TAPE (dis-pu_amd/_lib.py:Tape): the same launches on the
        same streams in the same order as the eager step, recorded once per (B, N, loss weights) with their ctypes arguments already
        converted, then replayed as a flat loop of foreign calls.  The eager step spends ~10 us of Python per launch (1.4 ms for ~130
        launches at 8 patches -- as long as the GPU's critical chain, so the main queue idles wherever a chain kernel is submitted
        behind side work); the replay spends ~1.5 us and the GPU-side schedule is the eager one (a hipGraph replay of the same step ran
        its branches nearly one after the other on this runtime, 11 - 15 % slower than eager: removed in round 5).  Inputs are
        copied into static buffers (the tape holds raw pointers); the all-reduce and Adam stay eager."""
        B, N = inputs.shape[0], inputs.shape[1]
        gt, radius = self._check_targets(gt, radius, B, N * self.up_ratio)
        wf = weight_fine(self.epoch)
        key = (B, N, wf, self.opts.use_repulse, torch.cuda.current_stream(self.device).cuda_stream)
        t = self._tapes.get(key)
        if t is None:
            st = dict(x=inputs.clone(), gt=gt.clone(), radius=radius.clone())
            mm, mv = self.moving_mean.clone(), self.moving_var.clone()
            for _ in range(2):                               # every workspace / scratch buffer exists and has its final size
                self.zero_grad()
                self.forward(st["x"])
                self.loss_backward(st["gt"], st["radius"])
                self.backward()
            self.moving_mean.copy_(mm)                       # the warm-up passes are not training steps
            self.moving_var.copy_(mv)
            torch.cuda.synchronize(self.device)
            _lib.tape_begin()
            try:
                self.zero_grad()
                self.forward(st["x"])
                self.loss_backward(st["gt"], st["radius"])
                self.backward()
            finally:
                st["tape"] = _lib.tape_end()
            self.moving_mean.copy_(mm)                       # ... and neither is the recording pass
            self.moving_var.copy_(mv)
            st["loss_vals"] = self._workspace(B, N)["loss_vals"]
            t = self._tapes[key] = st
        t["x"].copy_(inputs)
        t["gt"].copy_(gt)
        t["radius"].copy_(radius)
        t["tape"].replay()
        terms = self._terms(t["loss_vals"], wf)
        world = self.all_reduce_grads()
        self.adam(world)
        self.global_step += 1
        return terms

    def train_step(self, inputs, gt, radius):
        """one iteration of the loop body of Model.train (model.py:215-232) -> loss terms (device scalars)."""
        if isinstance(inputs, torch.Tensor) and inputs.dim() == 3:       # refuse bad targets before any launch
            self._check_targets(gt, radius, inputs.shape[0], inputs.shape[1] * self.up_ratio)
        self.zero_grad()
        self.forward(inputs)
        terms = self.loss_backward(gt, radius)
        self._ar_armed = True                            # data parallel: the refine bucket's all-reduce starts inside backward()
        try:
            self.backward()
        finally:
            self._ar_armed = False
        world = self.all_reduce_grads()
        self.adam(world)
        self.global_step += 1
        return terms
