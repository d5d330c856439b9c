"""Dis-PU generator forward pass (inference) on MI355X.

Counterpart of DisPU/generator.py:21-88 (`Generator(opts, is_training)(inputs) -> (coarse, fine)`), issuing the
same op sequence as the reference graph (Common/ops.py: feature_extraction_GCN :1437, dense_conv :1897,
duplicate_up :1152, coordinate_regressor :1089, PointShuffle2 :1012, PointNonLocalCell :302) through the C ABI
of libdispu_hip.so.  torch is used for device memory and the launch stream only.

What is different from the reference's execution (not its results):
  * no tf.concat / tf.tile / gather_nd tensors: dense-block outputs are written straight into one
    [rows, 480] feature buffer, prep convs read column slices of it;
  * the xyz k-NN of PointShuffle2 runs on the device (reference: tf.py_func -> nanoflann on the host);
  * duplicate_up's 482-wide conv is evaluated once per SOURCE point (same fmaf chain, 4x fewer FLOPs);
  * PointShuffle2's conv0 over [B,N,16,134] is evaluated per source point (linear in its input groups,
    16x fewer FLOPs; reassociated -> tolerance-checked).
Everything up to and including `coarse` is bit-identical to oracle/generator.py (fp32 MFMA == fmaf chain).
"""
import contextlib
import math

import numpy as np
import torch

from . import _lib

K_NEIGH = 16
GROWTH = 24
DENSE_BLOCKS = 4
BN_EPS = 1e-3


def gen_grid(up_ratio):
    """Common/ops.py:60-76."""
    sq = int(math.sqrt(up_ratio)) + 1
    for i in range(sq, 0, -1):
        if up_ratio % i == 0:
            num_x, num_y = i, up_ratio // i
            break
    gx = np.linspace(-0.2, 0.2, num_x, dtype=np.float32)
    gy = np.linspace(-0.2, 0.2, num_y, dtype=np.float32)
    x, y = np.meshgrid(gx, gy)
    return np.stack([x, y], -1).reshape(-1, 2).astype(np.float32)


class _Opts(object):
    patch_num_point = 256
    up_ratio = 4


class Generator(object):
    """Generator(opts, is_training=False)(inputs[B,N,3]) -> (coarse[B,4N,3], fine[B,4N,3]).

    `params` maps the TF variable names of the reference graph ('generator/feature_extraction_coarse/layer0/weights',
    ..., see oracle/generator.py:layer_shapes) to arrays: weights [C_in_total, C_out], biases [C_out], plus the
    four BatchNorm vectors of 'refine/PointShuffle/weight_net/wconv0/bn/'.

    Threading / streams: one Generator serves ONE caller at a time on the stream current at the call (like a session's run()).  The
    forward forks its non-local branch onto a private auxiliary stream and joins it before it returns; workspace buffers are reused
    by the next call with the same (B, N), and the results (return_views = False) are fresh tensors ordered on the caller's stream.
    Use one Generator per concurrent caller (two in flight: bench.py's alt_two_in_flight holds two)."""

    MAX_BATCH = 2048     # patches per launch sequence: keeps every row*stride product below 2^31 and the workspace
                         # (~17 MB per patch, dominated by F' [B*1024, 2048]) at ~35 GB; larger batches run in chunks

    def __init__(self, opts=None, is_training=False, name="Generator", params=None, device=None):
        self.opts = opts if opts is not None else _Opts()
        self.is_training = is_training
        self.name = name
        self.num_point = int(self.opts.patch_num_point)
        self.up_ratio = int(self.opts.up_ratio)
        if self.up_ratio != 4:
            raise NotImplementedError("the shipped generator graph is built for up_ratio 4 (DisPU/configs.py)")
        self.out_num_point = self.num_point * self.up_ratio
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.P = {}
        self._ws = {}
        self.profile = None          # set to [] to collect (name, start_event, end_event) per launch
        self.fused_local = True      # PointShuffle2 local cell in one kernel (False: the 4-kernel chain, for A/B tests)
        self.fused_attention = True  # non-local cell attention on chip (False: GEMM -> softmax -> GEMM through HBM)
        self.fused_project = True    # conv_back_project as the attention kernel's epilogue (False: separate GEMM)
        # EXPLORATORY: after_conv's products as 3-way split-bf16 MFMAs (fp32-accurate, not the fmaf chain; the branch is
        # tolerance-checked).  Off by default; bench.py --split-bf16 reports it beside the strict-fp32 line.
        self.split_bf16 = False
        self._planes = {}
        self.split_up3 = True        # the N = 320 product as 256 + 64 columns (each launch reads up128 once)
        # non-local cell on a second stream next to the grouping / skip / local cell (round 4: on by default, -1 % since the head chains
        # form their own inputs -- the branch now joins right before the fine chain; rounds 1 - 3 measured it +1 %)
        self.branches = True           # False: everything on the launch stream (profiling passes: every kernel alone on the device)
        self._aux = None
        self.fused_residual = True
        self.fused_heads = True      # one launch per head chain
        self.keep_intermediates = False   # fused heads: also write the aggregation output (tests compare it)
        # round 4: the head chains form their own input tiles (csrc/mlp_chain.hip XMODE): duplicate_up's rows from the per-source-point
        # product (no dup_grid launch, no [B*4N, 256] tensor) and relu(after_conv) + skip + non-local in the fine chain's loader (the
        # residual reads leave the after_conv GEMM's epilogue).  False = the producer kernels of rounds 1 - 3 (bit-identical results; A/B tests)
        self.chain_inputs = True
        # one launch per dense block (neighbour search + edge features + dense_conv, csrc/edge.hip KNN variant) for clouds of <= 256
        # points; False = the dispu_knn_feat_strided -> dispu_edge_dense_conv pair (A/B tests; larger clouds always take the pair)
        self.fused_stem = True
        # forward() computes into a reusable per-(B, N) workspace.  By default the two results are returned as fresh
        # tensors (like sess.run in the reference); return_views = True hands out the workspace buffers themselves
        # (no copies -- bench.py / hipGraph capture), which the NEXT call with the same (B, N) overwrites.
        self.return_views = False
        # optional caller-owned [B, 4N, 3] float32 buffer the fine head writes its clouds into instead of the workspace's (the
        # sharded serving loop alternates two of them so that the all-gather of one step overlaps the next step: parallel.GatherPipeline)
        self.fine_out = None
        self._trainer = None
        if params is not None:
            self.load_params(params)

    # ------------------------------------------------------------------------------------------ weights ----
    def load_params(self, params):
        dev = self.device
        if self.is_training:
            # training graph (DisPU/generator.py:22-31 with is_training=True; DisPU/model.py:68): BatchNorm on batch
            # statistics with moving averages updated (decay 0.95), every activation kept for the backward pass.  The
            # forward is Trainer.forward (train.py); `trainer` exposes loss / backward / Adam on the same variables.
            from .train import Trainer, TrainOpts
            topts = TrainOpts()
            for k in dir(self.opts):
                if not k.startswith("_"):
                    setattr(topts, k, getattr(self.opts, k))
            self._trainer = Trainer(opts=topts, params=params, device=dev)
            self.P = self._trainer.P
            return
        self.P = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev) for k, v in params.items()}
        self._planes = {}
        bn = "refine/PointShuffle/weight_net/wconv0/bn/"
        g, b = params[bn + "gamma"].astype(np.float64), params[bn + "beta"].astype(np.float64)
        mu, var = params[bn + "moving_mean"].astype(np.float64), params[bn + "moving_variance"].astype(np.float64)
        scale = g / np.sqrt(var + BN_EPS)
        self.bn_scale = torch.from_numpy(scale.astype(np.float32)).to(dev)
        self.bn_shift = torch.from_numpy((b - mu * scale).astype(np.float32)).to(dev)
        self.grid = torch.from_numpy(gen_grid(self.up_ratio)).to(dev)
        w1 = self.P["generator/upshuffle_0/conv1/weights"]
        self.w_up_feat = w1[:480].contiguous()
        self.w_up_grid = w1[480:482].contiguous()
        w0 = self.P["refine/PointShuffle/conv0/weights"]
        self.w_c0_feat = w0[6:134].contiguous()
        # the three 1x1 convs that read up128 (conv_kv, conv_query, the feature part of PointShuffle2's conv0) as ONE
        # [128, 320] GEMM: columns 0:128 K|V, 128:192 Q, 192:320 conv0 features (its bias is added in ps_prep)
        nlp = "refine/PointShuffle/PointShuffle/"
        self.w_up3 = torch.cat([self.P[nlp + "conv_kv/weights"], self.P[nlp + "conv_query/weights"], self.w_c0_feat], dim=1).contiguous()
        self.b_up3 = torch.cat([self.P[nlp + "conv_kv/biases"], self.P[nlp + "conv_query/biases"],
                                torch.zeros(128, dtype=torch.float32, device=dev)]).contiguous()
        wsk = self.P["refine/PointShuffle/skip/weights"]
        self.w_skip_pad = torch.cat([wsk, torch.zeros((10, wsk.shape[1]), dtype=torch.float32, device=dev)], dim=0).contiguous()

    def _w(self, scope):
        return self.P[scope + "/weights"], self.P[scope + "/biases"]

    # ---------------------------------------------------------------------------------------- workspace ----
    def _workspace(self, B, N):
        key = (B, N)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, f32, i32 = self.device, torch.float32, torch.int32
        M = N * self.up_ratio
        rn, rm, k = B * N, B * M, K_NEIGH

        def E(*shape, dtype=f32):
            return torch.empty(shape, dtype=dtype, device=dev)

        ws = dict(
            feat=E(rn, 480), prep=E(rn, 48), prep_b=E(rn, 48), kidx=E(rn, k + 1, dtype=i32), h256=E(rn, 256),
            up256=E(rm, 256), up128=E(rm, 128), c256=E(rm, 256), c64=E(rm, 64), coarse=E(B, M, 3),
            psidx=E(rm, k, dtype=i32), up3=E(rm, 320), att=E(rm, 64), nl=E(rm, 256),
            skipin=torch.zeros((rm, 144), dtype=f32, device=dev), skip=E(rm, 256), am=E(rm, 128),
            fp=E(rm, 2048), aft=E(rm, 256), agg=E(rm, 256),
            f256=E(rm, 256), f64=E(rm, 64), fine=E(B, M, 3))
        ws["kv"], ws["q"], ws["gm"] = ws["up3"][:, 0:128], ws["up3"][:, 128:192], ws["up3"][:, 192:320]   # views, row stride 320
        self._ws[key] = ws
        return ws

    def _pair_buffers(self, B, N):
        """[B*M*16, .] pair tensors of the UNFUSED local cell (A/B testing only; the default path never allocates them)."""
        key = ("pair", B, N)
        if key not in self._ws:
            rows = B * N * self.up_ratio * K_NEIGH
            E = lambda c: torch.empty((rows, c), dtype=torch.float32, device=self.device)
            self._ws[key] = (E(128), E(128), E(16))
        return self._ws[key]

    # ------------------------------------------------------------------------------------------- launch ----
    def _call(self, name, fn, *args):
        """One C-ABI launch; when self.profile is a list, bracket it with HIP events on the launch stream."""
        if self.profile is None:
            _lib.check(fn(*args), name)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(fn(*args), name)
        e1.record()
        self.profile.append((name, e0, e1))

    def _linear(self, st, X, K, W, bias, act, Y, N, M=None, ldx=None, ldw=None, ldy=None, batch=1, sx=0, sw=0, sy=0,
                transb=0, R1=None, R2=None, xoff=0, woff=0, yoff=0):
        """Y[:, yoff:yoff+N] = R2 + R1 + act(X[:, xoff:xoff+K] . W + bias) via dispu_linear (element offsets in floats)."""
        L = _lib.lib()
        M = X.shape[0] if M is None else M
        ldx = X.stride(0) if ldx is None else ldx
        ldw = W.stride(0) if ldw is None else ldw
        ldy = Y.stride(0) if ldy is None else ldy
        p = lambda t, off=0: _lib.C.c_void_p(t.data_ptr() + 4 * off) if t is not None else _lib.C.c_void_p(0)
        name = "linear"
        if self.profile is not None:     # "linear<BM, BN, 2, 2, BK, transb, edge, epi>[MxKxN]": the instantiation rocprofv3 reports
            t = L.dispu_linear_tile2(batch, M, K, N, int(bool(transb)))
            bm, bn = {128257: (128, 256), 128128: (128, 128), 64128: (64, 128), 128064: (128, 64)}.get(t, (64, 64))
            al = lambda q, o=0: q is None or (q.data_ptr() + 4 * o) % 16 == 0
            ok = (M % bm == 0 and N % bn == 0 and ldx % 4 == 0 and ldw % 4 == 0 and sx % 4 == 0 and sw % 4 == 0 and al(X, xoff) and al(W, woff)
                  and ldy % 4 == 0 and sy % 4 == 0 and al(Y, yoff) and al(bias) and (R1 is None or (R1.stride(0) % 4 == 0 and al(R1)))
                  and (R2 is None or (R2.stride(0) % 4 == 0 and al(R2))))
            bkr = 16 if t == 128257 else 32                 # register-staged paths (transposed B, edge tiles) of the smaller tiles: BK 32
            if ok and not transb and K % 16 == 0:
                bk, edge = 16, False                        # the DMA pipeline
            elif ok and K % bkr == 0:
                bk, edge = bkr, False
            else:
                bk, edge = bkr, True
            epi = 0 if (R1 is None and R2 is None) else 4       # epilogue variant: 0 bias/act, 4 with residual inputs
            if epi == 0 and (bm, bn, bk) == (128, 256, 16) and not transb and not edge and K >= 1024:
                epi = 6                                          # long contractions: an instantiation of their own (csrc/linear.hip)
            name = "linear<%d, %d, 2, 2, %d, %s, %s, %d>[%dx%dx%d]" % (bm, bn, bk, "true" if transb else "false",
                                                                      "true" if edge else "false", epi, M * batch, K, N)
            # latency-bound shapes leave the tiled kernel (csrc/linear_skinny.hip:linear_skinny_dispatch; same conditions)
            tiles64 = ((M + 63) // 64) * ((N + 63) // 64)
            if (batch == 1 and R2 is None and 4 <= K <= 384 and K % 4 == 0 and ldx % 4 == 0 and (X.data_ptr() + 4 * xoff) % 16 == 0
                    and not (transb and (ldw % 4 or (W.data_ptr() + 4 * woff) % 16))
                    and ((N <= 64 and tiles64 < 256) or N <= 32 or (K <= 32 and N <= 128))):
                name = "linear_skinny<%d, %s>[%dx%dx%d]" % (2 if K <= 32 else 8 if K <= 128 else 16 if K <= 256 else 24,
                                                           "true" if transb else "false", M, K, N)
        self._call(name, L.dispu_linear, batch, M, K, N,
                   p(X, xoff), ldx, sx, p(W, woff), ldw, sw, transb, p(bias), act, p(Y, yoff), ldy, sy, p(R1),
                   R1.stride(0) if R1 is not None else 0, 0, p(R2), R2.stride(0) if R2 is not None else 0, 0, st)

    def __call__(self, inputs):
        return self.forward(inputs)

    @property
    def trainer(self):
        """the Trainer behind a Generator(is_training=True): loss_backward / backward / adam / train_step (train.py)."""
        return self._trainer

    def forward(self, inputs):
        if not self.P:
            raise RuntimeError("Generator has no parameters: call load_params() first")
        if self.is_training:
            coarse, fine = self._trainer.forward(inputs)
            return (coarse, fine) if self.return_views else (coarse.clone(), fine.clone())
        if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 3
                and inputs.shape[2] == 3):
            raise ValueError("Generator expects a float32 [B,N,3] tensor on a ROCm device")
        inputs = inputs.contiguous()
        B, N, _ = inputs.shape
        M = N * self.up_ratio
        if B > self.MAX_BATCH:                                  # patches are independent: run the batch in chunks
            if self.return_views:
                raise ValueError("return_views needs B <= %d (one workspace)" % self.MAX_BATCH)
            outs = [self.forward(inputs[lo:lo + self.MAX_BATCH]) for lo in range(0, B, self.MAX_BATCH)]
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        rn, rm, k = B * N, B * M, K_NEIGH
        ws = self._workspace(B, N)
        L = _lib.lib()
        st = _lib.stream_ptr(inputs.device)
        ptr = _lib.ptr
        off = lambda t, o: _lib.C.c_void_p(t.data_ptr() + 4 * o)
        feat = ws["feat"]
        fe = "generator/feature_extraction_coarse/"

        # ---- feature_extraction_GCN (ops.py:1437-1486): features accumulate right-to-left inside feat[:, 0:480]
        wl0, bl0 = self._w(fe + "layer0")
        stem = self.fused_stem and N <= 256 and N % 2 == 0 and N > k
        if not stem:       # (fused stem: the first block's launch evaluates layer0 while it stages its cloud)
            self._call("layer0", L.dispu_linear_small_k, rn, 3, 24, ptr(inputs), 3, ptr(wl0), ptr(bl0), 0, off(feat, 456), 480, st)
        col = 456          # left edge of the features produced so far
        prep_done = False  # the previous block's launch already ran this block's bottleneck conv (fused stem)
        for d in range(1, DENSE_BLOCKS + 1):
            pbuf = ws["prep"] if d % 2 == 0 else ws["prep_b"]        # block d reads pbuf while its launch writes the other one
            if d == 1:
                F, ldf, foff, C = feat, 480, 456, 24
            else:
                if not prep_done:
                    w, b = self._w(fe + "layer%d_prep" % d)
                    self._linear(st, feat, 480 - col, w, b, 1, pbuf, 48, xoff=col)
                F, ldf, foff, C = pbuf, 48, 0, 48
            w0, b0 = self._w(fe + "layer%d/l0" % d)
            w1, b1 = self._w(fe + "layer%d/l1" % d)
            w2, b2 = self._w(fe + "layer%d/l2" % d)
            width = 3 * GROWTH + C
            col -= width
            if stem:
                # + the next block's bottleneck conv over [this block's outputs | the 480 - (col + width) older columns] for d < 4
                nxt = ws["prep_b"] if d % 2 == 0 else ws["prep"]
                wp, bp = self._w(fe + "layer%d_prep" % (d + 1)) if d < DENSE_BLOCKS else (None, None)
                self._call("stem_block", L.dispu_stem_block, rn, N, C, off(F, foff), ldf, k + 1, 1, ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(w2),
                           ptr(b2), off(feat, col), 480, ptr(ws["kidx"]) if self.keep_intermediates else None,
                           ptr(wp) if d < DENSE_BLOCKS else None, ptr(bp) if d < DENSE_BLOCKS else None, 480 - col - width,
                           ptr(nxt) if d < DENSE_BLOCKS else None, 48, ptr(inputs) if d == 1 else None, ptr(wl0) if d == 1 else None,
                           ptr(bl0) if d == 1 else None, off(feat, 456) if d == 1 else None, 480, st)
                prep_done = d < DENSE_BLOCKS
                continue
            prep_done = False
            nbf = L.dispu_knn_feat_scratch_bytes(B, N, N, C, k + 1)   # > 0 for 512 < N <= 4096 (second pass of 16x): chunked search
            if nbf and (ws.get("knnf_scratch") is None or ws["knnf_scratch"].numel() < nbf):
                ws["knnf_scratch"] = torch.empty((nbf,), dtype=torch.uint8, device=self.device)
            self._call("knn_feat", L.dispu_knn_feat_strided_ws, B, N, N, C, k + 1, off(F, foff), ldf, off(F, foff), ldf, None, ptr(ws["kidx"]),
                       ptr(ws["knnf_scratch"]) if nbf else None, nbf, st)
            self._call("edge_dense_conv", L.dispu_edge_dense_conv, rn, N, C, off(F, foff), ldf, ptr(ws["kidx"]), k + 1, 1, ptr(w0), ptr(b0), ptr(w1),
                                        ptr(b1), ptr(w2), ptr(b2), off(feat, col), 480, st)
        assert col == 0

        # ---- duplicate_up (ops.py:1152-1199) + coarse coordinate_regressor (:1089-1110)
        _, b1 = self._w("generator/upshuffle_0/conv1")
        self._linear(st, feat, 480, self.w_up_feat, None, 0, ws["h256"], 256)
        cs = "generator/coarse_coordinate_regressor/"
        coarse = ws["coarse"]
        heads = self.fused_heads and rm % 128 == 0
        in_chain = heads and self.chain_inputs
        fine = ws["fine"]
        if self.fine_out is not None:
            fine = self.fine_out
            if not (isinstance(fine, torch.Tensor) and fine.device == inputs.device and fine.dtype == torch.float32
                    and tuple(fine.shape) == (B, M, 3) and fine.is_contiguous()):
                raise ValueError("fine_out must be a contiguous float32 [%d, %d, 3] tensor on %s" % (B, M, inputs.device))
        if not in_chain:
            self._call("dup_grid", L.dispu_dup_grid, B, N, 256, self.up_ratio, ptr(ws["h256"]), 256, ptr(self.w_up_grid), ptr(b1), ptr(self.grid),
                       ptr(ws["up256"]), 256, st)
        if in_chain:
            w1, b1_ = self._w("generator/upshuffle_0/conv2")
            w2, b2_ = self._w(cs + "fc_layer0")
            w3, b3_ = self._w(cs + "fc_layer1")
            w4, b4_ = self._w(cs + "fc_layer2")
            self._call("mlp_chain[coarse]", L.dispu_mlp_chain_dup, B, N, self.up_ratio, 256, 128, 256, 64, ptr(ws["h256"]), 256, ptr(self.w_up_grid),
                       ptr(b1), ptr(self.grid), ptr(w1), ptr(b1_), ptr(w2), ptr(b2_), ptr(w3), ptr(b3_), ptr(w4), ptr(b4_), ptr(ws["up128"]), 128,
                       0, None, 0, ptr(coarse), 3, st)
        elif heads:
            # conv2 -> fc_layer0 -> fc_layer1 -> fc_layer2 in one launch; up128 (needed by PointShuffle2) is written on the way
            w1, b1_ = self._w("generator/upshuffle_0/conv2")
            w2, b2_ = self._w(cs + "fc_layer0")
            w3, b3_ = self._w(cs + "fc_layer1")
            w4, b4_ = self._w(cs + "fc_layer2")
            self._call("mlp_chain[coarse]", L.dispu_mlp_chain, rm, 256, 128, 256, 64, ptr(ws["up256"]), 256, ptr(w1), ptr(b1_), ptr(w2), ptr(b2_),
                       ptr(w3), ptr(b3_), ptr(w4), ptr(b4_), ptr(ws["up128"]), 128, 0, None, 0, ptr(coarse), 3, st)
        else:
            w, b = self._w("generator/upshuffle_0/conv2")
            self._linear(st, ws["up256"], 256, w, b, 1, ws["up128"], 128)
            w, b = self._w(cs + "fc_layer0")
            self._linear(st, ws["up128"], 128, w, b, 1, ws["c256"], 256)
            w, b = self._w(cs + "fc_layer1")
            self._linear(st, ws["c256"], 256, w, b, 1, ws["c64"], 64)
            w, b = self._w(cs + "fc_layer2")
            self._call("coarse", L.dispu_linear_small_n, rm, 64, 3, ptr(ws["c64"]), 64, ptr(w), ptr(b), 0, None, 0, ptr(coarse), 3, st)

        # ---- PointShuffle2 (ops.py:1012-1087)
        ps = "refine/PointShuffle/"
        up128 = ws["up128"]
        # The non-local cell reads up128 only.  With self.branches it runs on a second stream next to the grouping / skip / local
        # cell (fork here; the local cell waits for up3 = conv0's feature part, after_conv for nl): the small launches of one side
        # fill the CUs the other leaves idle.  Captured into the same hipGraph as two parallel branches.
        br = self.branches
        if br:
            if self._aux is None:
                self._aux = (torch.cuda.Stream(device=self.device), torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
            aux, ev_fork, ev_up3, ev_nl = self._aux
            ev_fork.record(torch.cuda.current_stream(self.device))
            aux.wait_event(ev_fork)
        with (torch.cuda.stream(aux) if br else contextlib.nullcontext()):
            sb = _lib.C.c_void_p(aux.cuda_stream) if br else st
            # PointNonLocalCell (ops.py:302-346)
            if self.split_up3:
                # N = 320 as 256 + 64 columns: each launch reads up128 ONCE (128 x 256 / 128 x 64 tiles); one launch with 128 x 64
                # tiles re-read it five times (128 MB of counter traffic against 59 MB algorithmic in round 1)
                self._linear(sb, up128, 128, self.w_up3, self.b_up3, 0, ws["up3"], 256)
                self._linear(sb, up128, 128, self.w_up3, self.b_up3[256:], 0, ws["up3"], 64, woff=256, yoff=256)
            else:
                self._linear(sb, up128, 128, self.w_up3, self.b_up3, 0, ws["up3"], 320)  # K|V, Q and conv0's feature part at once
            if br:
                ev_up3.record(aux)
            w_bp, b_bp = self._w(ps + "PointShuffle/conv_back_project")
            projected = False
            if self.fused_attention and M % 32 == 0 and self.fused_project:
                # softmax(Q.K^T / 8).V.W_bp on chip: neither the [B, M, M] logits nor the [B*M, 64] attention output reach HBM
                self._call("attention_project", L.dispu_attention_project, B, M, M, 64, ptr(ws["q"]), 320, ptr(ws["kv"]), 320,
                           off(ws["kv"], 64), 320, 0.125, ptr(w_bp), ptr(b_bp), 256, ptr(ws["nl"]), 256, sb)
                projected = True
            elif self.fused_attention and M % 32 == 0:
                # softmax(Q.K^T / 8).V on chip (flash style): the [B, M, M] logits never exist in HBM
                self._call("attention", L.dispu_attention, B, M, M, 64, ptr(ws["q"]), 320, ptr(ws["kv"]), 320, off(ws["kv"], 64), 320,
                           0.125, ptr(ws["att"]), 64, sb)
            else:
                key = ("scores", B, N)
                if key not in self._ws:
                    self._ws[key] = torch.empty((B, M, M), dtype=torch.float32, device=self.device)
                s = self._ws[key]
                self._linear(sb, ws["q"], 64, ws["kv"], None, 0, s, M, M=M, ldx=320, ldw=320, ldy=M, batch=B, sx=M * 320,
                             sw=M * 320, sy=M * M, transb=1)
                self._call("softmax", L.dispu_softmax_rows, rm, M, 0.125, ptr(s), M, sb)
                self._linear(sb, s, M, ws["kv"], None, 0, ws["att"], 64, M=M, ldx=M, ldw=320, ldy=64, batch=B, sx=M * M,
                             sw=M * 320, sy=M * 64, woff=64)
            if not projected:
                self._linear(sb, ws["att"], 64, w_bp, b_bp, 1, ws["nl"], 256)
            if br:
                ev_nl.record(aux)
        nb = L.dispu_knn_xyz_scratch_bytes(B, M, M, k)             # > 0 for M > 1024 (second pass of 16x upsampling): chunked search
        if nb and ws.get("knn_scratch") is None:
            ws["knn_scratch"] = torch.empty((nb,), dtype=torch.uint8, device=self.device)
        self._call("knn_xyz", L.dispu_knn_xyz_ws, B, M, M, k, ptr(coarse), ptr(coarse), ptr(ws["psidx"]), None,
                   ptr(ws["knn_scratch"]) if nb else None, nb, _lib.ARITH_PLAIN, st)
        # skip connection
        self._call("skip_max", L.dispu_ps_skip_max, rm, M, k, 128, ptr(ws["psidx"]), ptr(coarse), ptr(up128), 128, ptr(ws["skipin"]), 144, st)
        _, b = self._w(ps + "skip")
        # K padded 134 -> 144 with zero columns / zero weight rows (exact) so the GEMM takes its predicate-free path
        self._linear(st, ws["skipin"], 144, self.w_skip_pad, b, 1, ws["skip"], 256)
        # local cell: conv0 per source point, conv1 per pair
        if br:
            torch.cuda.current_stream(self.device).wait_event(ev_up3)
        w0, b0 = self._w(ps + "conv0")
        self._call("ps_prep", L.dispu_ps_prep, rm, 128, ptr(coarse), ptr(w0), ptr(b0), ptr(ws["gm"]), 320, ptr(ws["am"]), 128, st)
        w1, b1 = self._w(ps + "conv1")
        ww, bw = self._w(ps + "weight_net/wconv0")
        if self.fused_local:
            # gather_sub_relu + conv1 + weight_net + feature x weight in one kernel: only F' [rm, 2048] touches HBM
            self._call("ps_local", L.dispu_ps_local, rm, M, k, 128, ptr(ws["psidx"]), ptr(coarse), ptr(ws["gm"]), 320, ptr(ws["am"]),
                       ptr(w1), ptr(b1), ptr(ww), ptr(bw), ptr(self.bn_scale), ptr(self.bn_shift), ptr(ws["fp"]), st)
        else:
            x1, x2, wv = self._pair_buffers(B, N)
            self._call("gather_sub_relu", L.dispu_ps_gather_sub_relu, rm, M, k, 128, ptr(ws["psidx"]), ptr(ws["gm"]), 320,
                       ptr(ws["am"]), 128, ptr(x1), 128, st)
            self._linear(st, x1, 128, w1, b1, 1, x2, 128)
            self._call("weight_net", L.dispu_ps_weight_net, rm, M, k, 16, ptr(ws["psidx"]), ptr(coarse), ptr(ww), ptr(bw),
                       ptr(self.bn_scale), ptr(self.bn_shift), ptr(wv), st)
            self._call("point_matmul", L.dispu_ps_point_matmul, rm, k, 128, 16, ptr(x2), 128, ptr(wv), ptr(ws["fp"]), 2048, st)
        w, b = self._w(ps + "after_conv")
        nl_late = br and heads and self.chain_inputs      # nl is first read by the fine chain's loader
        if br and not nl_late:
            torch.cuda.current_stream(self.device).wait_event(ev_nl)
        if self.split_bf16 and rm % 128 == 0:
            pl = self._planes.get("after_conv")
            if pl is None:
                pl = torch.empty((3 * 2048 * 256,), dtype=torch.bfloat16, device=self.device)
                _lib.check(L.dispu_bf16x3_split_weights(2048, 256, ptr(w), 256, ptr(pl), st), "dispu_bf16x3_split_weights")
                self._planes["after_conv"] = pl
            in_chain = heads and self.chain_inputs            # + skip + nl then happen in the fine chain's loader
            self._call("linear_bf16x3[%dx2048x256]" % rm, L.dispu_linear_bf16x3, rm, 2048, 256, ptr(ws["fp"]), 2048, ptr(pl), ptr(b), 1,
                       ptr(ws["aft"]), 256, None if in_chain else ptr(ws["skip"]), 256, None if in_chain else ptr(ws["nl"]), 256, st)
        elif heads and self.chain_inputs:
            self._linear(st, ws["fp"], 2048, w, b, 1, ws["aft"], 256)          # relu(after_conv) alone; + skip + nl in the fine chain's loader
        elif self.fused_residual:
            self._linear(st, ws["fp"], 2048, w, b, 1, ws["aft"], 256, R1=ws["skip"], R2=ws["nl"])
        else:
            # same arithmetic order ((act(.) + skip) + nl) in a separate streaming kernel
            self._linear(st, ws["fp"], 2048, w, b, 1, ws["aft"], 256)
            self._call("add3", L.dispu_add3, rm * 256, ptr(ws["aft"]), ptr(ws["skip"]), ptr(ws["nl"]), ptr(ws["aft"]), st)
        fs = "refine/fine_coordinate_regressor/"
        if nl_late:
            torch.cuda.current_stream(self.device).wait_event(ev_nl)
        if heads and self.chain_inputs:
            w1, b1_ = self._w(ps + "aggregation")
            w2, b2_ = self._w(fs + "fc_layer0")
            w3, b3_ = self._w(fs + "fc_layer1")
            w4, b4_ = self._w(fs + "fc_layer2")
            self._call("mlp_chain[fine]", L.dispu_mlp_chain_sum3, rm, 256, 256, 256, 64, ptr(ws["aft"]), ptr(ws["skip"]), ptr(ws["nl"]), 256,
                       ptr(w1), ptr(b1_), ptr(w2), ptr(b2_), ptr(w3), ptr(b3_), ptr(w4), ptr(b4_),
                       ptr(ws["agg"]) if self.keep_intermediates else None, 256, 1, ptr(coarse), 3, ptr(fine), 3, st)
        elif heads:
            w1, b1_ = self._w(ps + "aggregation")
            w2, b2_ = self._w(fs + "fc_layer0")
            w3, b3_ = self._w(fs + "fc_layer1")
            w4, b4_ = self._w(fs + "fc_layer2")
            self._call("mlp_chain[fine]", L.dispu_mlp_chain, rm, 256, 256, 256, 64, ptr(ws["aft"]), 256, ptr(w1), ptr(b1_), ptr(w2), ptr(b2_),
                       ptr(w3), ptr(b3_), ptr(w4), ptr(b4_), ptr(ws["agg"]) if self.keep_intermediates else None, 256, 1, ptr(coarse), 3,
                       ptr(fine), 3, st)
        else:
            w, b = self._w(ps + "aggregation")
            self._linear(st, ws["aft"], 256, w, b, 1, ws["agg"], 256)
            # ---- fine coordinate_regressor (is_off) + residual (generator.py:76-81)
            w, b = self._w(fs + "fc_layer0")
            self._linear(st, ws["agg"], 256, w, b, 1, ws["f256"], 256)
            w, b = self._w(fs + "fc_layer1")
            self._linear(st, ws["f256"], 256, w, b, 1, ws["f64"], 64)
            w, b = self._w(fs + "fc_layer2")
            self._call("fine", L.dispu_linear_small_n, rm, 64, 3, ptr(ws["f64"]), 64, ptr(w), ptr(b), 1, ptr(coarse), 3, ptr(fine), 3, st)
        if self.return_views:
            return coarse, fine
        return coarse.clone(), fine.clone()
