import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import generator as OG, train_oracle as T
from dispu_amd import synth
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = OG.init_params(seed=1234, bias_scale=0.05, bn_random=True)
x, gt = synth.patch_with_gt(2, 256, 1024, seed=seed)
radius = np.array([1.0, 1.3], np.float32)
loss, terms, grads, bn, (coarse, fine) = T.loss_and_grads(P, x, gt, radius, epoch=0)
tr = Trainer(params=P, device=dev)
tr.zero_grad()
tr.forward(torch.from_numpy(x).to(dev))
tr.loss_backward(torch.from_numpy(gt).to(dev), torch.from_numpy(radius).to(dev))
tr.backward()
torch.cuda.synchronize()
got = tr.grads()
for k, r in grads.items():
    g = got[k].astype(np.float64)
    print("%-70s relmax %.2e  relL2 %.2e  max|g| %.2e" % (k[-70:], np.abs(g - r).max() / max(np.abs(r).max(), 1e-30),
          np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30), np.abs(r).max()))
ws = tr._ws[(2, 256)]
k = "generator/feature_extraction_coarse/layer1/l0/weights"
err = np.abs(got[k].astype(np.float64) - grads[k]) / np.abs(grads[k]).max()
print("l0 W err rows(max over cols):", np.round(err.max(1) * 1e4, 1))
print("l0 W err cols(max over rows):", np.round(err.max(0) * 1e4, 1))
Eb = ws["edge"][1].cpu().numpy().astype(np.float64)
feat = ws["feat"].cpu().numpy().astype(np.float64)
kidx = ws["kidx"][1].cpu().numpy()[:, 1:]
F = feat[:, 456:480].reshape(2, 256, 24)
nbr = np.stack([F[b][kidx.reshape(2, 256, 16)[b]] for b in range(2)])
cen = np.broadcast_to(F[:, :, None, :], nbr.shape)
E0 = np.concatenate([cen, nbr - cen], -1).reshape(-1, 48)
print("edge feature err", np.abs(E0 - Eb[:, 72:120]).max())
W = P[k].astype(np.float64); b = P[k.replace("weights", "biases")].astype(np.float64)
l0 = np.maximum(E0 @ W + b, 0)
print("l0 err", np.abs(l0 - Eb[:, 48:72]).max(), "l0 scale", np.abs(l0).max())
dE = ws["dedge"].cpu().numpy()
print("dE finite", np.isfinite(dE).all())
# ---- block-1 backward recomputed in float64 from the HIP buffers
dfeat = ws["dfeat"].cpu().numpy().astype(np.float64)
g = dfeat[:, 360:456]                                   # d loss / d max-pooled block-1 output
Y = feat[:, 360:456]
Ev = Eb[:, :96].reshape(512, 16, 96)
ind = (Ev == Y[:, None, :]).astype(np.float64)
dmax = ind / ind.sum(1, keepdims=True) * g[:, None, :]
dmax = dmax.reshape(-1, 96)
sc = "generator/feature_extraction_coarse/layer1"
W2 = P[sc + "/l2/weights"].astype(np.float64); W1 = P[sc + "/l1/weights"].astype(np.float64); W0 = P[sc + "/l0/weights"].astype(np.float64)
d = np.zeros((8192, 120)); d[:, :96] = dmax
dz2 = d[:, 0:24]
d[:, 24:96] += dz2 @ W2.T
dz1 = d[:, 24:48] * (Eb[:, 24:48] > 0)
d[:, 48:96] += dz1 @ W1.T
dz0 = d[:, 48:72] * (Eb[:, 48:72] > 0)
d[:, 72:120] += dz0 @ W0.T
dEh = dE[:, :120].astype(np.float64)
print("dz2 err", np.abs(dEh[:, 0:24] - dz2).max() / np.abs(dz2).max())
print("dz1 err", np.abs(dEh[:, 24:48] - dz1).max() / np.abs(dz1).max())
print("dz0 err", np.abs(dEh[:, 48:72] - dz0).max() / np.abs(dz0).max())
print("dx err", np.abs(dEh[:, 72:120] - d[:, 72:120]).max() / np.abs(d[:, 72:120]).max())
dW0 = Eb[:, 72:120].T @ dz0
print("dW0 (f64 from HIP buffers) vs HIP", np.abs(dW0 - got[k]).max() / np.abs(dW0).max(), " vs oracle", np.abs(dW0 - grads[k]).max() / np.abs(dW0).max())
print("ties in block-1 max (count>1), by column group:", [(int((ind.sum(1)[:, a:b] > 1).sum())) for a, b in ((0, 24), (24, 48), (48, 72), (72, 96))])
# ---- oracle gradients w.r.t. the dense blocks' outputs
Pt = T.to_torch(P); idx = T.neighbour_indices(P, x); tap = {}
c_, f_ = T.generator_forward(Pt, x, idx, True, {}, tap)
for kk in ("dc1", "dc2", "dc3", "dc4", "up128", "coarse", "fine_feat"): tap[kk].retain_grad()
f_.retain_grad()
tot, _ = T.pu_loss(c_, f_, torch.as_tensor(gt, dtype=T.DT), radius, 0)
tot.backward()
for d_, (a, b) in zip((1, 2, 3, 4), ((360, 456), (240, 360), (120, 240), (0, 120))):
    r = tap["dc%d" % d_].grad.numpy().reshape(512, -1)
    h = dfeat[:, a:b]
    e = np.abs(h - r).max(0) / np.abs(r).max()
    print("dc%d grad relmax %.2e; per-col x1e4:" % (d_, e.max()), np.round(e * 1e4, 1)[:96])
r = tap["up128"].grad.numpy().reshape(2048, 128); h = ws["dup128"].cpu().numpy()
print("dup128 relmax", np.abs(h - r).max() / np.abs(r).max())
r = tap["coarse"].grad.numpy().reshape(2048, 3); h = ws["dcoarse"].cpu().numpy().reshape(2048, 3)
print("dcoarse relmax", np.abs(h - r).max() / np.abs(r).max())
r = tap["fine_feat"].grad.numpy().reshape(2048, 256); h = ws["dagg"].cpu().numpy()
print("dagg(masked) vs fine_feat grad relmax", np.abs(h - r * (ws["agg"].cpu().numpy() > 0)).max() / np.abs(r).max())
hm = ws["dagg"].cpu().numpy(); rr = tap["fine_feat"].grad.numpy().reshape(2048, 256) * (ws["agg"].cpu().numpy() > 0)
print("dagg relL2", np.linalg.norm(hm - rr) / np.linalg.norm(rr))
e = np.abs(hm - rr); i, j = np.unravel_index(e.argmax(), e.shape); print("worst entry", i, j, hm[i, j], rr[i, j], "row err", e[i].max(), "row L2 rel", np.linalg.norm(hm[i]-rr[i])/np.linalg.norm(rr[i]))
rowerr = np.linalg.norm(hm - rr, axis=1) / np.maximum(np.linalg.norm(rr, axis=1), 1e-30)
print("rows with rel err > 1e-3:", int((rowerr > 1e-3).sum()), "of", len(rowerr), " median", np.median(rowerr))
fo = f_.detach().numpy().reshape(2048, 3); fh = ws["fine"].cpu().numpy().reshape(2048, 3)
print("fine abs err max", np.abs(fo - fh).max())
dfo = None

r = f_.grad.numpy().reshape(2048, 3); h = ws["dfine"].cpu().numpy().reshape(2048, 3)
re = np.linalg.norm(h - r, axis=1) / np.maximum(np.linalg.norm(r, axis=1), 1e-30)
print("dfine rows with rel err > 1e-3:", np.nonzero(re > 1e-3)[0], re[re > 1e-3], "median", np.median(re))
for d_, (a, b) in zip((1, 2, 3, 4), ((360, 456), (240, 360), (120, 240), (0, 120))):
    r = tap["dc%d" % d_].grad.numpy().reshape(512, -1); h = dfeat[:, a:b]
    re = np.linalg.norm(h - r, axis=1) / np.maximum(np.linalg.norm(r, axis=1), 1e-30)
    print("dc%d rows with rel err > 1e-3:" % d_, np.nonzero(re > 1e-3)[0][:20], np.round(re[re > 1e-3][:20], 4), "median", np.median(re))
