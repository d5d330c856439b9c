"""Evaluator: counterpart of /evaluate.py:33-41,153-162 of the reference (CD and Hausdorff between a predicted and a
ground-truth cloud, both normalised first with Common/ops.py:1954-1963).  The reference builds a TF graph around
tf_nndistance and runs one cloud pair per session call; here both clouds are normalised and matched on the device
(dispu_normalize_patches + dispu_nn_distance at (1, 8192, 8192)) and only four scalars travel to the host."""
import csv
import os
from glob import glob

import numpy as np
import torch

from . import _lib
from .tf_nndistance import nn_distance
from .upsample import normalize_patches


def evaluate_pair(pred, gt):
    """pred [n,3], gt [m,3] device tensors or arrays -> {"CD": mean fwd + mean bwd, "hausdorff": max fwd + max bwd}."""
    dev = pred.device if isinstance(pred, torch.Tensor) else torch.device("cuda:0")
    p = (pred if isinstance(pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pred[:, :3], np.float32)).to(dev))
    g = (gt if isinstance(gt, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(gt[:, :3], np.float32)).to(dev))
    pn = normalize_patches(p.reshape(1, -1, 3).contiguous())[0]
    gn = normalize_patches(g.reshape(1, -1, 3).contiguous())[0]
    fwd, _, bwd, _ = nn_distance(pn, gn)
    L = _lib.lib()
    out = torch.empty(4, dtype=torch.float32, device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(L.dispu_row_mean_max(1, fwd.shape[1], _lib.ptr(fwd), _lib.ptr(out), _lib.C.c_void_p(out.data_ptr() + 4), st), "row_mean_max")
    _lib.check(L.dispu_row_mean_max(1, bwd.shape[1], _lib.ptr(bwd), _lib.C.c_void_p(out.data_ptr() + 8),
                                    _lib.C.c_void_p(out.data_ptr() + 12), st), "row_mean_max")
    mf, xf, mb, xb = (float(v) for v in out.cpu())
    return {"CD": mf + mb, "hausdorff": xf + xb, "cd_forward": mf, "cd_backward": mb}


def evaluate_dirs(pred_dir, gt_dir, csv_name="evaluation.csv"):
    """evaluate.py:128-175 (CD / hausdorff columns): every gt/<name>.xyz against pred/<name>.xyz; writes the CSV next to
    the predictions and returns the rows plus the averages."""
    rows = []
    for gt_path in sorted(glob(os.path.join(gt_dir, "*.xyz"))):
        name = os.path.basename(gt_path)
        pred_path = os.path.join(pred_dir, name)
        if not os.path.isfile(pred_path):
            continue
        r = evaluate_pair(np.loadtxt(pred_path)[:, :3], np.loadtxt(gt_path)[:, :3])
        rows.append({"name": name, "CD": r["CD"], "hausdorff": r["hausdorff"]})
    if rows:
        avg = {"name": "avg", "CD": float(np.mean([r["CD"] for r in rows])), "hausdorff": float(np.mean([r["hausdorff"] for r in rows]))}
        with open(os.path.join(pred_dir, csv_name), "w") as f:
            w = csv.DictWriter(f, fieldnames=["name", "CD", "hausdorff"], restval="-", extrasaction="ignore")
            w.writeheader()
            for r in rows + [avg]:
                w.writerow(r)
    return rows
