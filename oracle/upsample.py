"""CPU ORACLE (test infrastructure) of the whole-cloud test path: DisPU/model.py:306-381 with
Common/pc_util.py:83-92 (extract_knn_patch) and :147-161 (normalize_point_cloud).  numpy over the other oracles."""
import numpy as np

from . import generator as OG
from . import oracle as O


def normalize_point_cloud(p):
    """pc_util.py:147-161 for [N,3] or [B,N,3]."""
    axis = 0 if p.ndim == 2 else 1
    c = np.mean(p, axis=axis, keepdims=True)
    q = p - c
    f = np.amax(np.sqrt(np.sum(q ** 2, axis=-1, keepdims=True)), axis=axis, keepdims=True)
    return q / f, c, f


def extract_knn_patch_idx(queries, pc, k):
    """pc_util.py:83-92 (sklearn NearestNeighbors; tie order unspecified there, lower index first here)."""
    d = queries[:, None, :] - pc[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return np.argsort(d2.astype(np.float32), axis=1, kind="stable")[:, :k].astype(np.int32)


def upsample_cloud(P, pc, patch_num_point=256, patch_num_ratio=3, final_ratio=4):
    pc = np.ascontiguousarray(pc, np.float32)
    n = pc.shape[0]
    cloud_n, c0, f0 = normalize_point_cloud(pc)
    seed_num = int(n / patch_num_point * patch_num_ratio)
    seeds = O.farthest_point_sample(seed_num, cloud_n[None])[0]
    pidx = extract_knn_patch_idx(cloud_n[seeds], cloud_n, patch_num_point)
    patches = cloud_n[pidx]
    pn, pc_c, pc_f = normalize_point_cloud(patches)
    coarse, fine = OG.generator_forward(P, pn.astype(np.float32))
    pred = pc_c + fine * pc_f
    merged = pred.reshape(-1, 3) * f0 + c0
    merged = merged.astype(np.float32)
    sel = O.farthest_point_sample(int(n * final_ratio), merged[None])[0]
    return merged[sel], dict(cloud_n=cloud_n, seeds=seeds, pidx=pidx, patches_n=pn.astype(np.float32), fine=fine, merged=merged, sel=sel)
