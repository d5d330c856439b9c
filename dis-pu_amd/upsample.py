"""Whole-cloud upsampling (the reference's `Model.test()` data path, DisPU/model.py:306-381) on one MI355X:

    normalise cloud -> FPS seeds (N/256*3) -> 256-NN patches -> per-patch normalise -> generator (ALL patches in one
    batch; the reference feeds them one by one with batch size 1) -> de-normalise -> concat -> FPS to final_ratio*N.

Every stage is a kernel of libdispu_hip.so; the cloud crosses the host boundary once in and once out
(the reference crosses it ~2 x 24 times per 2048-point cloud plus the nanoflann hop inside every generator call)."""
import math

import numpy as np
import torch

from . import _lib
from .tf_sampling import farthest_point_sample, gather_point


def knn_patch(cloud, queries, k):
    """pc_util.extract_knn_patch (:83-92): cloud[b,n,3], queries[b,m,3] -> idx[b,m,k] int32 (ascending distance)."""
    b, n, _ = cloud.shape
    m = queries.shape[1]
    idx = torch.empty((b, m, k), dtype=torch.int32, device=cloud.device)
    _lib.check(_lib.lib().dispu_knn_patch(b, n, m, k, _lib.ptr(cloud.contiguous()), _lib.ptr(queries.contiguous()), _lib.ptr(idx),
                                          _lib.stream_ptr(cloud.device)), "dispu_knn_patch")
    return idx


def normalize_patches(p):
    """pc_util.normalize_point_cloud (:147-161) per patch: -> (normalised [b,n,3], centroid [b,3], furthest [b])."""
    b, n, _ = p.shape
    out = torch.empty_like(p)
    c = torch.empty((b, 3), dtype=torch.float32, device=p.device)
    f = torch.empty((b,), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().dispu_normalize_patches(b, n, _lib.ptr(p.contiguous()), _lib.ptr(out), _lib.ptr(c), _lib.ptr(f),
                                                  _lib.stream_ptr(p.device)), "dispu_normalize_patches")
    return out, c, f


def denormalize_patches(p, centroid, furthest):
    b, m, _ = p.shape
    out = torch.empty_like(p)
    _lib.check(_lib.lib().dispu_denormalize_patches(b, m, _lib.ptr(p.contiguous()), _lib.ptr(centroid), _lib.ptr(furthest),
                                                    _lib.ptr(out), _lib.stream_ptr(p.device)), "dispu_denormalize_patches")
    return out


def generator_chain(gen, patches, final_ratio=4, step_ratio=4):
    """Model.build_model_test (DisPU/model.py:114-118): G once, then round(final_ratio ** (1/step_ratio)) - 1 more times on
    its own output (final_ratio 16 -> two passes: 256 -> 1024 -> 4096 points per patch)."""
    coarse, fine = gen(patches)
    for _ in range(round(math.pow(final_ratio, 1.0 / step_ratio)) - 1):
        coarse, fine = gen(fine if not gen.return_views else fine.clone())   # views alias the reusable workspace
    return coarse, fine


def upsample_clouds(gen, clouds, patch_num_point=256, patch_num_ratio=3, final_ratio=4, return_stages=False):
    """A BATCH of equally sized clouds [C, N, 3] -> [C, final_ratio*N, 3] (device tensor).  Every stage of Model.test
    (model.py:343-381) already is a batched kernel, so C clouds cost the launches of one: the seed FPS and the final
    FPS - m - 1 dependent rounds on ONE CU per cloud, 17 ms for 24576 -> 8192 - run C clouds on C CUs at once, and the
    generator sees all C * seed_num patches in one batch.  Cloud c of the result is bit-identical to upsample_cloud(clouds[c])."""
    dev = gen.device
    cloud = torch.as_tensor(np.ascontiguousarray(clouds, np.float32) if not isinstance(clouds, torch.Tensor) else clouds,
                            dtype=torch.float32, device=dev)
    if cloud.dim() != 3 or cloud.shape[2] != 3:
        raise ValueError("upsample_clouds expects [C, N, 3]")
    C, n, _ = cloud.shape
    cloud_n, c0, f0 = normalize_patches(cloud)                                   # whole-cloud normalisation (model.py:364)
    seed_num = int(n / patch_num_point * patch_num_ratio)
    seeds = farthest_point_sample(seed_num, cloud_n)                             # model.py:323
    seed_xyz = gather_point(cloud_n, seeds)
    pidx = knn_patch(cloud_n, seed_xyz, patch_num_point)                         # pc_util.extract_knn_patch
    patches = gather_point(cloud_n, pidx.reshape(C, -1)).reshape(C * seed_num, patch_num_point, 3)
    pn, pc_c, pc_f = normalize_patches(patches)                                  # model.py:306-308
    coarse, fine = generator_chain(gen, pn, final_ratio)
    pred = denormalize_patches(fine, pc_c, pc_f)                                 # model.py:310
    merged = denormalize_patches(pred.reshape(C, -1, 3), c0, f0)                 # model.py:371-372
    out_num = int(n * final_ratio)
    sel = farthest_point_sample(out_num, merged)                                 # model.py:375
    result = gather_point(merged, sel)
    if return_stages:
        return result, dict(cloud_n=cloud_n, seeds=seeds, pidx=pidx, patches_n=pn, fine=fine, merged=merged, sel=sel)
    return result


def upsample_cloud(gen, pc, patch_num_point=256, patch_num_ratio=3, final_ratio=4, return_stages=False):
    """pc: [N,3] float32 (numpy or device tensor) -> upsampled [final_ratio*N, 3] numpy array (model.py:343-381).
    `gen` is a dispu_amd.generator.Generator with up_ratio 4 (final_ratio 4: one generator pass, model.py:117-118)."""
    cloud = pc if isinstance(pc, torch.Tensor) else np.ascontiguousarray(pc, np.float32)
    res = upsample_clouds(gen, cloud.reshape(1, -1, 3), patch_num_point, patch_num_ratio, final_ratio, return_stages)
    if return_stages:
        return res[0][0].cpu().numpy(), res[1]
    return res[0].cpu().numpy()


def save_xyz(path, points):
    """np.savetxt(path, pred_pc, fmt='%.6f') (model.py:381)."""
    np.savetxt(path, points, fmt="%.6f")
