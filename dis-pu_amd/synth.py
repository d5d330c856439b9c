"""Synthetic inputs of SURVEY.md section 8(d): spherical-cap patches with the reference's jitter and
patch normalisation (Common/point_operation.py:74-86 sigma = 0.005 jitter; Common/pc_util.py:147-161
centroid / max-radius normalisation).  Pure numpy; deterministic in (seed)."""
import numpy as np


def cap_patch(rng, npoint, jitter=0.005):
    """npoint points on the z > median half of the unit sphere (a 2-manifold patch), jittered."""
    pts = np.empty((0, 3), np.float64)
    while pts.shape[0] < npoint:
        g = rng.standard_normal((4 * npoint, 3))
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        g = g[g[:, 2] > 0.5]
        pts = np.concatenate([pts, g], axis=0)
    pts = pts[:npoint]
    pts = pts + rng.normal(0.0, jitter, pts.shape)
    return pts


def normalize(pts):
    """pc_util.normalize_point_cloud: subtract centroid, divide by the furthest distance."""
    c = pts.mean(axis=0, keepdims=True)
    p = pts - c
    r = np.sqrt((p ** 2).sum(axis=1)).max()
    return p / r, c, r


def patches(batch, npoint=256, seed=0):
    """[batch, npoint, 3] float32 normalised patches."""
    rng = np.random.default_rng(seed)
    out = np.empty((batch, npoint, 3), np.float32)
    for i in range(batch):
        out[i] = normalize(cap_patch(rng, npoint))[0].astype(np.float32)
    return out


def patch_with_gt(batch, npoint=256, ngt=1024, seed=0):
    """(input [batch,npoint,3], ground truth [batch,ngt,3]) drawn from the same cap, same normalisation."""
    rng = np.random.default_rng(seed)
    x = np.empty((batch, npoint, 3), np.float32)
    g = np.empty((batch, ngt, 3), np.float32)
    for i in range(batch):
        p, c, r = normalize(cap_patch(rng, npoint))
        x[i] = p.astype(np.float32)
        g[i] = ((cap_patch(rng, ngt) - c) / r).astype(np.float32)
    return x, g
