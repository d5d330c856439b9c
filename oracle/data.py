"""CPU ORACLE (test infrastructure) for the data path and the evaluator.

Restates, with the SAME numpy global-RNG call sequence (so a seeded run reproduces the reference's batches):
  Common/point_operation.py: nonuniform_sampling :10-18, rotate_point_cloud_and_gt :32-71 (z_rotated=True ->
  Rz only, but all three angles are drawn), jitter_perturbation_point_cloud :73-85, shift_point_cloud_and_gt
  :87-104, random_scale_point_cloud_and_gt :107-123;
  DisPU/dataset.py: normalize_point_cloud :26-40, Fetcher.next_batch :118-143 (including its off-by-one: batch_idx
  is incremented BEFORE the slice, so the first batch of an epoch is skipped and the last one may be short/empty);
  evaluate.py:33-41,153-162: normalise both clouds, nn_distance, CD = mean fwd + mean bwd, HD = max fwd + max bwd.
Pinned against golden vectors produced by importing the reference's point_operation.py in the authoring container
(tests/golden/ref_point_operation.npz, tests/golden/make_golden.py).  Only tests/ import this module."""
import numpy as np


def nonuniform_sampling(num=4096, sample_num=1024):
    sample = set()
    loc = np.random.rand() * 0.8 + 0.1
    while len(sample) < sample_num:
        a = int(np.random.normal(loc=loc, scale=0.3) * num)
        if a < 0 or a >= num:
            continue
        sample.add(a)
    return list(sample)


def rotation_z(angles):
    c, s = np.cos(angles[2]), np.sin(angles[2])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rotate_point_cloud_and_gt(batch_data, batch_gt=None):
    for k in range(batch_data.shape[0]):
        angles = np.random.uniform(size=(3)) * 2 * np.pi
        R = rotation_z(angles)
        batch_data[k, ..., 0:3] = np.dot(batch_data[k, ..., 0:3].reshape((-1, 3)), R)
        if batch_gt is not None:
            batch_gt[k, ..., 0:3] = np.dot(batch_gt[k, ..., 0:3].reshape((-1, 3)), R)
    return batch_data, batch_gt


def jitter_perturbation_point_cloud(batch_data, sigma=0.005, clip=0.02):
    B, N, C = batch_data.shape
    jittered = np.clip(sigma * np.random.randn(B, N, C), -1 * clip, clip)
    jittered[:, :, 3:] = 0
    jittered += batch_data
    return jittered


def shift_point_cloud_and_gt(batch_data, batch_gt=None, shift_range=0.3):
    B = batch_data.shape[0]
    shifts = np.random.uniform(-shift_range, shift_range, (B, 3))
    for b in range(B):
        batch_data[b, :, 0:3] += shifts[b, 0:3]
    if batch_gt is not None:
        for b in range(B):
            batch_gt[b, :, 0:3] += shifts[b, 0:3]
    return batch_data, batch_gt


def random_scale_point_cloud_and_gt(batch_data, batch_gt=None, scale_low=0.5, scale_high=2):
    B = batch_data.shape[0]
    scales = np.random.uniform(scale_low, scale_high, B)
    for b in range(B):
        batch_data[b, :, 0:3] *= scales[b]
    if batch_gt is not None:
        for b in range(B):
            batch_gt[b, :, 0:3] *= scales[b]
    return batch_data, batch_gt, scales


def normalize_point_cloud(inputs):
    centroid = np.mean(inputs, axis=1, keepdims=True)
    pc = inputs - centroid
    furthest = np.amax(np.sqrt(np.sum(pc ** 2, axis=-1, keepdims=True)), axis=1, keepdims=True)
    return pc / furthest, centroid, furthest


class Fetcher(object):
    """DisPU/dataset.py:81-143 on in-memory arrays (the HDF5 read, :52-78, is `input = gt = poisson_<out_num>` when
    opts.random, then normalisation by the ground truth's centroid / furthest distance)."""

    def __init__(self, gt_patches, batch_size, patch_num_point=256, augment=True, shuffle=True, random=True,
                 jitter_sigma=0.01, jitter_max=0.03):
        gt, centroid, furthest = normalize_point_cloud(np.asarray(gt_patches))
        self.input_data = (np.asarray(gt_patches) - centroid) / furthest
        self.gt_data = gt
        self.radius_data = np.ones(shape=(len(gt)))
        self.batch_size, self.patch_num_point = batch_size, patch_num_point
        self.length = self.input_data.shape[0]
        self.augment, self.shuffle, self.random = augment, shuffle, random
        self.jitter_sigma, self.jitter_max = jitter_sigma, jitter_max
        self.reset()

    def reset(self):
        self.idxs = np.arange(0, self.length)
        if self.shuffle:
            np.random.shuffle(self.idxs)
            self.input_data = self.input_data[self.idxs]
            self.gt_data = self.gt_data[self.idxs]
        self.num_batches = (self.length + self.batch_size - 1) // self.batch_size
        self.batch_idx = 0

    def has_next_batch(self):
        return self.batch_idx < self.num_batches

    def next_batch(self):
        self.batch_idx += 1
        start = self.batch_idx * self.batch_size
        end = min((self.batch_idx + 1) * self.batch_size, self.length)
        x = self.input_data[start:end, :, :].copy()
        gt = self.gt_data[start:end, :, :].copy()
        radius = self.radius_data[start:end].copy()
        if self.random:
            new = np.zeros((self.batch_size, self.patch_num_point, x.shape[2]))
            for i in range(self.batch_size):
                idx = nonuniform_sampling(self.input_data.shape[1], sample_num=self.patch_num_point)
                new[i, ...] = x[i][idx]
            x = new
        if self.augment:
            x = jitter_perturbation_point_cloud(x, sigma=self.jitter_sigma, clip=self.jitter_max)
            x, gt = rotate_point_cloud_and_gt(x, gt)
            x, gt, _ = random_scale_point_cloud_and_gt(x, gt, scale_low=0.8, scale_high=1.2)
        return x, gt, radius


def evaluate_pair(pred, gt):
    """evaluate.py:33-41,153-162 -> (CD, hausdorff); float64 brute force."""
    p = normalize_point_cloud(np.asarray(pred, np.float64)[None])[0][0]
    g = normalize_point_cloud(np.asarray(gt, np.float64)[None])[0][0]
    fwd = np.empty(len(p))
    bwd = np.full(len(g), np.inf)
    for a in range(0, len(p), 512):                                   # row blocks keep the distance matrix small
        d = ((p[a:a + 512, None, :] - g[None, :, :]) ** 2).sum(-1)
        fwd[a:a + 512] = d.min(1)
        bwd = np.minimum(bwd, d.min(0))
    return float(fwd.mean() + bwd.mean()), float(fwd.max() + bwd.max())
