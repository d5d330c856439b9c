"""Training data path: counterpart of DisPU/dataset.py (Fetcher :81-143, load_h5_data :52-78, normalize_point_cloud
:26-40) and of the augmentations of Common/point_operation.py it calls.

Same class / method names and the same numpy global-RNG call sequence as the reference (a run seeded with
np.random.seed reproduces the reference's batch order, sub-sampling, jitter, angles and scales -- including
next_batch's off-by-one, which skips the first batch of every epoch).  What differs: the patch arrays live in HBM
and every batch is produced by two device launches (row gather of the 256-of-1024 sub-sample through
dispu_group_point, then dispu_augment), returning device tensors ready for Trainer.train_step; the reference does
this in numpy on a background thread and feeds the result through a TF placeholder.
"""
import numpy as np
import torch

from . import _lib


def nonuniform_sampling(num=4096, sample_num=1024):
    """Common/point_operation.py:10-18 (host RNG logic; index list only)."""
    sample = set()
    loc = np.random.rand() * 0.8 + 0.1
    while len(sample) < sample_num:
        a = int(np.random.normal(loc=loc, scale=0.3) * num)
        if a < 0 or a >= num:
            continue
        sample.add(a)
    return list(sample)


def load_patches(path, in_num=256, out_num=1024, random=True):
    """load_h5_data (dataset.py:52-78): returns (input, gt) arrays [n, P, 3].  HDF5 files are read through the HDF5 C
    library (dispu_amd.h5: ctypes binding, same library h5py wraps); .npz / .npy files with the same dataset names
    ('poisson_<num>') are read directly."""
    if path.endswith((".h5", ".hdf5")):
        from . import h5
        with h5.File(path) as f:
            gt = f["poisson_%d" % out_num]
            inp = gt if random else f["poisson_%d" % in_num]
    elif path.endswith(".npz"):
        z = np.load(path)
        gt = z["poisson_%d" % out_num]
        inp = gt if random else z["poisson_%d" % in_num]
    else:
        gt = np.load(path)
        inp = gt
    assert len(inp) == len(gt)
    return np.asarray(inp, np.float32), np.asarray(gt, np.float32)


class Fetcher(object):
    """Fetcher(opts-like values, arrays) with reset() / has_next_batch() / next_batch() -> (input[B,256,3],
    gt[B,1024,3], radius[B]) device tensors."""

    def __init__(self, input_patches, gt_patches, batch_size, patch_num_point=256, augment=True, shuffle=True, random=True,
                 jitter_sigma=0.01, jitter_max=0.03, device=None):
        self.device = torch.device(device if device is not None else "cuda:0")
        gt = np.asarray(gt_patches)
        inp = np.asarray(input_patches)
        # load_h5_data (dataset.py:70-74): one-time host preprocessing in the arrays' own dtype, same expression order
        # as the reference; both sets are normalised by the GROUND TRUTH's centroid / furthest distance
        centroid = np.mean(gt, axis=1, keepdims=True)
        pc = gt - centroid
        furthest = np.amax(np.sqrt(np.sum(pc ** 2, axis=-1, keepdims=True)), axis=1, keepdims=True)
        inp = inp - centroid
        self._input_host = np.ascontiguousarray(inp / furthest, np.float32)
        self._gt_host = np.ascontiguousarray(pc / furthest, np.float32)
        self.batch_size, self.patch_num_point = int(batch_size), int(patch_num_point)
        self.length = self._input_host.shape[0]
        self.augment, self.shuffle, self.random = augment, shuffle, random
        self.jitter_sigma, self.jitter_max = jitter_sigma, jitter_max
        self.reset()

    def __len__(self):
        return self.length

    def reset(self):
        self.idxs = np.arange(0, self.length)
        if self.shuffle:
            np.random.shuffle(self.idxs)
            self._input_host = self._input_host[self.idxs]
            self._gt_host = self._gt_host[self.idxs]
        self.input_data = torch.from_numpy(self._input_host).to(self.device)     # resident in HBM for the epoch
        self.gt_data = torch.from_numpy(self._gt_host).to(self.device)
        self.num_batches = (self.length + self.batch_size - 1) // self.batch_size
        self.batch_idx = 0

    def has_next_batch(self):
        return self.batch_idx < self.num_batches

    def next_batch(self):
        """dataset.py:118-143, same RNG draws in the same order."""
        L = _lib.lib()
        dev = self.device
        st = _lib.stream_ptr(dev)
        self.batch_idx += 1
        start = self.batch_idx * self.batch_size
        end = min((self.batch_idx + 1) * self.batch_size, self.length)
        bsize = max(end - start, 0)
        x = self.input_data[start:end].contiguous()
        gt = self.gt_data[start:end].contiguous()
        radius = torch.ones(bsize, dtype=torch.float32, device=dev)
        if self.random:
            if bsize != self.batch_size:
                raise IndexError("short batch (%d of %d): the reference fails here too (dataset.py:131-134 indexes "
                                 "batch_input_data[i] for i < batch_size)" % (bsize, self.batch_size))
            idx = np.stack([np.asarray(nonuniform_sampling(self.input_data.shape[1], sample_num=self.patch_num_point), np.int32)
                            for _ in range(self.batch_size)])
            didx = torch.from_numpy(idx).to(dev).view(bsize, self.patch_num_point, 1)
            sub = torch.empty((bsize, self.patch_num_point, 1, 3), dtype=torch.float32, device=dev)
            _lib.check(L.dispu_group_point(bsize, x.shape[1], 3, self.patch_num_point, 1, _lib.ptr(x), _lib.ptr(didx), _lib.ptr(sub), st),
                       "dispu_group_point")
            x = sub.view(bsize, self.patch_num_point, 3)
        if self.augment and bsize:
            n_in = x.shape[1]
            noise = np.clip(self.jitter_sigma * np.random.randn(bsize, n_in, 3), -1 * self.jitter_max, self.jitter_max)
            rot = np.empty((bsize, 3, 3))
            for k in range(bsize):
                angles = np.random.uniform(size=(3)) * 2 * np.pi           # three angles drawn, z rotation used (z_rotated=True)
                c, s = np.cos(angles[2]), np.sin(angles[2])
                rot[k] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
            scales = np.random.uniform(0.8, 1.2, bsize)
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
            dn, dr, ds = d(noise), d(rot.reshape(bsize, 9)), d(scales)
            xo, go = torch.empty_like(x), torch.empty_like(gt)
            _lib.check(L.dispu_augment(bsize, n_in, _lib.ptr(x), _lib.ptr(dn), _lib.ptr(dr), _lib.ptr(ds), None, _lib.ptr(xo), st), "dispu_augment")
            _lib.check(L.dispu_augment(bsize, gt.shape[1], _lib.ptr(gt), None, _lib.ptr(dr), _lib.ptr(ds), None, _lib.ptr(go), st), "dispu_augment")
            x, gt = xo, go
        return x, gt, radius
