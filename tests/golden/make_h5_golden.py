#!/usr/bin/env python3
"""Writes tests/golden/patches_small.h5 with the genuine HDF5 tools (h5import of HDF5 1.10.6, /opt/conda/bin) -- the file
format fixture for dispu_amd.h5 / dataset.load_patches.  Layout mirrors the reference's training file
(DisPU/dataset.py:52-78): datasets 'poisson_256' [n, 256, 3] and 'poisson_1024' [n, 1024, 3], float32.
'poisson_1024' is stored chunked + gzip (how the published file is stored), 'poisson_256' contiguous big-endian
(exercises the library's byte-order conversion), 'labels' int32.  Values come from a seeded generator that the test
re-creates:  numpy default_rng(20260928): gt = rng.random((4, 1024, 3), float32) - 0.5;  inp = gt[:, ::4];  labels = arange(4).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

H5IMPORT = os.environ.get("H5IMPORT", "/opt/conda/bin/h5import")


def arrays():
    rng = np.random.default_rng(20260928)
    gt = rng.random((4, 1024, 3), dtype=np.float32) - np.float32(0.5)
    inp = np.ascontiguousarray(gt[:, ::4])
    return inp, gt, np.arange(4, dtype=np.int32)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "patches_small.h5")
    inp, gt, labels = arrays()
    with tempfile.TemporaryDirectory() as td:
        def spec(name, a, extra):
            a.tofile(os.path.join(td, name + ".bin"))
            cls = "FP" if a.dtype.kind == "f" else "IN"
            with open(os.path.join(td, name + ".cfg"), "w") as f:
                f.write("PATH %s\nINPUT-CLASS %s\nINPUT-SIZE %d\nRANK %d\nDIMENSION-SIZES %s\nOUTPUT-CLASS %s\nOUTPUT-SIZE %d\n%s"
                        % (name, cls, a.itemsize * 8, a.ndim, " ".join(str(d) for d in a.shape), cls, a.itemsize * 8, extra))
            return [os.path.join(td, name + ".bin"), "-c", os.path.join(td, name + ".cfg")]
        cmd = [H5IMPORT]
        cmd += spec("poisson_1024", gt, "CHUNKED-DIMENSION-SIZES 1 1024 3\nCOMPRESSION-TYPE GZIP\nCOMPRESSION-PARAM 6\n")
        cmd += spec("poisson_256", inp, "OUTPUT-BYTE-ORDER BE\n")
        cmd += spec("labels", labels, "")
        cmd += ["-o", out]
        subprocess.run(cmd, check=True)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
