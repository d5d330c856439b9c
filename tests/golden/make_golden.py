"""Regenerates tests/golden/*.npz.  Run in the authoring container only (needs /root/reference):

    python oracle/build.py && python tests/golden/make_golden.py

ref_*.npz    inputs + outputs of the REFERENCE's own CPU functions (oracle/_ref, compiled from
             /root/reference by oracle/build.py).  These pin the oracle (tests/test_oracle.py) and, on
             the GPU, the HIP kernels in DISPU_ARITH_PLAIN mode bit-for-bit.
oracle_*.npz outputs of the repo's oracle for the paths that have NO CPU implementation in the
             reference (FPS, gather, GPU-flavour approxmatch, GEMM-form k-NN): regression pins of the
             restatement, labelled as such.  Fixtures are data only (KB-sized arrays).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import ref as R  # noqa: E402

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "dis-pu_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print(name, {k: v.shape for k, v in kw.items()})


def point_operation_golden():
    """Outputs of the reference's own Common/point_operation.py (pure numpy, importable here) under a fixed seed."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from Common import point_operation as PO   # noqa: E402  (authoring container only)
    x = synth.patches(3, 64, seed=77).astype(np.float64)
    gt = synth.patches(3, 96, seed=78).astype(np.float64)
    np.random.seed(4242)
    idx = np.array(PO.nonuniform_sampling(1024, 256), np.int64)
    jit = PO.jitter_perturbation_point_cloud(x.copy(), sigma=0.01, clip=0.03)
    rx, rgt = PO.rotate_point_cloud_and_gt(jit.copy(), gt.copy())
    sx, sgt, scales = PO.random_scale_point_cloud_and_gt(rx.copy(), rgt.copy(), scale_low=0.8, scale_high=1.2)
    hx, hgt = PO.shift_point_cloud_and_gt(sx.copy(), sgt.copy(), shift_range=0.3)
    save("ref_point_operation.npz", seed=np.array(4242), x=x, gt=gt, idx=idx, jit=jit, rx=rx, rgt=rgt, sx=sx, sgt=sgt,
         scales=scales, hx=hx, hgt=hgt)


def main():
    assert R.available(), "build oracle/_ref first: python oracle/build.py"
    point_operation_golden()
    rng = np.random.default_rng(20260928)

    # --- nn_distance: reference CPU nnsearch, both directions -------------------------------------
    x1 = synth.patches(3, 200, seed=11)
    x2 = synth.patches(3, 333, seed=12)
    d1, i1 = R.nnsearch(x1, x2)
    d2, i2 = R.nnsearch(x2, x1)
    save("ref_nndistance.npz", xyz1=x1, xyz2=x2, dist1=d1, idx1=i1, dist2=d2, idx2=i2)

    # --- three_nn / three_interpolate (+grad) -------------------------------------------------------
    u = synth.patches(2, 300, seed=13)
    kn = synth.patches(2, 77, seed=14)
    d, i = R.threenn(u, kn)
    pts = rng.standard_normal((2, 77, 20)).astype(np.float32)
    w = rng.random((2, 300, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    out = R.threeinterpolate(pts, i, w)
    go = rng.standard_normal((2, 300, 20)).astype(np.float32)
    gp = R.threeinterpolate_grad(pts, i, w, go)
    save("ref_interpolate.npz", xyz1=u, xyz2=kn, dist=d, idx=i, points=pts, weight=w, out=out, grad_out=go, grad_points=gp)

    # --- query_ball_point / group_point (+grad): reference CPU twins --------------------------------
    q1 = synth.patches(3, 400, seed=15)
    q2 = q1[:, ::3].copy()          # queries are dataset points: every query hits itself
    idx = R.query_ball_point(0.2, 12, q1, q2)
    feat = rng.standard_normal((3, 400, 7)).astype(np.float32)
    grp = R.group_point(feat, idx)
    gg = rng.standard_normal(grp.shape).astype(np.float32)
    gpg = R.group_point_grad(feat, idx, gg)
    save("ref_grouping.npz", xyz1=q1, xyz2=q2, radius=np.float32(0.2), nsample=np.int32(12), idx=idx, points=feat,
         out=grp, grad_out=gg, grad_points=gpg)

    # --- nanoflann k-NN (live forward path): self-query and disjoint query ---------------------------
    s = synth.patches(4, 512, seed=16)
    qs = synth.patches(4, 100, seed=17)
    save("ref_knn.npz", support=s, query=qs, k=np.int32(16), idx_self=R.knn_batch(s, s, 16, omp=True).astype(np.int32),
         idx_query=R.knn_batch(s, qs, 16, omp=False).astype(np.int32))

    # --- approxmatch: reference CPU variant is only a loose bound (11 levels, double) ----------------
    a1 = synth.patches(2, 256, seed=18)
    a2 = synth.patches(2, 256, seed=19)
    m_cpu = R.approxmatch(a1, a2)                       # [b,n,m]
    c_cpu = R.matchcost(a1, a2, m_cpu)
    m_orc = O.approx_match(a1, a2, contract=1)          # [b,m,n]  GPU-flavour restatement
    c_ref_on_orc = R.matchcost(a1, a2, np.ascontiguousarray(m_orc.transpose(0, 2, 1)))
    g1, g2 = R.matchcostgrad(a1, a2, np.ascontiguousarray(m_orc.transpose(0, 2, 1)))
    save("ref_approxmatch.npz", xyz1=a1, xyz2=a2, cost_cpu_variant=c_cpu, oracle_match=m_orc,
         ref_matchcost_on_oracle_match=c_ref_on_orc, ref_grad1_on_oracle_match=g1, ref_grad2_on_oracle_match=g2)

    # --- selection sort known answer (tf_ops/grouping/selection_sort.cpp:65-94, run via oracle/_ref) ---
    dist = (10 - np.arange(16, dtype=np.float32)).reshape(2, 2, 4)
    save("ref_selection_sort.npz", dist=dist, k=np.int32(3),
         out=np.array([7, 8, 9, 10, 3, 4, 5, 6, -1, 0, 1, 2, -5, -4, -3, -2], np.float32).reshape(2, 2, 4),
         outi=np.tile(np.array([3, 2, 1, 0], np.int32), 4).reshape(2, 2, 4))

    # --- oracle-generated pins for GPU-only reference kernels ---------------------------------------
    f = synth.patches(3, 700, seed=20)
    fi = O.farthest_point_sample(96, f, contract=1)
    fi0 = O.farthest_point_sample(96, f, contract=0)
    feat24 = rng.standard_normal((2, 256, 24)).astype(np.float32)
    kd, ki = O.knn_point_2(17, feat24, feat24)
    save("oracle_gpu_only.npz", fps_inp=f, fps_idx_contract=fi, fps_idx_plain=fi0,
         feat=feat24, knn2_dist=kd, knn2_idx=ki[..., 1].astype(np.int32),
         am_cost=O.match_cost(a1, a2, m_orc, contract=1))


def chunked_approxmatch_pin():
    """Round 2: the MI355X EMD kernels associate the auction's sums in chunks of 128 partners (oracle chunk = AM_CHUNK).  This
    freezes that restatement (pinned exp -> bit-reproducible on any IEEE machine) so a later edit of either side is caught.
    Oracle-generated (the CUDA kernel cannot run here); the sequential order stays pinned by ref_approxmatch.npz."""
    a1 = synth.patches(2, 300, seed=31)
    a2 = synth.patches(2, 260, seed=32)
    m = O.approx_match(a1, a2, contract=1, pinned_exp=True, chunk=O.AM_CHUNK)
    save("oracle_approxmatch_chunk128.npz", xyz1=a1, xyz2=a2, chunk=np.int32(O.AM_CHUNK), match_pinned=m,
         cost=O.match_cost(a1, a2, m, contract=1))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "chunked":
        chunked_approxmatch_pin()
    else:
        main()
        chunked_approxmatch_pin()
