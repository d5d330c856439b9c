"""Blocks of the reference's Common/ops.py that are compositions of PointNet++ modules, with the
reference's function names, argument order and return values; variables come from `params` (scope -> array, the names a
TF1 checkpoint of the block would hold) instead of TF variable scopes.  The generator's own blocks (feature_extraction_GCN,
duplicate_up, PointShuffle2, coordinate_regressor) live in generator.py as one launch sequence."""
from .pointnet_util import pointnet_fp_module, pointnet_sa_module


def hierachy_feature_extractor(inputs, is_training, bn_decay=None, use_bn=False, scope="hierachy_feature_extractor",
                               npoints=(1024, 384, 128), radius=(0.1, 0.2, 0.4), params=None):
    """Common/ops.py:505-550: PointNet++ encoder / decoder over a cloud [B, N, 3] -> per-point features [B, N, 128].

    Three set-abstraction levels (FPS to npoints[i] centres, ball query of radius[i] with 64 samples, MLPs [32,32,64] /
    [64,64,128] / [128,128,256], max over the samples), one group-all level ([256,256,512]), four feature-propagation levels
    (3-NN inverse-distance interpolation + MLPs [512,512] / [512,256] / [256,128] / [128,128,128]).  As in the reference the
    modules run with their default bn=True whatever `use_bn` says (ops.py:514-540 never forwards it) and the variable scopes are
    'layer1'..'layer4', 'fa_layer1'..'fa_layer4' (no outer scope: `scope` is unused there too).  Each of the three sampled levels
    is one fused launch after its FPS / ball query (csrc/sa_fused.hip)."""
    l0_xyz, l0_points = inputs, None
    l1_xyz, l1_points, _ = pointnet_sa_module(l0_xyz, l0_points, npoints[0], radius[0], 64, [32, 32, 64], None, False, is_training,
                                              bn_decay, "layer1", params=params)
    l2_xyz, l2_points, _ = pointnet_sa_module(l1_xyz, l1_points, npoints[1], radius[1], 64, [64, 64, 128], None, False, is_training,
                                              bn_decay, "layer2", params=params)
    l3_xyz, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoints[2], radius[2], 64, [128, 128, 256], None, False, is_training,
                                              bn_decay, "layer3", params=params)
    l4_xyz, l4_points, _ = pointnet_sa_module(l3_xyz, l3_points, None, None, None, [256, 256, 512], None, True, is_training,
                                              bn_decay, "layer4", params=params)
    l3_points = pointnet_fp_module(l3_xyz, l4_xyz, l3_points, l4_points, [512, 512], is_training, bn_decay, "fa_layer1", params=params)
    l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [512, 256], is_training, bn_decay, "fa_layer2", params=params)
    l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [256, 128], is_training, bn_decay, "fa_layer3", params=params)
    l0_points = pointnet_fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [128, 128, 128], is_training, bn_decay, "fa_layer4", params=params)
    return l0_points


def hierachy_feature_extractor_variables():
    """[(scope, C_in, C_out)] of the block above in graph order (each with weights, biases and the bn/ quartet)."""
    spec = []
    cin = 3
    for lvl, mlp in (("layer1", [32, 32, 64]), ("layer2", [64, 64, 128]), ("layer3", [128, 128, 256]), ("layer4", [256, 256, 512])):
        for i, co in enumerate(mlp):
            spec.append(("%s/conv%d" % (lvl, i), cin, co))
            cin = co
        cin = 3 + mlp[-1]
    for lvl, c0, mlp in (("fa_layer1", 512 + 256, [512, 512]), ("fa_layer2", 512 + 128, [512, 256]), ("fa_layer3", 256 + 64, [256, 128]),
                         ("fa_layer4", 128, [128, 128, 128])):
        cin = c0
        for i, co in enumerate(mlp):
            spec.append(("%s/conv_%d" % (lvl, i), cin, co))
            cin = co
    return spec
