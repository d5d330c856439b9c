"""Variable inventory and initialiser of the generator graph (product side; the oracle keeps its own restatement).

Names are the TF variable scopes of DisPU/generator.py:45,60 + Common/ops.py (what a TF1 checkpoint of the reference
holds under 'generator/...'); shapes follow Common/tf_util.py:87-105,155-176 ([kh, kw, C_in, C_out] conv kernels,
stored here flattened to [kh*kw*C_in, C_out]).  Initialisation = the reference's: Xavier-uniform weights
(tf.contrib.layers.xavier_initializer, tf_util.py:41-45), zero biases (:104-105), BN gamma 1 / beta 0 / moving
mean 0 / moving variance 1."""
import math
from collections import OrderedDict

import numpy as np

GROWTH = 24
DENSE_BLOCKS = 4
BN_SCOPE = "refine/PointShuffle/weight_net/wconv0/bn/"


def layer_shapes():
    """[(scope, kernel shape)] in graph order."""
    L = []
    fe = "generator/feature_extraction_coarse/"
    L.append((fe + "layer0", (1, 1, 3, 24)))
    c_in, width = 24, 24
    for d in range(1, DENSE_BLOCKS + 1):
        if d > 1:
            L.append((fe + "layer%d_prep" % d, (1, width, 2 * GROWTH)))
            c_in = 2 * GROWTH
        L.append((fe + "layer%d/l0" % d, (1, 1, 2 * c_in, GROWTH)))
        L.append((fe + "layer%d/l1" % d, (1, 1, GROWTH + c_in, GROWTH)))
        L.append((fe + "layer%d/l2" % d, (1, 1, 2 * GROWTH + c_in, GROWTH)))
        width += 3 * GROWTH + c_in
    L.append(("generator/upshuffle_0/conv1", (1, 1, width + 2, 256)))
    L.append(("generator/upshuffle_0/conv2", (1, 1, 256, 128)))
    for s, cin in (("generator/coarse_coordinate_regressor/", 128), ("refine/fine_coordinate_regressor/", 256)):
        L.append((s + "fc_layer0", (1, cin, 256)))
        L.append((s + "fc_layer1", (1, 256, 64)))
        L.append((s + "fc_layer2", (1, 64, 3)))
    ps = "refine/PointShuffle/"
    L.append((ps + "PointShuffle/conv_kv", (1, 1, 128, 128)))
    L.append((ps + "PointShuffle/conv_query", (1, 1, 128, 64)))
    L.append((ps + "PointShuffle/conv_back_project", (1, 1, 64, 256)))
    L.append((ps + "skip", (1, 134, 256)))
    L.append((ps + "conv0", (1, 1, 134, 128)))
    L.append((ps + "conv1", (1, 1, 128, 128)))
    L.append((ps + "weight_net/wconv0", (1, 1, 3, 16)))
    L.append((ps + "after_conv", (1, 128, 16, 256)))
    L.append((ps + "aggregation", (1, 256, 256)))
    return L


def init_params(seed=1234):
    """name -> float32 array, the mapping Generator.load_params / Trainer.load_params take."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, shp in layer_shapes():
        recept = int(np.prod(shp[:-2]))
        lim = math.sqrt(6.0 / (recept * shp[-2] + recept * shp[-1]))
        P[name + "/weights"] = rng.uniform(-lim, lim, (int(np.prod(shp[:-1])), shp[-1])).astype(np.float32)
        rng.standard_normal(shp[-1])      # keeps the generator stream aligned with initialisers that also draw biases
        P[name + "/biases"] = np.zeros(shp[-1], np.float32)
    P[BN_SCOPE + "gamma"] = np.ones(16, np.float32)
    P[BN_SCOPE + "beta"] = np.zeros(16, np.float32)
    P[BN_SCOPE + "moving_mean"] = np.zeros(16, np.float32)
    P[BN_SCOPE + "moving_variance"] = np.ones(16, np.float32)
    return P


def num_params(P):
    return int(sum(v.size for k, v in P.items() if k.endswith(("weights", "biases", "gamma", "beta"))))
