"""Hot-path hyper-parameters of the three model families.

Values are the ones the reference's YAML files hand to the models as ``param``
(configs/patch_aug_net.yaml:9-54, configs/pptnet_origin.yaml:30-66); only the
keys the descriptor-extraction path reads are kept.  A user's own YAML dict with
the same keys can be passed instead (the model classes index it like the
reference does, e.g. place_recognition/patch_aug_net/models/patch_aug_net.py:23-38).
"""
import copy

PATCH_AUG_NET = {
    "AGGREGATION": "spvlad",
    "AGGREGATION_TYPE": 2,
    "GROUP": 8,
    "NUM_POINTS": 4096,
    "FEATURE_SIZE": [256, 256, 256],
    "MAX_SAMPLES": [128, 1024, 4096],
    "CLUSTER_SIZE": [4, 16, 64],
    "OUTPUT_DIM": [256, 256, 256],
    "USE_ORIGIN_PC_IN_FP": True,
    "GATING": False,
    "SAMPLING": [1024, 128, 16],
    "KNN": [20, 20, 20],
    "KNN_DILATION": 2,
}

PPTNET = {
    "AGGREGATION": "spvlad",
    "GROUP": 8,
    "NUM_POINTS": 4096,
    "FEATURE_SIZE": [256, 256, 256, 256],
    "MAX_SAMPLES": [64, 256, 1024, 4096],
    "CLUSTER_SIZE": [1, 4, 16, 64],
    "OUTPUT_DIM": [256, 256, 256, 256],
    "GATING": True,
    "SAMPLING": [1024, 256, 64, 16],
    "KNN": [20, 20, 20, 20],
}


def patch_aug_net_config():
    return copy.deepcopy(PATCH_AUG_NET)


def pptnet_config():
    return copy.deepcopy(PPTNET)


def scaled_config(cfg, num_points):
    """Shrink a config to a smaller cloud (tests): every level keeps its ratio to NUM_POINTS."""
    cfg = copy.deepcopy(cfg)
    r = cfg["NUM_POINTS"] // num_points
    assert r >= 1 and cfg["NUM_POINTS"] % num_points == 0
    cfg["NUM_POINTS"] = num_points
    cfg["SAMPLING"] = [max(s // r, 4) for s in cfg["SAMPLING"]]
    cfg["MAX_SAMPLES"] = list(reversed(cfg["SAMPLING"][:-1])) + [num_points]
    return cfg
