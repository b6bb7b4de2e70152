"""Yaml config system -- the plugin contract of the reference (ssds/core/config.py): one global
``cfg`` tree with defaults, ``cfg_from_file(path)`` merges a yaml file with strict key and type
checking, derived fields are recomputed afterwards.

Keys, defaults and error behaviour follow the reference schema (config.py:35-230, 233-257, 260-285,
321-346) so existing ``experiments/cfgs/*.yml`` load unchanged; the implementation is a table-driven
rewrite (defaults live in one nested literal).
"""
import os.path as osp
from ast import literal_eval

import numpy as np


class AttrDict(dict):
    """dict with attribute access (reference config.py:19-32)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _tree(d):
    return AttrDict((k, _tree(v) if isinstance(v, dict) else v) for k, v in d.items())


_ROOT_DIR = osp.abspath(osp.join(osp.dirname(__file__), "..", ".."))

_DEFAULTS = {
    "MODEL": {
        "NETS": "vgg16",
        "SSDS": "ssd",
        "HALF_PRECISION": True,
        "IMAGE_SIZE": [300, 300],
        "NUM_IMAGES": 1,
        "NUM_CLASSES": 21,
        "FEATURE_LAYER": [[22, 34, "S", "S", "", ""], [512, 1024, 512, 256, 256, 256]],
        "STEPS": [],
        "SIZES": [0.2, 0.95],
        "ASPECT_RATIOS": [[2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
        "CLIP": True,
        "NUM_FUSED": 3,
    },
    "TRAIN": {
        "CHECKPOINTS_KEPT": 10,
        "CHECKPOINTS_EPOCHS": 5,
        "MAX_EPOCHS": 300,
        "BATCH_SIZE": 128,
        "TRAINABLE_SCOPE": "base,extras,norm,loc,conf",
        "RESUME_SCOPE": "",
        "CRITERION": "",
        "OPTIMIZER": {
            "OPTIMIZER": "sgd",
            "LEARNING_RATE": 0.001,
            "DIFFERENTIAL_LEARNING_RATE": [],
            "MOMENTUM": 0.9,
            "MOMENTUM_2": 0.99,
            "EPS": 1e-8,
            "WEIGHT_DECAY": 0.0001,
        },
        "LR_SCHEDULER": {
            "SCHEDULER": "step",
            "STEPS": [1],
            "GAMMA": 0.98,
            "LR_MIN": 0.0,
            "WARM_UP_EPOCHS": 0,
            "MAX_EPOCHS": 300,
        },
    },
    "TEST": {"BATCH_SIZE": 128, "TEST_SCOPE": [0, 300]},
    "MATCHER": {
        "NUM_CLASSES": 21,
        "CLASSIFY_LOSS": "FocalLoss",
        "LOCATE_LOSS": "SmoothL1Loss",
        "BACKGROUND_LABEL": 0,
        "MATCH_THRESHOLD": [0.5, 0.4],
        "CENTER_SAMPLING_RADIUS": 0.0,
        "FOCAL_ALPHA": 0.25,
        "FOCAL_GAMMA": 2,
        "NEGPOS_RATIO": 3,
        "VARIANCE": [0.1, 0.2],
    },
    "POST_PROCESS": {
        "NUM_CLASSES": 21,
        "BACKGROUND_LABEL": 0,
        "SCORE_THRESHOLD": 0.01,
        "IOU_THRESHOLD": 0.6,
        "MAX_DETECTIONS": 100,
        "MAX_DETECTIONS_PER_LEVEL": 300,
        "USE_DIOU": True,
        "RESCORE_CENTER": True,
        "VARIANCE": [0.1, 0.2],
    },
    "ROOT_DIR": _ROOT_DIR,
    "DATASET": {
        "DATASET": "",
        "DATASET_DIR": "",
        "TRAIN_SETS": [],
        "TEST_SETS": [],
        "PICKLE": False,
        "IMAGE_SIZE": [300, 300],
        "TRAIN_BATCH_SIZE": 128,
        "TEST_BATCH_SIZE": 128,
        "NUM_WORKERS": 8,
        "DEVICE_ID": [],
        "PREPROC": {
            "MEAN": 0,
            "STD": 255,
            "CROP_SCALE": [0.3, 1.0],
            "CROP_ASPECT_RATIO": [0.5, 2.0],
            "CROP_ATTEMPTS": 50,
            "HUE_DELTA": 9,
            "BRI_DELTA": 16,
            "CONTRAST_RANGE": [0.75, 1.25],
            "SATURATION_RANGE": [0.75, 1.25],
            "MAX_EXPAND_RATIO": 2.0,
        },
        "MULTISCALE": [],
    },
    "EXP_DIR": osp.abspath(osp.join(_ROOT_DIR, "experiments/models/")),
    "LOG_DIR": osp.abspath(osp.join(_ROOT_DIR, "experiments/models/")),
    "RESUME_CHECKPOINT": "",
    "CHECKPOINTS_PREFIX": "ssd_vgg16_",
    "PHASE": ["train", "eval", "test"],
    "DEVICE_ID": [],
}

__C = _tree(_DEFAULTS)
cfg = __C


def reset_cfg():
    """Restore the defaults (the reference keeps one mutable global; tests need a clean slate)."""
    __C.clear()
    __C.update(_tree(_DEFAULTS))
    return __C


def _decode_cfg_value(v):
    """yaml scalars that are strings may encode python literals (reference config.py:288-318)."""
    if isinstance(v, dict):
        return _tree(v)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _check_and_coerce_cfg_value_type(value_a, value_b, key, full_key):
    """Type of the override must match the default, modulo ndarray / str / tuple<->list
    (reference config.py:321-346)."""
    type_b, type_a = type(value_b), type(value_a)
    if type_a is type_b:
        return value_a
    if isinstance(value_b, np.ndarray):
        return np.array(value_a, dtype=value_b.dtype)
    if isinstance(value_b, str):
        return str(value_a)
    if isinstance(value_a, tuple) and isinstance(value_b, list):
        return list(value_a)
    if isinstance(value_a, list) and isinstance(value_b, tuple):
        return tuple(value_a)
    raise ValueError(
        "Type mismatch ({} vs. {}) with values ({} vs. {}) for config "
        "key: {}".format(type_b, type_a, value_b, value_a, full_key)
    )


def _merge_a_into_b(a, b, stack=None):
    """Merge ``a`` into ``b``; unknown keys raise KeyError (reference config.py:233-257)."""
    assert isinstance(a, AttrDict), "Argument `a` must be an AttrDict"
    assert isinstance(b, AttrDict), "Argument `b` must be an AttrDict"
    for k, v_ in a.items():
        full_key = ".".join(stack) + "." + k if stack is not None else k
        if k not in b:
            raise KeyError("Non-existent config key: {}".format(full_key))
        v = _check_and_coerce_cfg_value_type(_decode_cfg_value(v_), b[k], k, full_key)
        if isinstance(v, AttrDict):
            _merge_a_into_b(v, b[k], stack=[k] if stack is None else stack + [k])
        else:
            b[k] = v


def update_cfg():
    """Derived fields (reference config.py:260-273)."""
    __C.TRAIN.LR_SCHEDULER.MAX_EPOCHS = __C.TRAIN.MAX_EPOCHS - __C.TRAIN.LR_SCHEDULER.WARM_UP_EPOCHS
    __C.DATASET.IMAGE_SIZE = __C.MODEL.IMAGE_SIZE
    __C.DATASET.TRAIN_BATCH_SIZE = __C.TRAIN.BATCH_SIZE
    __C.DATASET.TEST_BATCH_SIZE = __C.TEST.BATCH_SIZE
    __C.MATCHER.NUM_CLASSES = __C.MODEL.NUM_CLASSES
    __C.POST_PROCESS.NUM_CLASSES = __C.MODEL.NUM_CLASSES
    __C.POST_PROCESS.BACKGROUND_LABEL = __C.MATCHER.BACKGROUND_LABEL
    __C.POST_PROCESS.VARIANCE = __C.MATCHER.VARIANCE
    __C.CHECKPOINTS_PREFIX = "{}_{}_{}".format(__C.MODEL.SSDS, __C.MODEL.NETS, __C.DATASET.DATASET)


def cfg_from_file(filename):
    """Load a yaml file and merge it into the global config (reference config.py:276-285)."""
    import yaml

    with open(filename, "r") as f:
        yaml_cfg = _tree(yaml.safe_load(f))
    _merge_a_into_b(yaml_cfg, __C)
    update_cfg()
    return cfg
