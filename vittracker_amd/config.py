"""Config surface of the vit_dist tracker.

Mirrors the key tree of the reference's ``lib/config/vit_dist/config.py:7-106`` (defaults)
and the strict YAML merge of ``lib/config/vit_dist/config.py:128-149`` (an unknown key in
the YAML raises ``ValueError("<key> not exist in config.py")``), without the ``easydict``
dependency: the tree is a nested ``Node`` (a dict with attribute access), which is all the
hot path reads (``cfg.MODEL.BACKBONE.CHANNELS`` etc. in ``lib/models/vit_dist/vit_dist.py:159-164``
and ``lib/models/layers/head.py:334-359``).
"""
from __future__ import annotations

import copy

import yaml


class Node(dict):
    """dict with attribute access; nested dicts become Nodes."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Node):
            v = Node(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _defaults() -> Node:
    # Values are the reference defaults (lib/config/vit_dist/config.py:7-106).
    return Node({
        "MODEL": {
            "PRETRAIN_FILE": "mae_pretrain_vit_base.pth",
            "EXTRA_MERGER": False,
            "RETURN_INTER": False,
            "RETURN_STAGES": [],
            "BACKBONE": {
                "TYPE": "vit_base_patch16_224", "STRIDE": 16, "MID_PE": False, "SEP_SEG": False,
                "CAT_MODE": "direct", "MERGE_LAYER": 0, "ADD_CLS_TOKEN": False,
                "CLS_TOKEN_USE_MODE": "ignore", "CHANNELS": 768, "HEADS": 12,
                "CE_LOC": [], "CE_KEEP_RATIO": [], "CE_TEMPLATE_RANGE": "ALL",
            },
            "HEAD": {"TYPE": "CENTER", "NUM_CHANNELS": 256},
        },
        "TRAIN": {
            "LR": 0.0001, "WEIGHT_DECAY": 0.0001, "EPOCH": 500, "LR_DROP_EPOCH": 400,
            "BATCH_SIZE": 16, "NUM_WORKER": 8, "OPTIMIZER": "ADAMW", "BACKBONE_MULTIPLIER": 0.1,
            "GIOU_WEIGHT": 2.0, "L1_WEIGHT": 5.0, "AUX_WEIGHT": 1.0, "AUX_TYPE": "3 output",
            "FREEZE_LAYERS": [0], "PRINT_INTERVAL": 50, "VAL_EPOCH_INTERVAL": 20,
            "GRAD_CLIP_NORM": 0.1, "AMP": False, "TEACHER": "ostrack",
            "CE_START_EPOCH": 20, "CE_WARM_EPOCH": 80, "DROP_PATH_RATE": 0.1,
            "SCHEDULER": {"TYPE": "step", "DECAY_RATE": 0.1},
        },
        "DATA": {
            "SAMPLER_MODE": "causal",
            "MEAN": [0.485, 0.456, 0.406], "STD": [0.229, 0.224, 0.225],
            "MAX_SAMPLE_INTERVAL": 200,
            "TRAIN": {"DATASETS_NAME": ["LASOT", "GOT10K_vottrain"], "DATASETS_RATIO": [1, 1],
                      "SAMPLE_PER_EPOCH": 60000},
            "VAL": {"DATASETS_NAME": ["GOT10K_votval"], "DATASETS_RATIO": [1],
                    "SAMPLE_PER_EPOCH": 10000},
            "SEARCH": {"SIZE": 320, "FACTOR": 5.0, "CENTER_JITTER": 4.5, "SCALE_JITTER": 0.5,
                       "NUMBER": 1},
            "TEMPLATE": {"NUMBER": 1, "SIZE": 128, "FACTOR": 2.0, "CENTER_JITTER": 0,
                         "SCALE_JITTER": 0},
        },
        "TEST": {"TEMPLATE_FACTOR": 2.0, "TEMPLATE_SIZE": 128, "SEARCH_FACTOR": 5.0,
                 "SEARCH_SIZE": 320, "EPOCH": 500},
    })


#: global mutable config, like the reference's module-level ``cfg``
cfg = _defaults()


def fresh_cfg() -> Node:
    """A new default tree (the reference only has the global; tests want isolation)."""
    return _defaults()


def _merge(base: Node, exp: dict) -> None:
    for k, v in exp.items():
        if k not in base:
            raise ValueError("{} not exist in config.py".format(k))
        if isinstance(v, dict):
            if not isinstance(base[k], dict):
                raise ValueError("{} is not a section in config.py".format(k))
            _merge(base[k], v)
        else:
            base[k] = v


def update_config_from_file(filename: str, base_cfg: Node | None = None) -> None:
    """Strict merge of a YAML experiment file (lib/config/vit_dist/config.py:141-149)."""
    with open(filename) as f:
        exp = yaml.safe_load(f) or {}
    _merge(cfg if base_cfg is None else base_cfg, exp)


def geometry(c: Node) -> dict:
    """Derived sizes the hot path needs, in one place.

    feat_sz follows ``build_box_head`` (lib/models/layers/head.py:356: DATA.SEARCH.SIZE / stride);
    token counts follow the stem's 16x down-sampling (lib/models/vit_dist/vit_dist.py:36-54).
    """
    stride = int(c.MODEL.BACKBONE.STRIDE)
    tx = int(c.DATA.SEARCH.SIZE)
    tz = int(c.DATA.TEMPLATE.SIZE)
    return {
        "search_size": tx, "template_size": tz, "stride": stride,
        "feat_sz": tx // stride, "feat_sz_z": tz // stride,
        "len_x": (tx // stride) ** 2, "len_z": (tz // stride) ** 2,
        "channels": int(c.MODEL.BACKBONE.CHANNELS), "heads": int(c.MODEL.BACKBONE.HEADS),
        "head_channels": int(c.MODEL.HEAD.NUM_CHANNELS), "head_type": str(c.MODEL.HEAD.TYPE),
    }
