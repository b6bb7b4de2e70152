"""Backbones, looked up by name from cfg.MODEL.NETS (reference model_builder.py:21).  Only the families
named by BASELINE.json's configs are part of the MI355X hot path:
MobileNet v1/v2, ResNet / ResNeXt, RegNetX."""
from .resnet import *  # noqa: F401,F403
from .mobilenet import *  # noqa: F401,F403
from .regnet import *  # noqa: F401,F403
