"""Detector heads, looked up by name from cfg.MODEL.SSDS (reference model_builder.py:16-21)."""
from .ssd import SSD
from .fpn import SSDFPN
from .bifpn import SSDBiFPN
