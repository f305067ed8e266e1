"""Minimal restatement of the timm==0.9.5 surface used by cav_mae_base.py (own code, see README.md)."""
from . import layers, models  # noqa: F401
from .models.vision_transformer import create_model  # noqa: F401
