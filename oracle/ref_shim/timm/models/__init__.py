from . import layers, vision_transformer  # noqa: F401
