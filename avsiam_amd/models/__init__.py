"""Model entry points, mirroring ``import models; models.CAVMAE_BASE(...)`` of the reference
(/root/reference/src/models/__init__.py:10, consumed at src/run_cavmae_pretrain_base.py:175)."""
from .cav_mae_base import CAVMAE, CAVMAE_BASE  # noqa: F401
from .cav_mae_ft import CAVMAEFT_BASE  # noqa: F401,E402
