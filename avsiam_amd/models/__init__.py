"""Model entry points, mirroring ``import models; models.CAVMAE_BASE(...)`` of the reference
(/root/reference/src/models/__init__.py:8-13, consumed at src/run_cavmae_pretrain_base.py:175): CAVMAE / CAVMAE_BASE, and the larger
skeletons the reference exports by name - CAVMAE_LARGE (:9), CAVMAE_HUGE (:13) - with their fine-tuned inference classes."""
from .cav_mae_base import CAVMAE, CAVMAE_BASE, CAVMAE_HUGE, CAVMAE_LARGE  # noqa: F401
from .cav_mae_ft import CAVMAEFT, CAVMAEFT_BASE, CAVMAEFT_HUGE, CAVMAEFT_LARGE  # noqa: F401,E402
