"""efficient_tts_amd -- MI355X-native EFTS-CNN acoustic-model hot path.

Drop-in for the hot path of liusongxiang/efficient_tts (``nntts.models.EfficientTTSCNN`` and
the ``EfficientTTSTrainer`` step around it): hand-written HIP/CDNA4 kernels behind the C ABI of
``libefts_hip.so`` (include/efts_abi.h), called from PyTorch-ROCm host code.
"""
from .model import EfficientTTSCNN  # noqa: F401
from . import models  # noqa: F401
from . import trainers  # noqa: F401

__all__ = ["EfficientTTSCNN", "models", "trainers"]
