"""Name-lookup namespace mirroring ``nntts.models`` (reference nntts/models/__init__.py:1):
``getattr(efficient_tts_amd.models, config["model_name"])(**config["model_params"])``."""
from .model import EfficientTTSCNN  # noqa: F401
