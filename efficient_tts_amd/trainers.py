"""Name-lookup namespace mirroring ``nntts.trainers`` (reference nntts/bin/train.py:220)."""
from .trainer import EfficientTTSTrainer  # noqa: F401
