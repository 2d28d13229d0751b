"""Name-lookup namespace mirroring ``nntts.schedulers`` (reference nntts/bin/train.py:194-204)."""
from .optim import WarmupLR  # noqa: F401
