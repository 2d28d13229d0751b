"""Name-lookup namespace mirroring ``nntts.optimizers`` (reference nntts/bin/train.py:186-193).
`Adam` is the fused clip + Adam-amsgrad of efficient_tts_amd.optim; it is constructed from the MODEL
(it re-homes the parameters into one flat buffer), not from `model.parameters()`."""
from .optim import EftsAdam as Adam  # noqa: F401
