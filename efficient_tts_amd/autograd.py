"""torch.autograd bridge: `loss.backward()` on the value returned by EfficientTTSCNN.forward()
(the reference contract, nntts/trainers/efficient_tts_trainer.py:152-153) runs the hand-written HIP
backward.  The fused engine computes forward and backward in one pass; the autograd Function hands
the per-parameter gradients (views of the engine's flat buffer) to the parameters' `.grad` when
backward is called."""
from __future__ import annotations

import torch

from . import lib as L, ops as O
from .model import LazyStats
from .train import TrainEngine


def engine_of(model, params=None) -> TrainEngine:
    """the model's training engine; `params` = tuple(model.parameters()) when the caller already walked the module tree (the
    per-step path: one walk per step instead of one per check)"""
    eng = getattr(model, "_engine", None)
    if eng is not None and params is not None and len(params) == len(eng.params) and params[0].device == eng.dev \
            and all(a is b for a, b in zip(params, eng.params)):
        return eng                                           # same Parameter objects on the same device: the layout still holds
    if eng is None or eng.dev != next(model.parameters()).device or eng.stale(model):
        if eng is not None and getattr(eng, "bound", None):
            # a fused optimizer / bucket reducer holds the OLD engine's flat gradient buffer: rebuilding silently would
            # leave it stepping on an orphaned zero gradient
            raise RuntimeError("the model's parameters changed identity (weight norm applied / removed, .to(), new parameters) after "
                               f"{', '.join(eng.bound)} was bound to its training engine: rebuild the optimizer / DistributedEFTS wrapper")
        eng = TrainEngine(model)
        object.__setattr__(model, "_engine", eng)
    return eng


class _FusedStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, text, text_lengths, speech, speech_lengths, *params):
        eng = engine_of(model, params)
        eng.step_params = params
        out3, _ = eng.forward_backward(text, text_lengths, speech, speech_lengths)
        eng.step_params = None
        ctx.eng = eng
        ctx.named = eng.named                                # (name, Parameter) in model.parameters() order, as `params`
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g_loss, _g_out3):
        eng = ctx.eng
        if eng.join_reduce is not None:                      # data-parallel: the bucketed all-reduce launched during the
            eng.join_reduce()                                # fused pass works in place on `flat`; join it before scaling
        # d(loss) scaling (1.0 in the reference loop: the kernel returns at once, no pass over the buffer)
        gl = g_loss.detach().to(device=eng.flat.device, dtype=torch.float32).reshape(1)
        with O.stream_scope():
            L.check(L.load().efts_scale_unless_one(eng.flat.data_ptr(), eng.numel, gl.data_ptr(), O._stream()), "efts_scale_unless_one")
        # The per-parameter gradients ARE views of the engine's flat buffer (what the fused clip + Adam and
        # the bucketed all-reduce work on).  They are attached as `.grad` directly: handing them to
        # autograd's AccumulateGrad would clone all ~75 of them every step.  Consequence: gradients do
        # not accumulate across backward() calls (the reference loop zero_grads every step,
        # efficient_tts_trainer.py:150-156).
        for n, p in ctx.named:
            p.grad = eng.g[n]
        return (None, None, None, None, None) + (None,) * len(ctx.named)


def training_forward(model, text, text_lengths, speech, speech_lengths):
    params = tuple(model.parameters())
    loss, out3 = _FusedStep.apply(model, text, text_lengths, speech, speech_lengths, *params)
    ws = model._workspace(("train", text.shape[0], text.shape[1], speech.shape[1]), text.device)
    rs2 = ws.bufs[("f", "Tmel_pred", text.shape[0], speech.shape[1], model.odim)]
    imv = ws.bufs[("t", "Timv", (text.shape[0], speech.shape[1]), torch.float32)]
    ralpha = ws.bufs[("t", "Tralpha", (text.shape[0], text.shape[1], speech.shape[1]), torch.float32)]
    return loss, LazyStats(out3), imv.clone(), ralpha.clone(), rs2.view().clone(), speech
