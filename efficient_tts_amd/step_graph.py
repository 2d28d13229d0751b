"""One optimisation step -- forward, backward, clip, Adam-amsgrad (nntts/trainers/efficient_tts_trainer.py:139-160) -- captured once per
batch shape and replayed as ONE hipGraph.

Why: the step is ~190 launches.  Issued eagerly the host needs 1.9 ms for them on an idle box against 3.6 ms of device time, but
4.5-5.7 ms on a loaded host (observed on the measurement pool), where the HOST then sets the step time.  A replay costs the host a
few tens of microseconds.  What changes from step to step lives in device memory, refreshed by one tiny stream-ordered launch in
front of every replay (efts_store_words): the learning rate and Adam's two bias corrections (efts_adam_amsgrad_dev) and the step
word of the duration predictor's Dropout seeds (`drop_seed_add` of efts_layernorm_rows / _dot / _bwd).  Parameters, optimizer
state and every result are bit-identical to the eager loop's (tests/test_gpu_train.py).

Data parallel (round 4): `GraphedStep(DistributedEFTS(model), ...)` captures the three bucket collectives too.  The reducer issues
them as stream-ordered synchronous ops on its communication stream (dist.py), which is a plain fork of the capturing stream joined
back in front of the optimizer launch -- the only dependency shape ROCm 7.2's capture handles (tools/attic/gpu_probe_capture4.py) -- so
a replay contains forward, backward, the RCCL exchanges overlapped with the backward, clip and Adam, and an 8-process host issues
one launch per step instead of ~190.

Not captured (the step then runs eagerly, same results): conv / prenet Dropout (dropout_rate > 0: their seeds are by-value launch
arguments), optimizers other than EftsAdam."""
from __future__ import annotations

import logging
from collections import OrderedDict

import torch

from . import ops as O
from . import train as T
from .autograd import engine_of
from .model import LazyStats
from .optim import EftsAdam


class GraphedStep:
    """step = GraphedStep(model, optimizer, scheduler); loss, stats = step(text, text_lengths, speech, speech_lengths)"""

    def __init__(self, model, optimizer, scheduler=None, grad_scale: float = None, capacity: int = 4):
        self.ddp = model if (hasattr(model, "module") and hasattr(model, "reducer")) else None      # dist.DistributedEFTS
        if self.ddp is not None:
            model = self.ddp.module
        if grad_scale is None:
            grad_scale = self.ddp.grad_scale if self.ddp is not None else 1.0
        self.model, self.opt, self.sch, self.grad_scale, self.capacity = model, optimizer, scheduler, float(grad_scale), capacity
        self.entries: "OrderedDict[tuple, dict]" = OrderedDict()
        self.words = None
        self.replays = 0

    def _eligible(self) -> bool:
        m = self.model
        conv_dropout = m.training and float(getattr(m, "dropout_rate", 0.0)) >= 1e-5
        return isinstance(self.opt, EftsAdam) and hasattr(m, "_weights") and not conv_dropout and not torch.cuda.is_current_stream_capturing()

    def inputs(self, text, text_lengths, speech, speech_lengths):
        """The device tensors a captured step of this shape reads (text, text_lengths int32, speech, speech_lengths int32), or None before
        the capture.  A data loader may copy its next batch straight into them and pass them back to the call: no second copy."""
        ent = self.entries.get((tuple(text.shape), tuple(speech.shape), text.dtype, speech.dtype))
        return None if ent is None or ent.get("graph") is None else tuple(ent["static"])

    def _eager(self, text, tl, speech, sl):
        loss, stats, *_ = self.model(text=text, text_lengths=tl, speech=speech, speech_lengths=sl)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step(grad_scale=self.grad_scale)
        if self.sch is not None:
            self.sch.step()
        return loss.detach(), stats

    def _tag(self, ws, eng):
        """what a captured step is valid for: the buffers its launches point at and every argument they carry by value"""
        m, g = self.model, self.opt.param_groups[0]
        red = self.ddp.reducer if self.ddp is not None else None
        return (ws.serial, id(eng), getattr(m, "_ptr_sig", None), float(self.opt.grad_norm), self.grad_scale, tuple(g["betas"]), float(g["eps"]),
                float(g["weight_decay"]), bool(m.training), id(red), None if red is None else red.algo, eng.bucket_hook is not None,
                m.opt.tag(), T.switch_tag())

    def _refresh(self, eng) -> None:
        """the words of the step about to run: Adam's scalars for step t + 1 and the Dropout step word of call dropout_calls + 1"""
        m = self.model
        O.store_words(self.words, self.opt.hyper_words(self.opt.t + 1) + [2 * (int(getattr(m, "dropout_calls", 0)) + 1)])

    def __call__(self, text, text_lengths, speech, speech_lengths):
        m = self.model
        if not self._eligible():
            return self._eager(text, text_lengths, speech, speech_lengths)
        dev = text.device
        key = (tuple(text.shape), tuple(speech.shape), text.dtype, speech.dtype)      # (the lengths are copied into int32 buffers whatever they come as)
        ent = self.entries.get(key)
        if ent is None:
            ent = self.entries[key] = dict(calls=0, graph=None)
            while len(self.entries) > self.capacity:
                self.entries.popitem(last=False)
        self.entries.move_to_end(key)
        ent["calls"] += 1
        eng = engine_of(m)
        pinned = tuple(e["keep"][0] for e in self.entries.values() if e.get("keep"))
        ws = m._workspace(("train", text.shape[0], text.shape[1], speech.shape[1]), dev, pin=pinned)
        tag = self._tag(ws, eng)
        if ent["graph"] is not None and ent["tag"] != tag:          # the buffers the launches point at were re-allocated, or a by-value
                                                                     # hyper-parameter of the captured launches changed
            ent.update(graph=None, calls=2, keep=None)               # (the old workspace is released BEFORE a new capture allocates)
        if ent["calls"] == 1 or ent.get("eager_only"):
            return self._eager(text, text_lengths, speech, speech_lengths)
        with O.stream_scope():
            if self.words is None:
                self.words = torch.zeros(8, dtype=torch.int32, device=dev)
            if ent["graph"] is None:
                # the captured launches read these; lengths are kept as int32 (what the kernels take: the copy below converts, so the
                # replay holds no conversion launches of its own)
                ent["static"] = [text.clone(), text_lengths.to(device=dev, dtype=torch.int32).clone(), speech.clone(),
                                 speech_lengths.to(device=dev, dtype=torch.int32).clone()]
                calls0 = int(getattr(m, "dropout_calls", 0))
                m._packed_sig = None                                  # the weight planes are (re)packed INSIDE the graph, every step
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                eng.step_words = self.words
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        with O.stream_scope():
                            out3, _ = eng.forward_backward(*ent["static"])
                            if eng.join_reduce is not None:           # data parallel: the optimizer waits for the last bucket
                                eng.join_reduce()
                            self.opt.launch(self.grad_scale, hyper_ptr=self.words.data_ptr())
                except Exception as exc:                              # noqa: BLE001
                    logging.warning("hipGraph capture of the training step failed (%s): this shape stays on eager launches", exc)
                    ent["eager_only"] = True
                    # streams that were forked into the failed capture may be left in capture mode: never use them again
                    object.__setattr__(m, "_side", None)
                    if self.ddp is not None and self.ddp.reducer is not None and self.ddp.reducer.comm_stream is not None:
                        self.ddp.reducer.comm_stream = torch.cuda.Stream(device=dev)
                        self.ddp.reducer.pending, self.ddp.reducer.events = False, {}
                    try:
                        torch.cuda.graph.default_capture_stream = None
                    except Exception:                             # noqa: BLE001
                        pass
                    eng.step_words = None
                    m.dropout_calls = calls0
                    torch.cuda.synchronize()
                    return self._eager(text, text_lengths, speech, speech_lengths)
                finally:
                    eng.step_words = None
                m.dropout_calls = calls0                              # (a capture runs nothing)
                ent.update(graph=g, out3=out3, tag=self._tag(m._workspace(("train", text.shape[0], text.shape[1], speech.shape[1]), dev), eng), keep=(ws, eng))
            else:
                for s_, t in zip(ent["static"], (text, text_lengths, speech, speech_lengths)):
                    if t is not s_:                                  # (a caller that fills `inputs(...)` in place passes them back: nothing to copy)
                        s_.copy_(t, non_blocking=True)
            self._refresh(eng)
        ent["graph"].replay()
        self.replays += 1
        # what the eager loop's Python does around its launches
        m.dropout_calls = int(getattr(m, "dropout_calls", 0)) + 1
        self.opt.t += 1
        m._packed_sig = None
        for n, p in eng.named:
            p.grad = eng.g[n]
        if self.sch is not None:
            self.sch.step()
        out3 = ent["out3"].clone()
        return out3[0], LazyStats(out3)
