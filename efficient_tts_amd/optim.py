"""Optimizer and schedule of the reference recipe, fused for MI355X.

* `EftsAdam` -- torch.optim.Adam(lr, betas, eps, weight_decay (coupled L2), amsgrad=True)
  (egs/lj/conf/efficient_tts_cnn_phnseq_noDropout.v1.yaml:34-40) as ONE HBM-bound kernel over flat
  fp32 buffers, with `clip_grad_norm_` (trainer.py:154-157) folded in: the global-norm reduction
  stays on the device, no host sync per step.  Parameters are re-homed as views of one flat buffer
  in the engine's gradient layout (so data-parallel buckets are contiguous).
* `WarmupLR` -- nntts/schedulers/warmup_lr.py:9-51: lr * w^0.5 * min(s^-0.5, s * w^-1.5).
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.optim.lr_scheduler import _LRScheduler

from . import lib as L
from . import ops as O
from .autograd import engine_of


class EftsAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0):
        if not amsgrad:
            raise NotImplementedError("the fused kernel implements amsgrad=True (the reference YAML)")
        self.model = model
        self.eng = engine_of(model)
        eng = self.eng
        eng.bound.add("EftsAdam")
        self.grad_norm = float(grad_norm)
        dev = eng.dev
        # re-home parameters into one flat buffer (engine layout)
        self.flat_p = torch.empty_like(eng.flat)
        with torch.no_grad():
            for n, p in eng.layout:
                a, b = eng.offsets[n]
                self.flat_p[a:b].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[a:b].view_as(p)
        self.m = torch.zeros_like(eng.flat)
        self.v = torch.zeros_like(eng.flat)
        self.vmax = torch.zeros_like(eng.flat)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sumsq_ws = torch.zeros(L.load().efts_sumsq_workspace_bytes() // 4, dtype=torch.float32, device=dev)
        self.t = 0
        super().__init__([p for _, p in eng.layout], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """clip (global norm of grad_scale * flat grads) + Adam-amsgrad, all on the device."""
        self.t += 1
        self.launch(grad_scale)
        # parameters changed in place through the flat view: invalidate packed-weight caches
        self.model._packed_sig = None

    def hyper_words(self, step: int):
        """{lr, 1 - beta1^step, sqrt(1 - beta2^step)} as the kernel derives them from its by-value arguments, as 32-bit words"""
        grp = self.param_groups[0]
        arr = (C.c_float * 3)()
        L.check(L.load().efts_adam_hyper(float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), int(step), arr), "efts_adam_hyper")
        return list((C.c_uint32 * 3).from_buffer(arr))

    @torch.no_grad()
    def launch(self, grad_scale: float = 1.0, hyper_ptr=None):
        """the launches of step() for step number `self.t`, nothing else.  hyper_ptr: device floats {lr, 1 - beta1^t, sqrt(1 - beta2^t)}
        read by the kernel instead of the by-value scalars (a step captured as a hipGraph: step_graph.GraphedStep)."""
        eng = self.eng
        grp = self.param_groups[0]
        n = eng.numel
        st = O._stream()
        lib = L.load()
        if self.grad_norm > 0:
            self.sumsq.zero_()
            L.check(lib.efts_sumsq(eng.flat.data_ptr(), n, self.sumsq.data_ptr(), self.sumsq_ws.data_ptr(), st), "efts_sumsq")
        sq = self.sumsq.data_ptr() if self.grad_norm > 0 else None
        if hyper_ptr is None:
            L.check(lib.efts_adam_amsgrad(self.flat_p.data_ptr(), eng.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.vmax.data_ptr(), n, sq,
                                          self.grad_norm, float(grad_scale), float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
                                          float(grp["eps"]), float(grp["weight_decay"]), self.t, st), "efts_adam_amsgrad")
        else:
            L.check(lib.efts_adam_amsgrad_dev(self.flat_p.data_ptr(), eng.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.vmax.data_ptr(), n, sq,
                                              self.grad_norm, float(grad_scale), hyper_ptr, float(grp["betas"][0]), float(grp["betas"][1]),
                                              float(grp["eps"]), float(grp["weight_decay"]), st), "efts_adam_amsgrad_dev")

    def zero_grad(self, set_to_none: bool = True):
        for _, p in self.eng.layout:
            p.grad = None

    def state_dict(self):
        """torch.optim.Adam(amsgrad=True) layout -- {"state": {i: {step, exp_avg, exp_avg_sq, max_exp_avg_sq}}, "param_groups"}
        with i = the parameter's position in model.parameters(), as the reference's optimizer numbers them
        (nntts/bin/train.py:196-205) -- so `--resume` works across the two implementations
        (nntts/trainers/efficient_tts_trainer.py:90-103 saves optimizer.state_dict() as is)."""
        eng, state = self.eng, {}
        names = [n for n, _ in self.model.named_parameters()]
        for i, n in enumerate(names):
            a, b = eng.offsets[n]
            shape = dict(eng.layout)[n].shape
            state[i] = dict(step=torch.tensor(float(self.t)), exp_avg=self.m[a:b].view(shape).clone(),
                            exp_avg_sq=self.v[a:b].view(shape).clone(), max_exp_avg_sq=self.vmax[a:b].view(shape).clone())
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d.update(amsgrad=True, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(names))))
            groups.append(d)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        eng = self.eng
        if "state" in sd:                                        # torch Adam layout (ours or the reference's)
            names = [n for n, _ in self.model.named_parameters()]
            if len(sd["state"]) not in (0, len(names)):
                raise ValueError(f"optimizer state holds {len(sd['state'])} parameters, the model has {len(names)}")
            steps = set()
            for i, n in enumerate(names):
                st = sd["state"].get(i)
                if st is None:
                    continue
                a, b = eng.offsets[n]
                self.m[a:b].copy_(st["exp_avg"].reshape(-1))
                self.v[a:b].copy_(st["exp_avg_sq"].reshape(-1))
                self.vmax[a:b].copy_(st.get("max_exp_avg_sq", st["exp_avg_sq"]).reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one")
            self.t = steps.pop() if steps else 0
        else:                                                    # round-1 flat layout
            self.t = int(sd["t"])
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.vmax.copy_(sd["vmax"])
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s_.items() if k in ("lr", "betas", "eps", "weight_decay", "initial_lr")})


class WarmupLR(_LRScheduler):
    """lr = base_lr * warmup^0.5 * min(step^-0.5, step * warmup^-1.5), step = last_epoch + 1
    (nntts/schedulers/warmup_lr.py:44-51)."""

    def __init__(self, optimizer, warmup_steps=25000, last_epoch=-1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, last_epoch)

    def __repr__(self):
        return f"{self.__class__.__name__}(warmup_steps={self.warmup_steps})"

    def get_lr(self):
        s = self.last_epoch + 1
        return [lr * self.warmup_steps ** 0.5 * min(s ** -0.5, s * self.warmup_steps ** -1.5) for lr in self.base_lrs]
