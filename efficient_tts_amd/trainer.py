"""Training loop driver for the MI355X EFTS-CNN engine.

Public contract = the reference trainer's (nntts/trainers/efficient_tts_trainer.py:20-281), because
the recipe's entry point builds it by name and positional/keyword arguments (nntts/bin/train.py:218-231):

    EfficientTTSTrainer(steps, epochs, data_loader, sampler, model, optimizer, scheduler, config, device)
    .run()   .save_checkpoint(path)   .load_checkpoint(path, load_only_params=False)   .steps   .epochs

Checkpoints are `torch.save`d dicts with the keys {"model", "optimizer", "scheduler", "steps", "epochs"}
and the model's reference `state_dict` key set, written as `checkpoint-{steps}steps.pkl` under
config["outdir"] every config["save_interval_steps"]; evaluation and logging follow
config["eval_interval_steps"] / config["log_interval_steps"]; training stops at config["train_max_steps"].

Everything inside is organised for this engine rather than copied from the reference loop:
  * the per-step statistics stay on the device (`LazyStats`) and are only read when a log line is due, so
    a step never synchronises the host;
  * with `EftsAdam` the gradient clip is part of the fused optimizer kernel; with `DistributedEFTS` the
    bucketed RCCL all-reduce launched during backward is joined right before the optimizer step;
  * an optional `frontend` (efficient_tts_amd.frontend.LogMelFrontend) turns waveform batches into
    log-mels on the GPU; `bucket_frames` / `bucket_phones` pad shapes up to multiples so the engine's
    per-shape workspaces are re-used across ragged batches;
  * tensorboardX / tqdm are used when importable, silently skipped otherwise.
"""
from __future__ import annotations

import logging
import os
from typing import Dict

import torch

try:
    from tensorboardX import SummaryWriter as _TBWriter
except Exception:                                       # pragma: no cover - optional dependency
    _TBWriter = None
try:
    from tqdm import tqdm as _tqdm
except Exception:                                       # pragma: no cover - optional dependency
    _tqdm = None

log = logging.getLogger(__name__)
_STAT_KEYS = (("loss", "loss"), ("mel_loss", "mel_loss"), ("dur_loss", "duration_loss"))   # (log suffix, stats key)


class _Meter:
    """Running sums of the three losses under a name prefix ("train" / "eval")."""

    def __init__(self, prefix: str):
        self.prefix = prefix
        self.reset()

    def reset(self) -> None:
        self.sums: Dict[str, float] = {f"{self.prefix}/{suffix}": 0.0 for suffix, _ in _STAT_KEYS}

    def add(self, stats) -> None:
        for suffix, key in _STAT_KEYS:
            self.sums[f"{self.prefix}/{suffix}"] += float(stats[key])

    def means(self, count: int) -> Dict[str, float]:
        return {k: v / max(count, 1) for k, v in self.sums.items()}


class EfficientTTSTrainer:
    def __init__(self, steps, epochs, data_loader, sampler, model, optimizer, scheduler, config,
                 device=torch.device("cpu")):
        self.steps = int(steps)
        self.epochs = int(epochs)
        self.data_loader = data_loader
        self.sampler = sampler
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.config = config
        self.device = device
        self.frontend = None                   # optional LogMelFrontend (set by efficient_tts_amd.bin.train)
        self.finish_train = False
        self._tb = _TBWriter(config["outdir"]) if _TBWriter is not None else None
        self._bar = None
        self._unread = []                      # LazyStats of the steps since the last log line
        self._train_meter = _Meter("train")
        self._eval_meter = _Meter("eval")

    # ------------------------------------------------------------------------------------------ helpers
    @property
    def _net(self):
        """the bare model (DistributedEFTS / DDP keep it under .module)"""
        return self.model.module if hasattr(self.model, "module") else self.model

    def _due(self, key: str) -> bool:
        every = int(self.config[key])
        return every > 0 and self.steps % every == 0

    def _publish(self, values: Dict[str, float]) -> None:
        for name, value in values.items():
            log.info(f"[step {self.steps}] {name}: {value:.4f}")
            if self._tb is not None:
                self._tb.add_scalar(name, value, self.steps)

    def _stage(self, batch):
        """batch -> (text, text_lengths, mel, mel_lengths) on the device.  Waveform batches
        (efficient_tts_amd.datasets.TextMelCollate) go through the GPU log-mel front-end."""
        if self.frontend is None and self._graphed_step() is not None and self._net.training:
            # `graph_steps`: once this batch shape is captured, the host tensors go straight into the buffers the captured launches read
            # (one host-to-device copy each, converting the lengths to int32 on the way; GraphedStep sees its own buffers and copies nothing)
            bufs = self._graph.inputs(*batch)
            if bufs is not None:
                for s_, t in zip(bufs, batch):
                    s_.copy_(t, non_blocking=True)
                return bufs
        text, text_lengths, third, third_lengths = (t.to(self.device) for t in batch)
        if self.frontend is None:
            return text, text_lengths, third, third_lengths
        n_frames = int(self.frontend.frames_of(third_lengths).max())
        # default buckets: ragged LJSpeech batches otherwise bring a new (B, T1, T2) almost every step, and every new shape
        # allocates and zero-fills a multi-GB activation workspace (4 are cached); 0 in the YAML turns the padding off
        # With an UNMASKED FastSpeechLoss (use_masking=False, the reference ctor default) the losses are means over the padded
        # tensors: extra padding would change their denominators and add |0 - pad| terms, so bucketing defaults to off there.
        masked = bool(getattr(self._net, "use_masking", True))
        frame_step = int(self.config.get("bucket_frames", 64 if masked else 0))
        phone_step = int(self.config.get("bucket_phones", 16 if masked else 0))
        if frame_step > 0:
            n_frames = -(-n_frames // frame_step) * frame_step
        mel, mel_lengths = self.frontend(third, third_lengths, max_frames=n_frames)
        if phone_step > 0 and text.shape[1] % phone_step:
            text = torch.nn.functional.pad(text, (0, phone_step - text.shape[1] % phone_step))
        return text, text_lengths, mel, mel_lengths

    # ------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, checkpoint_path: str) -> None:
        payload = {
            "model": {name: value.detach().clone() for name, value in self._net.state_dict().items()},
            "optimizer": self.optimizer.state_dict(),
            "steps": self.steps,
            "epochs": self.epochs,
        }
        if self.scheduler is not None:
            payload["scheduler"] = self.scheduler.state_dict()
        folder = os.path.dirname(checkpoint_path)
        if folder:
            os.makedirs(folder, exist_ok=True)
        torch.save(payload, checkpoint_path)

    def load_checkpoint(self, checkpoint_path: str, load_only_params: bool = False) -> None:
        payload = torch.load(checkpoint_path, map_location="cpu")
        net = self._net
        net.load_state_dict(payload["model"])
        net._packed_sig = None                 # operand planes are re-packed from the new parameters
        if load_only_params:
            return
        self.steps, self.epochs = payload["steps"], payload["epochs"]
        net.dropout_calls = int(self.steps)       # position in the counter-based dropout mask sequence: one draw per training step,
                                                  # so --resume continues the sequence instead of replaying it (no extra checkpoint key)
        self.optimizer.load_state_dict(payload["optimizer"])
        if self.scheduler is not None and "scheduler" in payload:
            self.scheduler.load_state_dict(payload["scheduler"])

    # ------------------------------------------------------------------------------------------ one optimisation step
    def _graphed_step(self):
        """the GraphedStep of this run, or None: off unless the YAML asks for it, and only for the fused optimizer.  Under data
        parallelism the wrapper itself is handed over: the bucket collectives are captured with the step (step_graph.py)"""
        if not self.config.get("graph_steps", False) or not hasattr(self.optimizer, "launch"):
            return None
        if getattr(self, "_graph", None) is None:
            from .step_graph import GraphedStep
            dp = hasattr(self.model, "finish_reduce")
            self._graph = GraphedStep(self.model if dp else self._net, self.optimizer, self.scheduler,
                                      grad_scale=getattr(self.model, "grad_scale", 1.0), capacity=int(self.config.get("graph_shapes", 4)))
        return self._graph

    def _train_step(self, batch) -> None:
        text, text_lengths, mel, mel_lengths = self._stage(batch)
        if self._graphed_step() is not None:
            # `graph_steps: true` (YAML): forward + backward + clip + Adam + schedule as one hipGraph replay per batch shape
            # (efficient_tts_amd/step_graph.py; same results as the launches below).  Pays with a fixed set of batch shapes
            # (bucketed / padded loaders): every new shape is captured once and keeps its activation workspace.
            self._graph.opt.grad_norm = float(self.config["grad_norm"])
            loss, stats = self._graph(text, text_lengths, mel, mel_lengths)
            if int(self.config.get("rank", 0)) == 0:
                self._unread.append(stats)
            self.steps += 1
            if self._bar is not None:
                self._bar.update(1)
            self.finish_train = self.steps >= int(self.config["train_max_steps"])
            return
        loss, stats, *_ = self.model(text=text, text_lengths=text_lengths, speech=mel, speech_lengths=mel_lengths)
        if int(self.config.get("rank", 0)) == 0:
            self._unread.append(stats)                          # drained by _after_step, which only rank 0 runs
        self.optimizer.zero_grad()
        loss.backward()
        if hasattr(self.model, "finish_reduce"):
            self.model.finish_reduce()                          # join the bucketed all-reduce
        clip = float(self.config["grad_norm"])
        scale = getattr(self.model, "grad_scale", 1.0)          # 1 / world under data parallelism
        if hasattr(self.optimizer, "grad_norm"):                # EftsAdam: scale + clip + update in one kernel
            self.optimizer.grad_norm = clip
            self.optimizer.step(grad_scale=scale)
        else:                                                   # any torch optimizer
            params = [p for p in self._net.parameters() if p.grad is not None]
            if scale != 1.0:
                for p in params:
                    p.grad.mul_(scale)
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(params, clip)
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        self.steps += 1
        if self._bar is not None:
            self._bar.update(1)
        self.finish_train = self.steps >= int(self.config["train_max_steps"])

    def _after_step(self) -> None:
        """rank-0 bookkeeping: log line, evaluation, checkpoint -- each on its own interval"""
        if self._due("log_interval_steps"):
            for stats in self._unread:                          # the only host reads of the training statistics
                self._train_meter.add(stats)
            self._unread = []
            self._publish(self._train_meter.means(int(self.config["log_interval_steps"])))
            self._train_meter.reset()
        if self._due("eval_interval_steps"):
            self._evaluate()
        if self._due("save_interval_steps"):
            path = os.path.join(self.config["outdir"], f"checkpoint-{self.steps}steps.pkl")
            self.save_checkpoint(path)
            log.info(f"[step {self.steps}] checkpoint written: {path}")

    # ------------------------------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def _evaluate(self) -> None:
        net = self._net
        net.eval()
        batches = 0
        for batch in self.data_loader["dev"]:
            text, text_lengths, mel, mel_lengths = self._stage(batch)
            out = net(text=text, text_lengths=text_lengths, speech=mel, speech_lengths=mel_lengths)
            self._eval_meter.add(out[1])
            batches += 1
        self._publish(self._eval_meter.means(batches))
        log.info(f"[step {self.steps}] evaluated {batches} dev batches")
        self._eval_meter.reset()
        net.train()

    # ------------------------------------------------------------------------------------------ driver
    def run(self) -> None:
        if _tqdm is not None:
            self._bar = _tqdm(initial=self.steps, total=int(self.config["train_max_steps"]), desc="[train]")
        is_main = int(self.config.get("rank", 0)) == 0
        while not self.finish_train:
            seen = 0
            for batch in self.data_loader["train"]:
                self._train_step(batch)
                seen += 1
                if is_main:
                    self._after_step()
                if self.finish_train:
                    break
            else:                                               # the loader was exhausted: one more epoch done
                self.epochs += 1
                log.info(f"[step {self.steps}] epoch {self.epochs} finished ({seen} steps)")
                train_sampler = (self.sampler or {}).get("train")
                if train_sampler is not None and hasattr(train_sampler, "set_epoch"):
                    train_sampler.set_epoch(self.epochs)
                if seen == 0:
                    raise RuntimeError("the training data loader yields no batches")
        if self._bar is not None:
            self._bar.close()
        log.info("training finished")
