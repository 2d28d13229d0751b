"""EfficientTTSTrainer -- same constructor, `run()`, checkpoint format and interval logic as the
reference trainer (nntts/trainers/efficient_tts_trainer.py:20-281), driving the MI355X model.

Differences that do not change the contract: tensorboardX and tqdm are optional (absent in this
image); `stats` values are read lazily (no per-step host sync); with `EftsAdam` the gradient clip is
fused into the optimizer kernel; with `DistributedEFTS` the bucketed RCCL all-reduce launched during
backward is joined right before the optimizer step.
"""
from __future__ import annotations

import logging
import os
from collections import defaultdict

import torch

try:                                     # optional, as in the reference (trainer.py:13)
    from tensorboardX import SummaryWriter
except Exception:                        # pragma: no cover
    SummaryWriter = None
try:
    from tqdm import tqdm
except Exception:                        # pragma: no cover
    tqdm = None


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


class EfficientTTSTrainer(object):
    def __init__(self, steps, epochs, data_loader, sampler, model, optimizer, scheduler, config,
                 device=torch.device("cpu")):
        self.steps, self.epochs = steps, epochs
        self.data_loader, self.sampler = data_loader, sampler
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.config, self.device = config, device
        self.writer = SummaryWriter(config["outdir"]) if SummaryWriter is not None else _NullWriter()
        self.finish_train = False
        self.total_train_loss = defaultdict(float)
        self.total_eval_loss = defaultdict(float)
        self._pending = []                # LazyStats of the steps since the last log (read at log time)
        self.frontend = None              # optional LogMelFrontend: batches then carry (audio, audio_lengths) instead of mels

    # ------------------------------------------------------------------ loop (trainer.py:62-76,167-191)
    def run(self):
        self.tqdm = tqdm(initial=self.steps, total=self.config["train_max_steps"], desc="[train]") if tqdm else None
        while True:
            self._train_epoch()
            if self.finish_train:
                break
        if self.tqdm:
            self.tqdm.close()
        logging.info("Finished training.")

    def _raw_model(self):
        return self.model.module if self.config.get("distributed") or hasattr(self.model, "module") else self.model

    # ------------------------------------------------------------------ checkpoints (trainer.py:78-119)
    def save_checkpoint(self, checkpoint_path):
        state_dict = {"optimizer": self.optimizer.state_dict(), "steps": self.steps, "epochs": self.epochs}
        if self.scheduler is not None:
            state_dict["scheduler"] = self.scheduler.state_dict()
        state_dict["model"] = {k: v.detach().clone() for k, v in self._raw_model().state_dict().items()}
        d = os.path.dirname(checkpoint_path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        torch.save(state_dict, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, load_only_params=False):
        state_dict = torch.load(checkpoint_path, map_location="cpu")
        self._raw_model().load_state_dict(state_dict["model"])
        self._raw_model()._packed_sig = None
        if not load_only_params:
            self.steps, self.epochs = state_dict["steps"], state_dict["epochs"]
            self.optimizer.load_state_dict(state_dict["optimizer"])
            if self.scheduler is not None:
                self.scheduler.load_state_dict(state_dict["scheduler"])

    # ------------------------------------------------------------------ one step (trainer.py:121-165)
    def _to_device(self, batch):
        text, text_lengths, mel, mel_lengths = [x.to(self.device) for x in batch]
        if self.frontend is not None:     # waveform batch (efficient_tts_amd.datasets.TextMelCollate): log-mel on the GPU
            # optional shape bucketing (config "bucket_frames" / "bucket_phones", 0 = exact batch maximum as the
            # reference pads): fewer distinct (T1, T2) shapes -> the engine's per-shape workspaces are re-used
            bf, bp = int(self.config.get("bucket_frames", 0)), int(self.config.get("bucket_phones", 0))
            frames = self.frontend.frames_of(mel_lengths)
            tmax = int(frames.max())
            if bf > 0:
                tmax = (tmax + bf - 1) // bf * bf
            mel, mel_lengths = self.frontend(mel, mel_lengths, max_frames=tmax)
            if bp > 0 and text.shape[1] % bp:
                text = torch.nn.functional.pad(text, (0, bp - text.shape[1] % bp))
        return text, text_lengths, mel, mel_lengths

    def _train_step(self, batch):
        text, text_lengths, mel, mel_lengths = self._to_device(batch)
        loss, stats, *_ = self.model(text=text, text_lengths=text_lengths, speech=mel, speech_lengths=mel_lengths)
        self._pending.append(stats)
        self.optimizer.zero_grad()
        loss.backward()
        if hasattr(self.model, "finish_reduce"):
            self.model.finish_reduce()                      # join the bucketed all-reduce
        scale = getattr(self.model, "grad_scale", 1.0)
        if hasattr(self.optimizer, "grad_norm"):            # EftsAdam: clip fused into the update kernel
            self.optimizer.grad_norm = float(self.config["grad_norm"])
            self.optimizer.step(grad_scale=scale)
        else:
            if scale != 1.0:
                for p in self._raw_model().parameters():
                    if p.grad is not None:
                        p.grad.mul_(scale)
            if self.config["grad_norm"] > 0:
                torch.nn.utils.clip_grad_norm_(self._raw_model().parameters(), self.config["grad_norm"])
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        self.steps += 1
        if self.tqdm:
            self.tqdm.update(1)
        self._check_train_finish()

    def _drain_stats(self):
        for st in self._pending:
            self.total_train_loss["train/loss"] += st["loss"]
            self.total_train_loss["train/mel_loss"] += st["mel_loss"]
            self.total_train_loss["train/dur_loss"] += st["duration_loss"]
        self._pending = []

    def _train_epoch(self):
        train_steps_per_epoch = 0
        for train_steps_per_epoch, batch in enumerate(self.data_loader["train"], 1):
            self._train_step(batch)
            if self.config["rank"] == 0:
                self._check_log_interval()
                self._check_eval_interval()
                self._check_save_interval()
            if self.finish_train:
                return
        self.epochs += 1
        self.train_steps_per_epoch = train_steps_per_epoch
        logging.info(f"(Steps: {self.steps}) Finished {self.epochs} epoch training "
                     f"({self.train_steps_per_epoch} steps per epoch).")
        if self.config["distributed"]:
            self.sampler["train"].set_epoch(self.epochs)

    # ------------------------------------------------------------------ eval (trainer.py:193-252)
    @torch.no_grad()
    def _eval_step(self, batch, plot=False):
        text, text_lengths, mel, mel_lengths = self._to_device(batch)
        loss, stats, imv, alpha, mel_pred, mel_gt = self._raw_model()(text=text, text_lengths=text_lengths, speech=mel,
                                                                      speech_lengths=mel_lengths)
        self.total_eval_loss["eval/loss"] += stats["loss"]
        self.total_eval_loss["eval/mel_loss"] += stats["mel_loss"]
        self.total_eval_loss["eval/dur_loss"] += stats["duration_loss"]

    def _eval_epoch(self):
        logging.info(f"(Steps: {self.steps}) Start evaluation.")
        self._raw_model().eval()
        n = 0
        for n, batch in enumerate(self.data_loader["dev"], 1):
            self._eval_step(batch, plot=(n == 1))
        logging.info(f"(Steps: {self.steps}) Finished evaluation ({n} steps per epoch).")
        for key in self.total_eval_loss.keys():
            self.total_eval_loss[key] /= max(n, 1)
            logging.info(f"(Steps: {self.steps}) {key} = {self.total_eval_loss[key]:.4f}.")
        self._write_to_tensorboard(self.total_eval_loss)
        self.total_eval_loss = defaultdict(float)
        self._raw_model().train()

    def _write_to_tensorboard(self, loss):
        for key, value in loss.items():
            self.writer.add_scalar(key, value, self.steps)

    # ------------------------------------------------------------------ intervals (trainer.py:259-281)
    def _check_save_interval(self):
        if self.steps % self.config["save_interval_steps"] == 0:
            self.save_checkpoint(os.path.join(self.config["outdir"], f"checkpoint-{self.steps}steps.pkl"))
            logging.info(f"Successfully saved checkpoint @ {self.steps} steps.")

    def _check_eval_interval(self):
        if self.steps % self.config["eval_interval_steps"] == 0:
            self._eval_epoch()

    def _check_log_interval(self):
        if self.steps % self.config["log_interval_steps"] == 0:
            self._drain_stats()
            for key in self.total_train_loss.keys():
                self.total_train_loss[key] /= self.config["log_interval_steps"]
                logging.info(f"(Steps: {self.steps}) {key} = {self.total_train_loss[key]:.4f}.")
            self._write_to_tensorboard(self.total_train_loss)
            self.total_train_loss = defaultdict(float)

    def _check_train_finish(self):
        if self.steps >= self.config["train_max_steps"]:
            self.finish_train = True
