"""`python -m efficient_tts_amd.bin.train` -- trains EFTS-CNN on MI355X from the reference recipe's files.

Drop-in for the command line and YAML schema of the reference entry point (nntts/bin/train.py:30-253), so the
recipe's `run.sh` and `egs/lj/conf/*.yaml` work unchanged:

    python -m efficient_tts_amd.bin.train --config conf.yaml --train_fid_scp train.txt --dev_fid_scp dev.txt \\
        --outdir exp/efts [--resume exp/efts/checkpoint-5000steps.pkl | --pretrain other.pkl] [--verbose 1]

Multi-GPU: one process per GPU, e.g. `python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1
-m efficient_tts_amd.bin.train ...`; RANK / LOCAL_RANK / WORLD_SIZE come from the environment and the backend
"nccl" is RCCL over xGMI.  The pieces named in the YAML (`dataset_type`, `collate_fn_type`, `model_name`,
`optimizer_type`, `scheduler_type`, `trainer_type`) are resolved in this package's registries
(efficient_tts_amd.datasets / models / optimizers / schedulers / trainers); the dataset yields waveforms and the
trainer computes log-mels on the GPU (efficient_tts_amd.frontend).  There is no CPU mode.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import yaml
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import efficient_tts_amd as pkg
from efficient_tts_amd import datasets, models, optimizers, schedulers, trainers
from efficient_tts_amd.dist import DistributedEFTS
from efficient_tts_amd.frontend import LogMelFrontend

# (flags, kwargs) of the reference command line, kept as data so the parser and the docs cannot drift apart
_CLI = (
    (("--config",), dict(type=str, required=True, help="YAML recipe (egs/*/conf/*.yaml schema)")),
    (("--outdir",), dict(type=str, required=True, help="where config.yml and checkpoint-*steps.pkl go")),
    (("--train_fid_scp",), dict(type=str, default=None, help="training file list: audiopath|phoneme sequence")),
    (("--dev_fid_scp",), dict(type=str, default=None, help="validation file list")),
    (("--resume",), dict(type=str, default="", nargs="?", help="checkpoint to continue from (model, optimizer, schedule, counters)")),
    (("--pretrain",), dict(type=str, default="", nargs="?", help="checkpoint to take only the parameters from")),
    (("--verbose",), dict(type=int, default=1, help="0 warnings only, 1 info, 2 debug")),
    (("--rank", "--local_rank"), dict(type=int, default=None, help="local device index; normally LOCAL_RANK")),
)


def get_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="efficient_tts_amd.bin.train", description=__doc__.splitlines()[0])
    for flags, kwargs in _CLI:
        parser.add_argument(*flags, **kwargs)
    return parser


@dataclass
class _Process:
    """where this process sits in the job (one process per GPU)"""
    rank: int
    local_rank: int
    world: int

    @property
    def distributed(self) -> bool:
        return self.world > 1

    @property
    def device(self) -> torch.device:
        return torch.device("cuda", self.local_rank)

    @staticmethod
    def from_env(cli_local_rank: Optional[int]) -> "_Process":
        env = os.environ
        local = cli_local_rank if cli_local_rank is not None else int(env.get("LOCAL_RANK", 0))
        return _Process(rank=int(env.get("RANK", 0)), local_rank=local, world=int(env.get("WORLD_SIZE", 1)))


def _setup_logging(verbose: int, quiet: bool) -> None:
    if quiet:                                   # every rank but 0 keeps silent, like the reference
        sys.stdout = open(os.devnull, "w")
    level = {0: logging.WARNING, 1: logging.INFO}.get(verbose, logging.DEBUG)
    logging.basicConfig(level=level, stream=sys.stdout, force=True,
                        format="%(asctime)s %(levelname)s %(name)s:%(lineno)d  %(message)s")


def _load_config(args: argparse.Namespace, proc: _Process) -> Dict[str, Any]:
    with open(args.config) as handle:
        config = yaml.safe_load(handle) or {}
    config.update(vars(args))                   # the reference merges the CLI into the recipe; trainer reads both
    config.update(rank=proc.rank, distributed=proc.distributed, world_size=proc.world,
                  version=getattr(pkg, "__version__", "0.1.0"))
    if proc.rank == 0:
        os.makedirs(args.outdir, exist_ok=True)
        with open(os.path.join(args.outdir, "config.yml"), "w") as handle:
            yaml.safe_dump(config, handle)
    for key in sorted(config):
        logging.info(f"config {key} = {config[key]}")
    return config


def _build_data(config: Dict[str, Any], args: argparse.Namespace, proc: _Process):
    make_dataset = getattr(datasets, config.get("dataset_type", "TextMelLoader"))
    params = config.get("dataset_params") or {}
    splits = {"train": make_dataset(meta_file=args.train_fid_scp, **params),
              "dev": make_dataset(meta_file=args.dev_fid_scp, **params)}
    collate = getattr(datasets, config.get("collate_fn_type", "TextMelCollate"))(**(config.get("collate_fn_params") or {}))
    samplers: Dict[str, Optional[DistributedSampler]] = {"train": None, "dev": None}
    if proc.distributed:
        samplers = {name: DistributedSampler(ds, num_replicas=proc.world, rank=proc.rank, shuffle=(name == "train"))
                    for name, ds in splits.items()}
    loaders = {name: DataLoader(ds, batch_size=int(config["batch_size"]), collate_fn=collate, sampler=samplers[name],
                                shuffle=samplers[name] is None, num_workers=int(config.get("num_workers", 0)),
                                pin_memory=bool(config.get("pin_memory", False)))
               for name, ds in splits.items()}
    for name, ds in splits.items():
        logging.info(f"{name}: {len(ds)} utterances")
    return loaders, samplers


def _build_trainer(config: Dict[str, Any], loaders, samplers, proc: _Process):
    device = proc.device
    net = getattr(models, config["model_name"])(**config["model_params"]).to(device)
    optimizer = getattr(optimizers, config.get("optimizer_type", "Adam"))(
        net, grad_norm=float(config.get("grad_norm", 1.0)), **config["optimizer_params"])
    scheduler = None
    if config.get("scheduler_type"):
        scheduler = getattr(schedulers, config["scheduler_type"])(optimizer=optimizer, **(config.get("scheduler_params") or {}))
    model = DistributedEFTS(net) if proc.distributed else net
    logging.info(f"{model}")
    trainer_class = getattr(trainers, config.get("trainer_type", "EfficientTTSTrainer"))
    trainer = trainer_class(steps=0, epochs=0, data_loader=loaders, sampler=samplers, model=model, optimizer=optimizer,
                            scheduler=scheduler, config=config, device=device)
    trainer.frontend = LogMelFrontend(device, **(config.get("frontend_params") or {}))
    return trainer


def main(argv=None) -> int:
    args = get_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X (gfx950) device visible: efficient_tts_amd has no CPU path")
    proc = _Process.from_env(args.rank)
    torch.cuda.set_device(proc.local_rank)
    if proc.distributed:
        torch.distributed.init_process_group(backend="nccl", init_method="env://", device_id=proc.device)
    _setup_logging(args.verbose, quiet=proc.rank != 0)
    config = _load_config(args, proc)
    loaders, samplers = _build_data(config, args, proc)
    trainer = _build_trainer(config, loaders, samplers, proc)
    if args.pretrain:
        trainer.load_checkpoint(args.pretrain, load_only_params=True)
        logging.info(f"parameters initialised from {args.pretrain}")
    if args.resume:
        trainer.load_checkpoint(args.resume)
        logging.info(f"resumed from {args.resume} at step {trainer.steps}")
    try:
        trainer.run()
    except KeyboardInterrupt:                   # same courtesy as the reference: keep what was learnt so far
        path = os.path.join(config["outdir"], f"checkpoint-{trainer.steps}steps.pkl")
        trainer.save_checkpoint(path)
        logging.info(f"interrupted: state saved to {path}")
    finally:
        if proc.distributed:
            torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
