"""Training entry point with the reference's command line and YAML schema (nntts/bin/train.py:30-253):

    python -m efficient_tts_amd.bin.train --config egs/lj/conf/efficient_tts_cnn_phnseq_noDropout.v1.yaml \
        --train_fid_scp train.txt --dev_fid_scp dev.txt --outdir exp/efts [--resume ckpt] [--pretrain ckpt]

One process per GPU; for N > 1 launch with `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 -m efficient_tts_amd.bin.train ...` (RANK / LOCAL_RANK / WORLD_SIZE from the
environment, backend "nccl" = RCCL over xGMI).  What differs from the reference: the model, optimizer,
gradient all-reduce and mel front-end are this package's HIP implementations (the dataset yields
waveforms, the trainer computes log-mels on the GPU); there is no CPU mode.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

import torch
import yaml
from torch.utils.data import DataLoader

import efficient_tts_amd
from efficient_tts_amd import datasets, models, optimizers, schedulers, trainers
from efficient_tts_amd.dist import DistributedEFTS
from efficient_tts_amd.frontend import LogMelFrontend


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train the EFTS-CNN acoustic model on MI355X (see efficient_tts_amd/bin/train.py).")
    p.add_argument("--train_fid_scp", default=None, type=str, help="file list for training (audiopath|phoneme sequence)")
    p.add_argument("--dev_fid_scp", default=None, type=str, help="file list for validation")
    p.add_argument("--outdir", type=str, required=True, help="directory to save checkpoints")
    p.add_argument("--config", type=str, required=True, help="yaml format configuration file")
    p.add_argument("--pretrain", default="", type=str, nargs="?", help="checkpoint to load parameters from")
    p.add_argument("--resume", default="", type=str, nargs="?", help="checkpoint to resume training from")
    p.add_argument("--verbose", type=int, default=1, help="logging level; higher is more logging")
    p.add_argument("--rank", "--local_rank", default=None, type=int, help="local rank; normally taken from LOCAL_RANK")
    return p


def main(argv=None) -> int:
    args = get_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("efficient_tts_amd needs an MI355X (gfx950) device: there is no CPU path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = args.rank if args.rank is not None else int(os.environ.get("LOCAL_RANK", "0"))
    args.rank = rank
    args.distributed = world > 1
    args.world_size = world
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.distributed:
        torch.distributed.init_process_group(backend="nccl", init_method="env://", device_id=device)
    if rank != 0:
        sys.stdout = open(os.devnull, "w")
    level = logging.DEBUG if args.verbose > 1 else (logging.INFO if args.verbose > 0 else logging.WARN)
    logging.basicConfig(level=level, stream=sys.stdout, force=True,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    os.makedirs(args.outdir, exist_ok=True)

    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.Loader)
    config.update(vars(args))
    config["version"] = getattr(efficient_tts_amd, "__version__", "0.1.0")
    if rank == 0:
        with open(os.path.join(args.outdir, "config.yml"), "w") as f:
            yaml.dump(config, f, Dumper=yaml.Dumper)
    for key, value in config.items():
        logging.info(f"{key} = {value}")

    dataset_class = getattr(datasets, config.get("dataset_type", "TextMelLoader"))
    data_params = config.get("dataset_params", {})
    dataset = {"train": dataset_class(meta_file=args.train_fid_scp, **data_params),
               "dev": dataset_class(meta_file=args.dev_fid_scp, **data_params)}
    logging.info(f"The number of training files = {len(dataset['train'])}.")
    logging.info(f"The number of development files = {len(dataset['dev'])}.")
    collate = getattr(datasets, config.get("collate_fn_type", "TextMelCollate"))(**config.get("collate_fn_params", {}))
    sampler = {"train": None, "dev": None}
    if args.distributed:
        from torch.utils.data.distributed import DistributedSampler
        sampler["train"] = DistributedSampler(dataset["train"], num_replicas=world, rank=rank, shuffle=True)
        sampler["dev"] = DistributedSampler(dataset["dev"], num_replicas=world, rank=rank, shuffle=False)
    data_loader = {k: DataLoader(dataset[k], shuffle=not args.distributed, collate_fn=collate, batch_size=config["batch_size"],
                                 num_workers=config.get("num_workers", 0), sampler=sampler[k],
                                 pin_memory=config.get("pin_memory", False)) for k in ("train", "dev")}

    model = getattr(models, config["model_name"])(**config["model_params"]).to(device)
    opt_params = dict(config["optimizer_params"])
    optimizer = getattr(optimizers, config.get("optimizer_type", "Adam"))(model, grad_norm=config.get("grad_norm", 1.0), **opt_params)
    scheduler = None
    if config.get("scheduler_type") is not None:
        scheduler = getattr(schedulers, config["scheduler_type"])(optimizer=optimizer, **config["scheduler_params"])
    if args.distributed:
        model = DistributedEFTS(model)
    logging.info(model)

    trainer = getattr(trainers, config.get("trainer_type", "EfficientTTSTrainer"))(
        steps=0, epochs=0, data_loader=data_loader, sampler=sampler, model=model, optimizer=optimizer, scheduler=scheduler,
        config=config, device=device)
    trainer.frontend = LogMelFrontend(device, **config.get("frontend_params", {}))      # waveform batches -> log-mel on the GPU
    if args.pretrain:
        trainer.load_checkpoint(args.pretrain, load_only_params=True)
        logging.info(f"Successfully load parameters from {args.pretrain}.")
    if args.resume:
        trainer.load_checkpoint(args.resume)
        logging.info(f"Successfully resumed from {args.resume}.")
    try:
        trainer.run()
    except KeyboardInterrupt:
        trainer.save_checkpoint(os.path.join(config["outdir"], f"checkpoint-{trainer.steps}steps.pkl"))
        logging.info(f"Successfully saved checkpoint @ {trainer.steps}steps.")
    if args.distributed:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
