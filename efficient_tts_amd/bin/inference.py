"""`python -m efficient_tts_amd.bin.inference` -- text (phoneme sequences) to 16-bit wav files on MI355X.

Takes the reference synthesis script's command line (nntts/bin/inference.py:128-176):

    python -m efficient_tts_amd.bin.inference --checkpoint exp/efts/checkpoint-100000steps.pkl --test_fid_scp test.txt \\
        --outdir exp/efts/wav [--config exp/efts/config.yml] [--verbose 1]

and runs the whole chain on the GPU: `EfficientTTSCNN.inference` (free-running acoustic model) followed by the
HiFi-GAN V1 generator (efficient_tts_amd.vocoder).  Differences from the reference script:
  * the vocoder weights are named explicitly (`--vocoder_config`, `--vocoder_checkpoint`): the reference hard-codes
    a checkpoint that is not part of its repository; without one the generator runs with random weights and says so
    (useful for smoke tests and timing only), `--no_vocoder` writes the mel-spectrograms as .npy instead;
  * `--batch_size N` synthesises N utterances per call with `inference_batch` (each item equals its B = 1 result);
  * every line of the list is processed (the reference stops after 10), alignment plots are not drawn;
  * RTF is reported as the reference does (wall time of model + vocoder over audio duration), synchronised per call.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time
from typing import List, Tuple

import numpy as np
import torch
import yaml

from efficient_tts_amd import models
from efficient_tts_amd.vocoder import HiFiGANGenerator, load_hifigan_generator

_V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
           resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=80)
SAMPLING_RATE = 22050


def get_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="efficient_tts_amd.bin.inference", description=__doc__.splitlines()[0])
    p.add_argument("--checkpoint", type=str, required=True, help="acoustic-model checkpoint (checkpoint-*steps.pkl)")
    p.add_argument("--test_fid_scp", type=str, required=True, help="utterance list: wav_path|phoneme sequence")
    p.add_argument("--outdir", type=str, required=True, help="where the generated speech goes")
    p.add_argument("--config", type=str, default=None, help="training config.yml (default: next to the checkpoint)")
    p.add_argument("--vocoder_config", type=str, default=None, help="HiFi-GAN config.json (default: the V1 LJSpeech configuration)")
    p.add_argument("--vocoder_checkpoint", type=str, default=None, help='HiFi-GAN checkpoint with a "generator" state_dict')
    p.add_argument("--no_vocoder", action="store_true", help="write <id>_<step>.npy mel-spectrograms instead of wav files")
    p.add_argument("--batch_size", type=int, default=1, help="utterances per acoustic-model call (default 1, as the reference)")
    p.add_argument("--precision", type=str, default="bf16x3", choices=["bf16x3", "bf16"])
    p.add_argument("--verbose", type=int, default=1)
    return p


def _read_list(path: str, phn2idx) -> List[Tuple[str, torch.Tensor]]:
    items = []
    with open(path) as handle:
        for line in handle:
            line = line.strip()
            if not line:
                continue
            wav_path, text = line.split("|")[:2]
            utt = os.path.splitext(os.path.basename(wav_path))[0]
            items.append((utt, torch.tensor([phn2idx[p] for p in text.split()], dtype=torch.long)))
    return items


def _write_wav(path: str, samples: torch.Tensor) -> None:
    from scipy.io.wavfile import write
    pcm = (samples.clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16).cpu().numpy()
    write(path, SAMPLING_RATE, pcm)


def run_tts(args) -> float:
    level = {0: logging.WARNING, 1: logging.INFO}.get(args.verbose, logging.DEBUG)
    logging.basicConfig(level=level, stream=sys.stdout, force=True, format="%(asctime)s %(levelname)s %(name)s:%(lineno)d  %(message)s")
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X (gfx950) device visible: efficient_tts_amd has no CPU path")
    device = torch.device("cuda")
    os.makedirs(args.outdir, exist_ok=True)
    config_path = args.config or os.path.join(os.path.dirname(args.checkpoint), "config.yml")
    with open(config_path) as handle:
        config = yaml.safe_load(handle)
    data_params = config.get("dataset_params") or {}
    if not data_params.get("use_phnseq", False):
        raise NotImplementedError("only phoneme-sequence recipes (dataset_params.use_phnseq: true) are supported")
    with open(data_params["phnset_path"]) as handle:
        phn2idx = {p.strip(): i for i, p in enumerate(handle)}
    items = _read_list(args.test_fid_scp, phn2idx)
    logging.info(f"{len(items)} utterances to synthesise")
    step = os.path.basename(args.checkpoint).split("-")[-1][:-4]

    model = getattr(models, config["model_name"])(precision=args.precision, **config["model_params"])
    state = torch.load(args.checkpoint, map_location="cpu")
    model.load_state_dict(state["model"])
    model = model.to(device).eval()
    model.remove_weight_norm()
    vocoder = None
    if not args.no_vocoder:
        if args.vocoder_checkpoint:
            if not args.vocoder_config:
                raise ValueError("--vocoder_checkpoint needs --vocoder_config")
            vocoder = load_hifigan_generator(device, args.vocoder_config, args.vocoder_checkpoint, precision=args.precision)
        else:
            logging.warning("no --vocoder_checkpoint: the HiFi-GAN generator runs with RANDOM weights (timing / smoke only)")
            vocoder = HiFiGANGenerator(_V1, precision=args.precision).to(device).eval()
            vocoder.remove_weight_norm()

    total_rtf, done = 0.0, 0
    bs = max(1, int(args.batch_size))
    for lo in range(0, len(items), bs):
        chunk = items[lo:lo + bs]
        torch.cuda.synchronize()
        start = time.perf_counter()
        with torch.no_grad():
            if len(chunk) == 1:
                mel, _ = model.inference(chunk[0][1][None].to(device))
                mels = [mel[0]]
            else:
                lens = torch.tensor([len(t) for _, t in chunk])
                ids = torch.zeros(len(chunk), int(lens.max()), dtype=torch.long)
                for n, (_, t) in enumerate(chunk):
                    ids[n, :len(t)] = t
                mel, mel_lens, _ = model.inference_batch(ids.to(device), lens.to(device))
                mels = [mel[n, :int(mel_lens[n])] for n in range(len(chunk))]
            if vocoder is None:
                outs = mels
            elif len(chunk) == 1:
                outs = [vocoder(mels[0].t()[None].contiguous())[0, 0]]
            else:                                       # one batched generator pass; every item equals its single-utterance result
                audio = vocoder(mel.transpose(1, 2).contiguous(), mel_lens)
                outs = [audio[n, 0, :int(mel_lens[n]) * 256] for n in range(len(chunk))]
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
        seconds = sum(m.shape[0] for m in mels) * 256 / SAMPLING_RATE
        total_rtf += elapsed / seconds * len(chunk)
        done += len(chunk)
        for (utt, _), out in zip(chunk, outs):
            if vocoder is None:
                np.save(os.path.join(args.outdir, f"{utt}_{step}.npy"), out.cpu().numpy())
            else:
                _write_wav(os.path.join(args.outdir, f"{utt}_{step}.wav"), out)
        logging.debug(f"{[u for u, _ in chunk]}: {elapsed * 1e3:.2f} ms for {seconds:.2f} s of audio")
    rtf = total_rtf / max(done, 1)
    logging.info(f"Finished generation of {done} utterances (RTF = {rtf:.05f}).")
    return rtf


def main(argv=None) -> int:
    run_tts(get_parser().parse_args(argv))
    return 0


if __name__ == "__main__":
    sys.exit(main())
