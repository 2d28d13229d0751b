"""Row f-2: datasets / collate / command line of the reference recipe (nntts/bin/train.py, nntts/datasets/taco2_data.py)."""
import os

import numpy as np
import pytest
import torch
import yaml

from efficient_tts_amd import datasets as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collate_layout_matches_reference_contract():
    """taco2_data.py:107-139: sort by decreasing text length, right zero-pad, LongTensor lengths."""
    g = torch.Generator().manual_seed(0)
    items = []
    for n, L in ((3, 700), (7, 500), (5, 900), (7, 650)):
        items.append((torch.randint(1, 76, (n,), generator=g), torch.randint(-3000, 3000, (L,), generator=g).to(torch.int16)))
    text, tl, audio, al = D.TextMelCollate()(items)
    assert text.dtype == torch.long and tl.dtype == torch.long and al.dtype == torch.long and audio.dtype == torch.int16
    assert tl.tolist() == [7, 7, 5, 3]
    order = [1, 3, 2, 0]                      # torch.sort(descending) is what the reference uses; ties keep torch's order
    assert sorted(al.tolist()) == sorted([700, 500, 900, 650])
    for i in range(4):
        j = [k for k in range(4) if len(items[k][0]) == tl[i] and items[k][1].shape[0] == al[i]][0]
        assert torch.equal(text[i, :tl[i]], items[j][0]) and (text[i, tl[i]:] == 0).all()
        assert torch.equal(audio[i, :al[i]], items[j][1]) and (audio[i, al[i]:] == 0).all()
    assert text.shape == (4, 7) and audio.shape == (4, 900)
    with pytest.raises(NotImplementedError):
        D.TextMelCollate(n_frames_per_step=2)


def test_text_mel_loader_reads_wavs_and_phonemes(tmp_path):
    from scipy.io.wavfile import write
    wavs = tmp_path / "wavs"
    wavs.mkdir()
    rng = np.random.default_rng(0)
    lines = []
    for i, (phones, n) in enumerate((("HH AH0 L OW1", 3000), ("W ER1 L D", 2500), ("AH0", 1200))):
        write(str(wavs / f"u{i}.wav"), 22050, rng.integers(-2000, 2000, n).astype(np.int16))
        lines.append(f"DUMMY/u{i}.wav|{phones}")
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "phn.txt").write_text("\n".join(["_", "HH", "AH0", "L", "OW1", "W", "ER1", "D"]) + "\n")
    ds = D.TextMelLoader(str(tmp_path / "list.txt"), wav_path=str(wavs), use_phnseq=True, phnset_path=str(tmp_path / "phn.txt"))
    assert len(ds) == 3 and len(ds.phn_map) == 8
    seen = {}
    for k in range(3):
        t, a = ds[k]
        assert a.dtype == torch.int16 and t.dtype == torch.long
        seen[a.shape[0]] = t.tolist()
    assert seen == {3000: [1, 2, 3, 4], 2500: [5, 6, 3, 7], 1200: [2]}
    with pytest.raises(NotImplementedError):
        D.TextMelLoader(str(tmp_path / "list.txt"), wav_path=str(wavs))
    write(str(wavs / "u0.wav"), 16000, np.zeros(100, np.int16))
    with pytest.raises(ValueError):
        [ds[k] for k in range(3)]


def test_cli_parser_accepts_the_reference_command_line():
    from efficient_tts_amd.bin.train import get_parser
    a = get_parser().parse_args(["--train_fid_scp", "tr", "--dev_fid_scp", "dv", "--outdir", "o", "--config", "c.yaml",
                                 "--resume", "ck", "--verbose", "0", "--local_rank", "3"])
    assert (a.train_fid_scp, a.dev_fid_scp, a.outdir, a.config, a.resume, a.pretrain, a.verbose, a.rank) == \
        ("tr", "dv", "o", "c.yaml", "ck", "", 0, 3)


def _write_config(path, **over):
    cfg = dict(dataset_type="SyntheticTextAudio", dataset_params=dict(n_items=12, min_phones=8, max_phones=16),
               collate_fn_type="TextMelCollate", model_name="EfficientTTSCNN",
               model_params=dict(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01),
               batch_size=4, pin_memory=False, num_workers=0, optimizer_type="Adam",
               optimizer_params=dict(lr=1.0e-3, betas=[0.9, 0.99], eps=1.0e-9, weight_decay=1.0e-5, amsgrad=True), grad_norm=1.0,
               scheduler_type="WarmupLR", scheduler_params=dict(warmup_steps=4000), train_max_steps=6, save_interval_steps=3,
               eval_interval_steps=3, log_interval_steps=2, bucket_frames=32, bucket_phones=8)
    cfg.update(over)
    with open(path, "w") as f:
        yaml.dump(cfg, f)


@pytest.mark.gpu
def test_cli_trains_saves_and_resumes(tmp_path):
    """python -m efficient_tts_amd.bin.train with the reference's YAML keys: config.yml written, checkpoints carry the
    reference's keys (trainer.py:78-97), --resume continues from the saved step count."""
    from efficient_tts_amd.bin.train import main
    cfg = tmp_path / "c.yaml"
    out = tmp_path / "exp"
    _write_config(str(cfg))
    assert main(["--outdir", str(out), "--config", str(cfg), "--verbose", "0"]) == 0
    saved = yaml.load(open(out / "config.yml"), Loader=yaml.Loader)
    assert saved["model_name"] == "EfficientTTSCNN" and saved["outdir"] == str(out) and saved["distributed"] is False
    ck3, ck6 = out / "checkpoint-3steps.pkl", out / "checkpoint-6steps.pkl"
    assert ck3.exists() and ck6.exists()
    sd = torch.load(ck6, map_location="cpu")
    assert set(sd) == {"model", "optimizer", "scheduler", "steps", "epochs"} and sd["steps"] == 6
    from efficient_tts_amd import EfficientTTSCNN
    ref_keys = list(EfficientTTSCNN(num_symbols=76, use_masking=True).state_dict().keys())
    assert list(sd["model"].keys()) == ref_keys
    assert all(torch.isfinite(v).all() for v in sd["model"].values())
    a = torch.load(ck3, map_location="cpu")["model"]
    assert any(not torch.equal(a[k], sd["model"][k]) for k in ref_keys)            # parameters moved between the checkpoints
    # resume from step 3 and run to step 6 again: identical data order (seeded shuffle is per process) is not
    # guaranteed, so only the bookkeeping is asserted
    out2 = tmp_path / "exp2"
    assert main(["--outdir", str(out2), "--config", str(cfg), "--verbose", "0", "--resume", str(ck3)]) == 0
    sd2 = torch.load(out2 / "checkpoint-6steps.pkl", map_location="cpu")
    assert sd2["steps"] == 6 and not (out2 / "checkpoint-3steps.pkl").exists()


@pytest.mark.gpu
def test_inference_cli_writes_wavs(tmp_path):
    """python -m efficient_tts_amd.bin.inference with the reference's arguments (nntts/bin/inference.py:128-176): wav files
    of T2 * 256 samples per utterance, the batched mode gives the same audio, --no_vocoder writes the mels."""
    from scipy.io.wavfile import read
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.bin.inference import main
    exp = tmp_path / "exp"
    exp.mkdir()
    phones = ["_"] + [f"P{i}" for i in range(1, 76)]
    (tmp_path / "phn.txt").write_text("\n".join(phones) + "\n")
    rng = np.random.default_rng(1)
    lines = [f"DUMMY/utt{n}.wav|" + " ".join(phones[int(i)] for i in rng.integers(1, 76, size=k)) for n, k in enumerate((9, 14, 11))]
    (tmp_path / "test.txt").write_text("\n".join(lines) + "\n")
    params = dict(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01)
    with open(exp / "config.yml", "w") as f:
        yaml.dump(dict(model_name="EfficientTTSCNN", model_params=params,
                       dataset_params=dict(use_phnseq=True, phnset_path=str(tmp_path / "phn.txt"))), f)
    torch.manual_seed(0)
    m = EfficientTTSCNN(**params)
    with torch.no_grad():
        m.duration_predictor.linear.bias.fill_(1.5)              # a few frames per phoneme with random weights
    torch.save({"model": m.state_dict(), "steps": 7}, exp / "checkpoint-7steps.pkl")
    base = ["--checkpoint", str(exp / "checkpoint-7steps.pkl"), "--test_fid_scp", str(tmp_path / "test.txt"), "--verbose", "0"]
    torch.manual_seed(0)
    assert main(base + ["--outdir", str(tmp_path / "w1")]) == 0
    torch.manual_seed(0)                                           # same random vocoder weights for the batched run
    assert main(base + ["--outdir", str(tmp_path / "w3"), "--batch_size", "3"]) == 0
    assert main(base + ["--outdir", str(tmp_path / "mel"), "--no_vocoder"]) == 0
    for n in range(3):
        sr, a = read(str(tmp_path / "w1" / f"utt{n}_7steps.wav"))
        _, b = read(str(tmp_path / "w3" / f"utt{n}_7steps.wav"))
        mel = np.load(tmp_path / "mel" / f"utt{n}_7steps.npy")
        assert sr == 22050 and a.dtype == np.int16 and a.shape == (mel.shape[0] * 256,) and mel.shape[1] == 80
        assert a.shape == b.shape and np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 2


def test_inference_cli_parser_and_list_reader(tmp_path):
    """Host side of the synthesis script (nntts/bin/inference.py:128-176, :88-101): arguments and the `wav|phonemes` list."""
    from efficient_tts_amd.bin import inference as I
    a = I.get_parser().parse_args(["--checkpoint", "exp/checkpoint-7steps.pkl", "--test_fid_scp", "t.txt", "--outdir", "o",
                                   "--batch_size", "4", "--no_vocoder"])
    assert a.checkpoint.endswith("7steps.pkl") and a.batch_size == 4 and a.no_vocoder and a.precision == "bf16x3" and a.config is None
    lst = tmp_path / "t.txt"
    lst.write_text("wavs/LJ001-0001.wav|HH AH0 L OW1\n\nwavs/LJ001-0002.wav|W ER1 L D|extra field\n")
    phn2idx = {p: i for i, p in enumerate(["AH0", "D", "ER1", "HH", "L", "OW1", "W"])}
    items = I._read_list(str(lst), phn2idx)
    assert [u for u, _ in items] == ["LJ001-0001", "LJ001-0002"]
    assert items[0][1].tolist() == [3, 0, 4, 5] and items[1][1].tolist() == [6, 2, 4, 1] and items[0][1].dtype == torch.long
    with pytest.raises(KeyError):
        I._read_list(str(lst), {"HH": 0})


def test_trainer_intervals_and_meters():
    """Host bookkeeping of the trainer (efficient_tts_trainer.py:147-149, :236-262): interval tests and running means."""
    from efficient_tts_amd.trainer import EfficientTTSTrainer, _Meter
    cfg = dict(outdir="/tmp", log_interval_steps=5, eval_interval_steps=0, save_interval_steps=10, train_max_steps=20, grad_norm=1.0)
    t = EfficientTTSTrainer(steps=0, epochs=0, data_loader={}, sampler={}, model=torch.nn.Linear(2, 2), optimizer=None, scheduler=None, config=cfg)
    hits = {k: [] for k in ("log_interval_steps", "eval_interval_steps", "save_interval_steps")}
    for step in range(1, 21):
        t.steps = step
        for k in hits:
            if t._due(k):
                hits[k].append(step)
    assert hits == {"log_interval_steps": [5, 10, 15, 20], "eval_interval_steps": [], "save_interval_steps": [10, 20]}
    m = _Meter("train")
    for v in (1.0, 3.0):
        m.add({"loss": v, "mel_loss": v / 2, "duration_loss": v / 4})
    assert m.means(2) == {"train/loss": 2.0, "train/mel_loss": 1.0, "train/dur_loss": 0.5}
    m.reset()
    assert m.means(0) == {"train/loss": 0.0, "train/mel_loss": 0.0, "train/dur_loss": 0.0}
