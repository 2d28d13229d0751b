"""Log-mel front-end: oracle known-answer checks on CPU, HIP parity on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import logmel_oracle as LO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "logmel_small.npz")
LOGMEL_TOL = 1e-3        # |log-mel (HIP, bf16x3 MFMA DFT) - log-mel (oracle, fp32 torch.stft)|; measured 1.6e-4


def test_mel_basis_known_answers():
    """Properties of librosa's Slaney filterbank for (22050, 1024, 80, 0, 8000) that do not depend on our code:
    shape, unit area in Hz per filter, contiguous triangles, every rFFT bin below fmax in at most two filters,
    linear (equal-width) filters below 1 kHz, nothing above fmax."""
    b = LO.slaney_mel_basis()
    assert b.shape == (80, 513) and b.dtype == np.float32
    hz_per_bin = 22050 / 1024
    area = b.sum(1) * hz_per_bin
    assert np.allclose(area[5:], 1.0, atol=0.08)                   # slaney norm: unit area (bin sampling makes it approximate)
    assert ((b > 0).sum(0) <= 2).all()
    fmax_bin = int(np.floor(8000 / hz_per_bin))
    assert (b[:, fmax_bin + 1:] == 0).all()
    peaks = b.argmax(1)
    assert (np.diff(peaks) >= 1).all()
    lin = peaks[(peaks * hz_per_bin) < 900]
    assert np.ptp(np.diff(lin)) <= 1                               # equally spaced centres on the linear part
    for m in range(80):                                            # one contiguous run of non-zeros per filter
        nz = np.nonzero(b[m])[0]
        assert nz.size and (np.diff(nz) == 1).all()


def test_mel_basis_matches_independent_slaney_filterbank():
    """f-3 pin: the oracle's filterbank against tests/golden/mel_basis_hf.npz, produced by an implementation outside this repo
    (transformers.audio_utils.mel_filter_bank, the librosa.filters.mel stand-in of the Whisper feature extractor;
    tools/gen_golden_melbasis.py), and against that implementation live where it is installed."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mel_basis_hf.npz"))
    assert g["consts"].tolist() == [LO.SR, LO.N_FFT, LO.N_MELS, LO.FMIN, LO.FMAX]
    b = LO.slaney_mel_basis()
    assert b.shape == g["basis"].shape
    assert np.abs(b - g["basis"]).max() < 2e-9                       # float32 rounding of weights <= 0.0265 (measured 9.2e-10)
    assert ((b > 0) == (g["basis"] > 1e-12)).all()                   # same support, bin for bin
    try:
        from transformers.audio_utils import mel_filter_bank
    except Exception:
        return
    live = mel_filter_bank(LO.N_FFT // 2 + 1, LO.N_MELS, LO.FMIN, LO.FMAX, LO.SR, norm="slaney", mel_scale="slaney").T
    assert np.abs(live - g["basis"]).max() < 1e-12


def test_oracle_matches_committed_fixture_and_naive_dft():
    g = np.load(GOLD)
    audio = torch.from_numpy(g["audio"]).float() / 32768.0
    mel, frames = LO.batch_logmel(audio, torch.from_numpy(g["lengths"]))
    assert frames.tolist() == g["frames"].tolist() == [int(l) // 256 for l in g["lengths"]]
    assert np.abs(mel.numpy() - g["mel"]).max() < 1e-5
    # the STFT leg against a float64 DFT written out by hand, one frame of item 0
    y = audio[0, :int(g["lengths"][0])].double()
    yp = torch.nn.functional.pad(y[None, None], (384, 384), mode="reflect")[0, 0]
    t = 5
    fr = yp[t * 256: t * 256 + 1024] * torch.hann_window(1024, dtype=torch.float64)
    k = torch.arange(1024, dtype=torch.float64)
    f = torch.arange(513, dtype=torch.float64)[:, None]
    re = (fr * torch.cos(2 * np.pi * f * k / 1024)).sum(1)
    im = -(fr * torch.sin(2 * np.pi * f * k / 1024)).sum(1)
    mag = torch.sqrt(re * re + im * im + 1e-9)
    ref = torch.log(torch.clamp(torch.from_numpy(LO.slaney_mel_basis()).double() @ mag, min=1e-5))
    assert (mel[0, t].double() - ref).abs().max() < 2e-4
    assert (mel[2, int(frames[2]):] == 0).all()                    # collate zero padding after the log


@pytest.mark.gpu
def test_frontend_matches_oracle_on_fixture():
    from efficient_tts_amd.frontend import LogMelFrontend
    g = np.load(GOLD)
    fe = LogMelFrontend("cuda:0")
    a16 = torch.from_numpy(g["audio"])
    lengths = torch.from_numpy(g["lengths"])
    mel, frames = fe(a16, lengths)                                  # int16 in, scaled by 1/32768 like TextMelLoader.get_mel
    assert frames.tolist() == g["frames"].tolist()
    assert mel.shape == g["mel"].shape
    err = np.abs(mel.cpu().numpy() - g["mel"]).max()
    assert err < LOGMEL_TOL, err
    mel2, _ = fe(a16.float().cuda() / 32768.0, lengths.cuda())      # float input already on the device
    assert torch.equal(mel, mel2)


@pytest.mark.gpu
def test_frontend_edge_cases_and_model_handoff():
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.frontend import LogMelFrontend
    fe = LogMelFrontend("cuda:0")
    g = torch.Generator().manual_seed(3)
    # ragged batch incl. the shortest legal item (one frame more than the reflect padding allows) and a non-multiple of hop
    lengths = torch.tensor([385 + 256, 256 * 7 + 255, 256 * 40, 1000])
    audio = (torch.rand(4, int(lengths.max()), generator=g) * 2 - 1) * 0.5
    for b, l in enumerate(lengths):
        audio[b, int(l):] = 123.0                                   # garbage past the length must never be read
    mel, frames = fe(audio, lengths)
    ref, rfr = LO.batch_logmel(audio, lengths)
    assert frames.tolist() == rfr.tolist() == [2, 7, 40, 3]
    assert (mel.cpu() - ref).abs().max() < LOGMEL_TOL
    with pytest.raises(ValueError):
        fe(audio[:, :300], torch.tensor([300, 300, 300, 300]))      # not longer than the reflect padding
    with pytest.raises(ValueError):
        fe(audio, lengths + 100000)
    # the output IS the model's (speech, speech_lengths)
    torch.manual_seed(0)
    model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True).cuda().eval()
    text = torch.randint(1, 76, (4, 12), device="cuda")
    tl = torch.tensor([12, 10, 12, 6], device="cuda")
    # forward wants lengths sorted like the collate does; only shapes / finiteness are checked here
    with torch.no_grad():
        out = model(text, tl, mel, frames)
    assert torch.isfinite(out[0]).all() and out[4].shape == mel.shape


@pytest.mark.gpu
@pytest.mark.parametrize("radix", [0, 1, 2, 4, 8, 16])
def test_frontend_radix_split_dft_equals_the_dense_product(radix):
    """round 6: the DFT split by decimation in time (`radix` real DFTs of 1024 / radix points as one batched contraction, recombined in the
    logmel kernel) against the oracle's torch.stft, for every radix incl. the dense 1024-point product (radix 1) -- same tolerance -- and
    against the dense product's own output (summation order only: well inside the tolerance)."""
    from efficient_tts_amd.frontend import LogMelFrontend
    g = np.load(GOLD)
    a16, lengths = torch.from_numpy(g["audio"]), torch.from_numpy(g["lengths"])
    mel, frames = LogMelFrontend("cuda:0", radix=radix)(a16, lengths)
    assert frames.tolist() == g["frames"].tolist()
    err = np.abs(mel.cpu().numpy() - g["mel"]).max()
    print(f"radix {radix}: log-mel max-abs vs the oracle fixture {err:.2e}")
    assert err < LOGMEL_TOL, err
    dense, _ = LogMelFrontend("cuda:0", radix=1)(a16, lengths)
    assert float((mel - dense).abs().max()) < LOGMEL_TOL / 2
    # ragged edge cases through the split path: shortest legal item, non-multiple of hop, garbage past the length
    gen = torch.Generator().manual_seed(3)
    lengths = torch.tensor([385 + 256, 256 * 7 + 255, 256 * 40, 1000])
    audio = (torch.rand(4, int(lengths.max()), generator=gen) * 2 - 1) * 0.5
    for b, l in enumerate(lengths):
        audio[b, int(l):] = 123.0
    mel, frames = LogMelFrontend("cuda:0", radix=radix)(audio, lengths)
    ref, rfr = LO.batch_logmel(audio, lengths)
    assert frames.tolist() == rfr.tolist() and float((mel.cpu() - ref).abs().max()) < LOGMEL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("num_mels,fmax", [(80, 8000.0), (64, 11025.0), (40, 4000.0), (72, 7600.0)])
def test_fused_fft_launch_on_other_filterbanks_and_frame_counts(num_mels, fmax):
    """efts_logmel_fft (radix 0) against the MFMA pipeline (radix 4, itself pinned to the oracle above) on filterbanks whose span lengths differ
    from the reference's -- fewer than 64 filters (no four-lane filters at all), wider triangles, a lower fmax -- and on odd / even frame counts,
    a batch padded past every item's length (`max_frames`), one item.  n_mels > 80 is refused by the fused launch (the front-end then takes the
    split MFMA product by itself)."""
    from efficient_tts_amd.frontend import LogMelFrontend
    gen = torch.Generator().manual_seed(num_mels)
    lengths = torch.tensor([256 * 9, 256 * 12 + 17, 700, 256 * 31])
    audio = (torch.rand(4, int(lengths.max()), generator=gen) * 2 - 1) * 0.4
    for b, l in enumerate(lengths):
        audio[b, int(l):] = float("nan")                          # never read past an item's length
    f0 = LogMelFrontend("cuda:0", num_mels=num_mels, fmax=fmax)
    f4 = LogMelFrontend("cuda:0", num_mels=num_mels, fmax=fmax, radix=4)
    assert f0.radix == 0
    for mf in (None, 35):
        m0, fr0 = f0(audio, lengths, max_frames=mf)
        m4, fr4 = f4(audio, lengths, max_frames=mf)
        assert m0.shape == m4.shape and fr0.tolist() == fr4.tolist() == [9, 12, 2, 31]
        assert torch.isfinite(m0).all()
        assert float((m0 - m4).abs().max()) < LOGMEL_TOL
        for b, n in enumerate(fr0.tolist()):
            assert float(m0[b, n:].abs().max() if n < m0.shape[1] else 0.0) == 0.0
    one, _ = f0(audio[:1], lengths[:1])
    assert torch.equal(one[0], f0(audio, lengths)[0][0, :9])
    assert LogMelFrontend("cuda:0", num_mels=96).radix == 4      # the fused launch holds at most 80 filters
    with pytest.raises(ValueError):
        LogMelFrontend("cuda:0", num_mels=96, radix=0)
