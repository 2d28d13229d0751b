"""CPU oracle of the log-mel front-end that feeds the EFTS-CNN path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
the product (efficient_tts_amd/) never imports this module.

Follows nntts/datasets/meldataset.py:49-82 (`mel_spectrogram`): reflect-pad (n_fft - hop)/2 = 384 samples
on both sides (:69), torch.stft n_fft 1024 / hop 256 / win 1024 / periodic hann, center=False (:72-73),
magnitude sqrt(re^2 + im^2 + 1e-9) (:75), 80-bin mel projection (:77), log(clamp(., 1e-5)) (:78, :27-28),
and TextMelCollate's zero padding of the time axis AFTER the log (nntts/datasets/taco2_data.py:122-134).

Mel filterbank -- NOT pinned by the reference itself, pinned by an independent implementation: the reference takes it from `librosa.filters.mel(22050, 1024, 80, 0, 8000)`
(meldataset.py:9,65; setup.py pins librosa>=0.8.0), a third-party dependency that is not vendored in
/root/reference and not installed here.  `slaney_mel_basis` restates librosa 0.8's published algorithm
(htk=False Slaney mel scale: linear below 1 kHz at 200/3 Hz per mel, log above with step log(6.4)/27;
triangular filters on the FFT bin centre frequencies; norm='slaney': each filter scaled by
2 / (f[m+2] - f[m])).  It is checked bin for bin (max |diff| 9.2e-10, identical support) against
tests/golden/mel_basis_hf.npz, the same filterbank produced by Hugging Face transformers'
`audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` -- the librosa.filters.mel replacement of the Whisper
feature extractor -- by tools/gen_golden_melbasis.py (tests/test_frontend.py).  librosa's own output remains unavailable
here, so this is a pin by a second independent implementation of the published algorithm, not by the reference's
dependency.  The STFT / magnitude / log part is pinned by construction: it IS torch.stft, the call
the reference makes (with return_complex=True, which the reference's torch predates).
"""
from __future__ import annotations

import numpy as np
import torch

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 22050, 1024, 256, 1024, 80, 0.0, 8000.0
PAD = (N_FFT - HOP) // 2


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_basis(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm='slaney') -> float32 [n_mels, n_fft//2+1]"""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def mel_spectrogram(y: torch.Tensor) -> torch.Tensor:
    """y [B, L] float32 in [-1, 1] -> log-mel [B, 80, L // 256]   (meldataset.py:49-82)"""
    basis = torch.from_numpy(slaney_mel_basis())
    y = torch.nn.functional.pad(y.unsqueeze(1), (PAD, PAD), mode="reflect").squeeze(1)              # :69-70
    spec = torch.stft(y, N_FFT, hop_length=HOP, win_length=WIN, window=torch.hann_window(WIN), center=False,
                      normalized=False, onesided=True, return_complex=True)                               # :72-73
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)                                         # :75
    mel = torch.matmul(basis, mag)                                                                        # :77
    return torch.log(torch.clamp(mel, min=1e-5))                                                          # :78, :27-28


def batch_logmel(audio: torch.Tensor, lengths: torch.Tensor):
    """What TextMelLoader.get_mel + TextMelCollate produce for a batch (taco2_data.py:66-76, :122-139):
    each item's own length is used for its reflect padding; the batch is zero-padded on the time axis and
    returned as [B, T, 80] with the frame counts.  audio [B, Lmax] float32 (int16 / 32768), lengths [B]."""
    mels = [mel_spectrogram(audio[b:b + 1, :int(lengths[b])])[0] for b in range(audio.shape[0])]
    T = max(m.shape[1] for m in mels)
    out = torch.zeros(audio.shape[0], T, N_MELS)
    for b, m in enumerate(mels):
        out[b, :m.shape[1]] = m.t()
    return out, torch.tensor([m.shape[1] for m in mels], dtype=torch.int64)
