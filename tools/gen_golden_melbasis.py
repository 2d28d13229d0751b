"""Pin the oracle's Slaney mel filterbank against an implementation that is independent of
this repo: Hugging Face `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")`, the filterbank the
Whisper feature extractor uses in place of `librosa.filters.mel` (librosa itself is a pip dependency of the reference,
`nntts/datasets/audio_processor.py`, and is not in this image).  Writes tests/golden/mel_basis_hf.npz: the float64
filterbank for the reference's front-end constants (22050 Hz, n_fft 1024, 80 mels, 0..8000 Hz) and the package version
that produced it.  Run here (CPU); the fixture travels, transformers need not."""
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank

SR, N_FFT, N_MELS, FMIN, FMAX = 22050, 1024, 80, 0.0, 8000.0

if __name__ == "__main__":
    fb = mel_filter_bank(N_FFT // 2 + 1, N_MELS, FMIN, FMAX, SR, norm="slaney", mel_scale="slaney").T   # [n_mels, n_fft//2+1]
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mel_basis_hf.npz")
    np.savez_compressed(out, basis=fb.astype(np.float64), consts=np.array([SR, N_FFT, N_MELS, FMIN, FMAX]),
                        source=np.array(f"transformers {transformers.__version__} audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')"))
    print(out, fb.shape, os.path.getsize(out))
