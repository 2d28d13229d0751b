"""Writes tests/golden/logmel_small.npz: a small int16 batch and the log-mel the CPU oracle
(oracle/logmel_oracle.py) computes for it.  NOTE: unlike the other goldens this one does NOT come from
the imported reference: nntts/datasets/meldataset.py imports librosa at module level (absent here) and
calls torch.stft without return_complex (rejected by this torch), so the oracle's restatement is the
source; its mel filterbank is pinned by tests/golden/mel_basis_hf.npz (tools/gen_golden_melbasis.py; see the oracle header)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import logmel_oracle as O


def synth_audio(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L, dtype=torch.float64) / 22050
    out = []
    for b in range(B):
        f0 = 90 + 140 * torch.rand(1, generator=g).item()
        y = torch.zeros(L, dtype=torch.float64)
        for h in range(1, 30):
            y += torch.sin(2 * np.pi * f0 * h * t + 6.28 * torch.rand(1, generator=g).item()) / h
        env = (0.5 + 0.5 * torch.sin(2 * np.pi * 3.1 * t + b)).clamp(min=0) ** 2
        y = 0.25 * y * env + 0.01 * torch.randn(L, generator=g, dtype=torch.float64)
        y[L // 3: L // 3 + 3000] = 0.0003 * torch.randn(3000, generator=g, dtype=torch.float64)   # near-silence
        y[L // 2: L // 2 + 2000] = 0.0                                                             # digital silence
        out.append(y)
    return torch.round(torch.stack(out).clamp(-1, 1) * 32767).to(torch.int16)


if __name__ == "__main__":
    a16 = synth_audio(3, 20000, 7)
    lengths = torch.tensor([20000, 14111, 6400])
    mel, frames = O.batch_logmel(a16.float() / 32768.0, lengths)
    basis = O.slaney_mel_basis()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "logmel_small.npz"), audio=a16.numpy(), lengths=lengths.numpy(),
                        mel=mel.numpy().astype(np.float32), frames=frames.numpy(),
                        basis_rowsum=basis.sum(1), basis_argmax=basis.argmax(1).astype(np.int32), basis_max=basis.max(1))
    print("wrote logmel_small.npz", mel.shape, frames.tolist())
