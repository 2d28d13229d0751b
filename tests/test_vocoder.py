"""HiFi-GAN V1 generator: oracle vs the reference's golden audio on CPU, HIP parity on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as HO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hifigan_small.npz")
CFG = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
           resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=80)
AUDIO_TOL = 1e-3          # |audio (HIP, bf16x3) - audio (reference, fp32)|, samples in [-1, 1]


def test_oracle_matches_reference_golden():
    g = np.load(GOLD)
    P = HO.fill_params()
    for name in ("a", "b"):
        y = HO.forward(P, torch.from_numpy(g[f"mel_{name}"]))
        assert y.shape == g[f"audio_{name}"].shape == (1, 1, g[f"mel_{name}"].shape[2] * 256)
        assert np.abs(y.numpy() - g[f"audio_{name}"]).max() < 2e-5


def test_module_has_the_reference_state_dict():
    from efficient_tts_amd.vocoder import HiFiGANGenerator
    m = HiFiGANGenerator(CFG)
    sd = m.state_dict()
    ref = HO.param_shapes()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    m.load_state_dict(HO.fill_params())
    m.remove_weight_norm()
    assert "conv_pre.weight" in m.state_dict() and "conv_pre.weight_g" not in m.state_dict()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 80, 4))                       # no CPU path
    with pytest.raises(NotImplementedError):
        HiFiGANGenerator(dict(CFG, resblock="2"))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("bf16x3", AUDIO_TOL), ("bf16", 6e-2)])
def test_generator_matches_reference_golden(precision, tol):
    from efficient_tts_amd.vocoder import HiFiGANGenerator
    g = np.load(GOLD)
    m = HiFiGANGenerator(CFG, precision=precision)
    m.load_state_dict(HO.fill_params())
    m = m.cuda().eval()
    for name in ("a", "b"):
        y = m(torch.from_numpy(g[f"mel_{name}"]).cuda())
        assert y.shape == g[f"audio_{name}"].shape
        err = np.abs(y.cpu().numpy() - g[f"audio_{name}"]).max()
        print(precision, name, "max abs err", err)
        assert err <= tol, (precision, name, err)
    # weight norm removed -> same audio; a batch of two -> each item as alone
    mel = torch.from_numpy(np.concatenate([g["mel_a"], g["mel_a"][:, :, ::-1].copy()], 0)).cuda()
    y2 = m(mel)
    y1 = m(mel[1:2])
    assert torch.equal(y2[1], y1[0])
    m.remove_weight_norm()
    y3 = m(mel[:1])
    assert (y3 - y2[:1]).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_ragged_batch_equals_single_utterances(precision):
    """`forward(mel, lengths)`: the items of a padded batch share one row space (zero rows between them, a validity mask
    per stage); each must come out exactly as if it had been synthesised alone without its padding, and be silent past
    its own length -- whatever the padding frames contain."""
    from efficient_tts_amd.vocoder import HiFiGANGenerator
    m = HiFiGANGenerator(CFG, precision=precision)
    m.load_state_dict(HO.fill_params())
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(7)
    lens = [37, 12, 50, 1]
    T = max(lens)
    mel = torch.randn(len(lens), 80, T, generator=g)              # padding frames are NOT zero: they must not matter
    y = m(mel.cuda(), torch.tensor(lens))
    assert y.shape == (len(lens), 1, T * 256)
    for b, n in enumerate(lens):
        alone = m(mel[b:b + 1, :, :n].contiguous().cuda())
        assert torch.equal(y[b, :, :n * 256], alone[0]), (precision, b)
        assert float(y[b, :, n * 256:].abs().max()) == 0.0 if n < T else True
