"""Writes tests/golden/hifigan_small.npz from the REFERENCE generator (imported from /root/reference, run in the
build container only): the reference Generator with the oracle's name-keyed parameter fill, two mel inputs."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from oracle import hifigan_oracle as HO
from nntts.vocoders.hifigan_model import Generator          # noqa: E402
from nntts.vocoders.env import AttrDict                     # noqa: E402

if __name__ == "__main__":
    h = AttrDict(json.load(open("/root/reference/nntts/vocoders/HiFiGAN_LJ_V1/config.json")))
    gen = Generator(h).eval()
    P = HO.fill_params()
    sd = gen.state_dict()
    assert list(sd.keys()) == list(P.keys()), "oracle.param_shapes() must list the reference's keys in order"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    gen.load_state_dict(P)
    out = {}
    for name, T, seed in (("a", 12, 1), ("b", 37, 2)):
        mel = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(seed)) * 1.5 - 4.0       # log-mel-like range
        with torch.no_grad():
            y = gen(mel)
            yo = HO.forward(P, mel)
        print(name, tuple(y.shape), "ref max", float(y.abs().max()), "oracle vs reference", float((y - yo).abs().max()))
        out[f"mel_{name}"] = mel.numpy()
        out[f"audio_{name}"] = y.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hifigan_small.npz"), **out)
    print("wrote hifigan_small.npz")
