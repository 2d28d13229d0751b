"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol of include/efts_abi.h.
No compute is launched (there is no GPU in the build container)."""
import os
import re

import pytest

from efficient_tts_amd import build as B
from efficient_tts_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build(verbose=False)
    return L.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "efts_abi.h")).read()
    declared = set(re.findall(r"\b(efts_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in efts_abi.h but not exported by libefts_hip.so"
    assert declared == set(L.exported_symbols()), (declared ^ set(L.exported_symbols()))


def test_version_and_argument_errors(lib):
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "efts_abi.h")).read()
    declared = int(re.search(r"#define EFTS_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.efts_version() == declared == L.ABI_VERSION      # library, header and ctypes mirror are one revision (load() refuses any other)
    g = L.GemmArgs()
    g.split, g.taps = 7, 5
    assert lib.efts_gemm(g, None) == -1            # EFTS_EINVAL before any launch
    assert b"split" in lib.efts_last_error()
    with pytest.raises(ValueError):
        L.check(lib.efts_gemm(g, None), "efts_gemm")


def test_args_layouts_match_header():
    """sizeof and field offsets of every argument struct of the ABI as compiled by gcc == the ctypes mirrors."""
    import subprocess, tempfile, ctypes
    structs = {
        "efts_gemm_args": (L.GemmArgs, ["out_bf16_lo", "tiling", "sign_mask", "soft_index", "key_len", "query_len", "drop_p", "drop_seed", "sqerr_target", "ld_target", "target_batch_stride", "sqerr_part"]),
        "efts_resconv5_args": (L.ResConv5Args, ["x", "x_lo", "x_f32", "w", "split", "rowmask", "y_f32", "y", "y_lo", "y_split", "plan", "taps", "no_residual", "sign_bits", "act_bwd_sign", "act_bwd_bias_part", "act_bwd_bias_rows", "act_bwd_slope", "kernel"]),
        "efts_frame_linear_args": (L.FrameLinearArgs, ["x", "w", "bias", "act", "B", "n", "y_f32", "y", "y_lo", "ldy", "y_split", "max_workgroups"]),
        "efts_wgrad_item": (L.WgradItem, ["dz_plane", "ldz", "x_plane", "ldx", "v", "g", "dw_or_dv", "dg", "bias_part", "dbias", "nparts", "reserved"]),
        "efts_expand_args": (L.ExpandArgs, ["e", "text_len", "mel_len", "sigma", "v", "ldv", "B", "n", "alpha_out", "y_f32", "ldo", "y", "y_lo", "ldy", "y_split"]),
    }
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "efts_abi.h"\nint main(){\n'
    want = []
    for name, (cls, fields) in structs.items():
        src += f'printf(" %zu", sizeof({name}));' + "".join(f'printf(" %zu", offsetof({name}, {f}));' for f in fields) + "\n"
        want += [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        got = [int(v) for v in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert got == want


def test_model_state_dict_keys_match_reference_layout():
    """state_dict names/shapes are the drop-in contract; checked against the oracle's table."""
    import torch
    from efficient_tts_amd import EfficientTTSCNN
    from oracle import efts_oracle as O
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01)
    sd = m.state_dict()
    shapes = O.param_shapes()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k
    m.load_state_dict(O.fill_params())
    m.remove_weight_norm()
    assert "decoder.layers.0.conv.0.weight" in m.state_dict()
    assert sum(p.numel() for p in EfficientTTSCNN(76, use_masking=True).parameters()) == 20587601


def test_model_fails_loudly_without_gpu():
    import torch
    from efficient_tts_amd import EfficientTTSCNN
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = EfficientTTSCNN(num_symbols=76, use_masking=True)
    with pytest.raises(Exception):
        m.inference(torch.zeros(1, 8, dtype=torch.long))


def test_train_mode_forward_without_gradients_reaches_the_device_path():
    """The reference applies Dropout inside ResConv1d and the prenet for dropout_rate > 0 (ctor default 0.1,
    nntts/layers/efts_modules.py:38-47, efficient_tts.py:76-80) in EVERY train()-mode forward.  Since round 4 the gradient-free
    train()-mode forward applies the same counter-based masks as the fused training pass (tests/test_gpu_train.py compares the two on
    the GPU) instead of refusing; on a CPU-only box it must fail on the missing device like any other call -- no silent CPU path --
    and the mask seeds are one definition shared with the training engine."""
    import torch
    from efficient_tts_amd import EfficientTTSCNN
    m = EfficientTTSCNN(num_symbols=76, use_masking=True)                # dropout_rate = 0.1, train() mode
    t = torch.zeros(1, 8, dtype=torch.long)
    if not torch.cuda.is_available():
        for mode in (m.train, m.eval):
            mode()
            with torch.no_grad(), pytest.raises(Exception) as ei:
                m(t, torch.tensor([8]), torch.zeros(1, 16, 80), torch.tensor([16]))
            assert not isinstance(ei.value, NotImplementedError)
    conv, s0, s1 = m._dropout_seeds(7)
    conv2, t0, t1 = m._dropout_seeds(8)
    assert conv(30) != conv(31) and conv(30) != conv2(30) and (s0, s1) != (t0, t1) and s1 == (s0 + 1) & 0xFFFFFFFF
    assert m._drop(30) == (0.0, 0)                                       # no pass in progress: no masks


def test_lazy_stats_behaves_like_a_dict():
    import torch
    from efficient_tts_amd.model import LazyStats
    s = LazyStats(torch.tensor([3.0, 2.0, 1.0]))
    assert s["loss"] == 3.0 and s.get("mel_loss") == 2.0 and s.get("nope", 7) == 7
    assert dict(s) == dict(loss=3.0, mel_loss=2.0, duration_loss=1.0)
    assert list(s.keys()) == ["loss", "mel_loss", "duration_loss"] and sorted(s.values()) == [1.0, 2.0, 3.0]


def test_ctor_option_space_is_decided_on_the_host():
    """k_size and nonlinear_activation (efts_modules.py:19-46: any odd kernel, any getattr(torch.nn, name)(**params)): what the ctor accepts,
    which path it selects, and what it refuses by name -- no device needed.  The arithmetic of every accepted option is tested on the GPU
    (tests/test_gpu_variants.py)."""
    import pytest
    from efficient_tts_amd import EfficientTTSCNN, lib as L

    def make(**kw):
        return EfficientTTSCNN(num_symbols=76, n_text_encoder_layer=1, n_mel_encoder_layer=1, n_decoder_layer=1, **kw)
    for k, gap in ((1, 2), (3, 2), (5, 2), (7, 3), (9, 4), (11, 5)):
        m = make(k_size=k)
        assert m.row_gap == gap and m.text_encoder.layers[0].conv[0].kernel_size == (k,)
    for k in (0, 2, 4, 13):
        with pytest.raises(NotImplementedError):
            make(k_size=k)
    assert make().act_general is None and make().slope == pytest.approx(0.1)
    assert make(nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.3, "inplace": True}).slope == pytest.approx(0.3)
    assert make(nonlinear_activation="ReLU", nonlinear_activation_params={}).slope == 0.0
    with pytest.raises(TypeError):                       # the reference's torch.nn.ReLU(**params) raises on anything but `inplace`
        make(nonlinear_activation="ReLU", nonlinear_activation_params={"negative_slope": 0.1})
    assert make(nonlinear_activation="ReLU", nonlinear_activation_params={"inplace": True}).slope == 0.0
    assert make(nonlinear_activation="GELU", nonlinear_activation_params={}).act_general == (6, 0.0, 0.0)
    assert make(nonlinear_activation="GELU", nonlinear_activation_params={"approximate": "tanh"}).act_general == (7, 0.0, 0.0)
    assert make(nonlinear_activation="ELU", nonlinear_activation_params={"alpha": 0.5}).act_general == (3, 0.5, 0.0)
    assert make(nonlinear_activation="Softplus", nonlinear_activation_params={"beta": 2.0}).act_general == (12, 2.0, 20.0)
    assert make(nonlinear_activation="ReLU6", nonlinear_activation_params={}).act_general == (13, 0.0, 6.0)
    # every name the table lists is a torch.nn module with those keyword arguments, and the ids are the header's
    import re
    import torch
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "efts_abi.h")).read()
    ids = {int(v) for v in re.findall(r"#define EFTS_ACTFN_(?!COUNT)\w+ (\d+)", hdr)}
    assert {aid for aid, _ in L.ACTFN.values()} | {7} == ids and int(re.search(r"#define EFTS_ACTFN_COUNT (\d+)", hdr).group(1)) == max(ids) + 1
    for name, (aid, keys) in L.ACTFN.items():
        getattr(torch.nn, name)(**{k: d for k, d in keys})
    for name, params in (("PReLU", {}), ("RReLU", {}), ("Softmax", {"dim": 1}), ("GLU", {}), ("GELU", {"approximate": "x"}), ("ELU", {"beta": 1.0})):
        with pytest.raises(NotImplementedError):
            make(nonlinear_activation=name, nonlinear_activation_params=params)
