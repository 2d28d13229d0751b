"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the oracle and the
reference-generated golden fixtures.  Tolerances: north_star states mel max-abs <= 1e-3."""
import os

import numpy as np
import pytest
import torch

from oracle import efts_oracle as O

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-3          # north_star: "Output mels match the reference within 1e-3 max-abs"


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model():
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd import lib as L
    L.load()
    L.require_device()
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False,
                        sigma=0.01, precision="bf16x3")
    m.load_state_dict(O.fill_params())
    return m.to(_dev()).eval()


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


# ------------------------------------------------------------------ the MFMA contraction alone
@pytest.mark.parametrize("split", [2, 1])
@pytest.mark.parametrize("taps,cin,cout,B,T", [(5, 512, 512, 3, 70), (3, 512, 512, 2, 131), (1, 80, 512, 2, 50),
                                              (1, 512, 80, 1, 300), (5, 512, 512, 1, 1)])
def test_gemm_conv_vs_fp64(split, taps, cin, cout, B, T):
    from efficient_tts_amd import lib as L, ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(taps * 1000 + cin + T)
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cout, cin, taps, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, T, cout, generator=g)
    rs = P.Rows(B, T)
    a = P.Plane.for_rows(rs, cin, split, dev)
    P.pack_rows(x.to(dev), None, a, rs)
    pw = P.PackedWeight(cout, cin, taps, split, dev)
    pw.pack(w.to(dev).contiguous())
    resid = P.F32Rows(rs, cout, dev)
    resid.view().copy_(res.to(dev))
    gap = torch.zeros(rs.rows, device=dev)
    P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
    out = P.F32Rows(rs, cout, dev)
    outp = P.Plane.for_rows(rs, cout, 2, dev)
    P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rs.rows, n=cout, act=L.ACT_LEAKY,
           slope=0.1, bias=bias.to(dev), resid_ptr=resid.ptr, ldr=cout, rowmask_ptr=gap.data_ptr(),
           out_f32_ptr=out.ptr, ldo=cout, out_plane=outp)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), padding=(taps - 1) // 2)
    ref = (res.double() + torch.nn.functional.leaky_relu(ref, 0.1).transpose(1, 2)).float()
    got = out.view().cpu()
    scale = float(ref.abs().max())
    tol = (2e-5 if split == 2 else 2e-2) * scale
    assert float((got - ref).abs().max()) <= tol
    # gap rows and guard rows stay exactly zero
    full = out.buf.cpu()
    assert float(full[:L.GUARD_LO].abs().max()) == 0.0
    gaprows = full[L.GUARD_LO:L.GUARD_LO + rs.rows].view(B, rs.Tp, cout)[:, T:]
    assert float(gaprows.abs().max()) == 0.0
    # the hi/lo operand plane written by the epilogue reconstructs the fp32 output to 2^-16
    pl = outp.buf[L.GUARD_LO:L.GUARD_LO + rs.rows].view(torch.bfloat16).view(rs.rows, outp.nchunk, 2, 32).float().cpu()
    recon = (pl[:, :, 0] + pl[:, :, 1]).reshape(B, rs.Tp, outp.nchunk * 32)[:, :T, :cout]
    assert float((recon - got).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("split", [2, 1])
@pytest.mark.parametrize("taps,dil,c,T", [(3, 1, 32, 2600), (7, 3, 32, 2500), (11, 5, 32, 3000), (11, 1, 64, 2200), (7, 5, 24, 2100)])
def test_resident_kernel_vs_fp64_and_ring_kernels(split, taps, dil, c, T):
    """One-K-chunk convolutions with <= 64 columns over many rows (the vocoder's narrow stages) take resident32_kernel
    (window + every tap's weights resident in LDS).  Dilated taps, residual, row mask (a ragged second item), activated
    output plane: against float64 and, bit for bit, against the ring kernels (efts_gemm_args.tiling: narrow vs resident vs auto).
    Row space of the vocoder (64 zero guard rows in front: the halo of a dilated k = 11 tap is 25 rows)."""
    from efficient_tts_amd import lib as L, ops as P
    from efficient_tts_amd.vocoder import _GUARD, _Rows
    dev = _dev()
    g = torch.Generator().manual_seed(taps * 100 + dil * 10 + c)
    B, lens, gap = 2, [T, T - 777], 40
    Tp = T + gap
    rows = B * Tp
    x = torch.randn(B, T, c, generator=g)
    x[1, lens[1]:] = 0.0                                           # what a masked producer would have left there
    w = torch.randn(c, c, taps, generator=g) * 0.2
    bias = torch.randn(c, generator=g).to(dev)
    res = torch.randn(B, Tp, c, generator=g)
    lib = L.load()
    a = _Rows(rows, c, split, dev, f32=False)
    L.check(lib.efts_pack_rows(x.to(dev).contiguous().data_ptr(), None, a.p.ptr, a.p.ld, B, T, Tp, c, a.p.nchunk * P.chunk_k(split), split, None),
            "efts_pack_rows")
    pw = P.PackedWeight(c, c, taps, split, dev)
    pw.pack(w.to(dev).contiguous())
    resid = _Rows(rows, c, split, dev, plane=False)
    resid.f[_GUARD:_GUARD + rows].copy_(res.reshape(rows, c).to(dev))
    lenmask = torch.zeros(rows, device=dev)
    L.check(lib.efts_row_masks(torch.tensor(lens, dtype=torch.int32, device=dev).data_ptr(), None, lenmask.data_ptr(), B, T, Tp, None), "efts_row_masks")
    outs = {}
    resident_ok = a.p.nchunk == 1                     # the resident tiling needs the whole K in one 128-byte chunk
    for flag, tiling in (("1", L.TILING_NARROW), ("r", L.TILING_RESIDENT if resident_ok else L.TILING_NARROW), (None, L.TILING_AUTO)):
        out = _Rows(rows, c, split, dev)
        P.gemm(a=a.p, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rows, n=c, bias=bias, resid_ptr=resid.fptr, ldr=c,
               rowmask_ptr=lenmask.data_ptr(), out_f32_ptr=out.fptr, ldo=c, out_plane=out.p, dilation=dil, plane_act=True, plane_slope=0.1,
               tiling=tiling)
        torch.cuda.synchronize()
        outs[flag] = (out.f.clone(), out.p.buf.clone())
    assert torch.equal(outs[None][0], outs["1"][0]) and torch.equal(outs[None][1], outs["1"][1])
    assert torch.equal(outs["r"][0], outs["1"][0]) and torch.equal(outs["r"][1], outs["1"][1])
    if not resident_ok:
        with pytest.raises(ValueError):                # an explicit tiling the shape does not allow is an error, not a fallback
            P.gemm(a=a.p, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rows, n=c, out_f32_ptr=out.fptr, ldo=c,
                   dilation=dil, tiling=L.TILING_RESIDENT)
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.cpu().double(), padding=(taps - 1) // 2 * dil, dilation=dil)
    ref = (res[:, :T].double() + ref.transpose(1, 2)).float()
    ref[1, lens[1]:] = 0.0
    full = outs[None][0].cpu()
    got = full[_GUARD:_GUARD + rows].view(B, Tp, c)
    scale = float(ref.abs().max())
    assert float((got[:, :T] - ref).abs().max()) <= (2e-5 if split == 2 else 2e-2) * scale
    assert float(got[:, T:].abs().max()) == 0.0                    # rows between the items: masked
    assert float(full[:_GUARD].abs().max()) == 0.0 and float(full[_GUARD + rows:].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["fwd_tiny", "fwd_small", "fwd_full", "fwd_long"])
def test_forward_matches_reference_golden(golden_dir, model, case):
    g = _golden(golden_dir, case)
    dev = _dev()
    args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "speech", "speech_lengths")]
    with torch.no_grad():
        (loss, stats, imv, ralpha, mel_pred, _), extra = model._forward_impl(*[a.to(dev) for a in args], keep=True)
    torch.cuda.synchronize()
    st, sa = int(g["mel_pred_stride"]), int(g["alpha_stride"])
    T1 = args[0].shape[1]
    err = {
        "mel_pred": float((mel_pred.cpu()[:, ::st] - torch.from_numpy(g["mel_pred"])).abs().max()),
        "reconst_alpha": float((ralpha.cpu()[:, ::sa, ::sa] - torch.from_numpy(g["reconst_alpha"])).abs().max()),
        "imv": float((imv.cpu() - torch.from_numpy(g["imv"])).abs().max()),
        "e": float((extra["e"].cpu() - torch.from_numpy(g["e"])).abs().max()),
        "dur_pred": float((extra["dur_pred"].cpu() - torch.from_numpy(g["dur_pred"])).abs().max()),
        "log_delta_e": float((extra["log_delta_e"].cpu() - torch.from_numpy(g["log_delta_e"])).abs().max()),
        "loss": abs(float(loss) - float(g["loss"])),
    }
    print(case, err)
    assert err["mel_pred"] <= MEL_TOL, err
    assert err["reconst_alpha"] <= 1e-3, err
    assert err["imv"] <= 2e-3 and err["e"] <= 1e-2, err          # index units (0..T1 / 0..T2)
    assert err["dur_pred"] <= 1e-3 and err["log_delta_e"] <= 1e-3, err
    assert err["loss"] <= 1e-4 * float(g["loss"]), err
    assert abs(stats["mel_loss"] - float(g["mel_loss"])) <= 1e-4 * float(g["mel_loss"])
    assert abs(stats["duration_loss"] - float(g["dur_loss"])) <= 2e-4 * max(1.0, float(g["dur_loss"]))
    # and against the oracle on the same inputs (full tensors, not strided samples)
    o = O.forward(O.fill_params(), *args)
    assert float((mel_pred.cpu() - o["mel_pred"]).abs().max()) <= MEL_TOL
    assert float((ralpha.cpu() - o["reconst_alpha"]).abs().max()) <= 1e-3


def test_inference_matches_reference_golden(golden_dir, model):
    g = _golden(golden_dir, "inference_lj")
    dev = _dev()
    for n in range(4):
        ids = torch.from_numpy(g[f"text{n}"]).to(dev)
        mel, ralpha = model.inference(ids)
        assert mel.shape[1] == int(g[f"t2_{n}"]) and ralpha.shape == (1, ids.shape[1], mel.shape[1])
        assert float((mel.cpu()[:, ::2] - torch.from_numpy(g[f"mel_pred{n}"])).abs().max()) <= MEL_TOL
        assert float((ralpha.cpu()[:, ::4, ::4] - torch.from_numpy(g[f"reconst_alpha{n}"])).abs().max()) <= 1e-4


def test_remove_weight_norm_keeps_outputs(model, golden_dir):
    from efficient_tts_amd import EfficientTTSCNN
    g = _golden(golden_dir, "inference_lj")
    ids = torch.from_numpy(g["text0"]).to(_dev())
    a, _ = model.inference(ids)
    m2 = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01)
    m2.load_state_dict(O.fill_params())
    m2 = m2.to(_dev()).eval()
    m2.remove_weight_norm()
    assert "decoder.layers.0.conv.0.weight" in m2.state_dict()
    b, _ = m2.inference(ids)
    # torch's fold (remove_weight_norm) and the fused fold kernel round g*v/||v|| differently by 1 ulp
    assert float((a - b).abs().max()) <= 2e-4


def test_bf16_mode_reports_its_own_error(golden_dir):
    """precision="bf16" (single bf16 MFMA per product) is the fast mode; it cannot meet 1e-3
    (the reference itself moves by 0.49 under bf16 autocast) -- bound its error."""
    from efficient_tts_amd import EfficientTTSCNN
    g = _golden(golden_dir, "fwd_full")
    dev = _dev()
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16")
    m.load_state_dict(O.fill_params())
    m = m.to(dev).eval()
    args = [torch.from_numpy(g[k]).to(dev) for k in ("text", "text_lengths", "speech", "speech_lengths")]
    with torch.no_grad():
        loss, stats, imv, ralpha, mel_pred, _ = m(*args)
    err = float((mel_pred.cpu()[:, ::int(g["mel_pred_stride"])] - torch.from_numpy(g["mel_pred"])).abs().max())
    print("bf16 mel max-abs", err)
    assert err <= 0.15 and abs(float(loss) - float(g["loss"])) <= 2e-2 * float(g["loss"])


def test_full_size_properties(model):
    """BASELINE config 2 shape (B=64, 128, 800): size-independent properties --
    batch-permutation equivariance, item independence, alpha' columns sum to 1, imv monotone."""
    dev = _dev()
    gen = torch.Generator().manual_seed(1234)
    B, T1, T2 = 64, 128, 800
    text = torch.randint(0, 76, (B, T1), generator=gen).to(dev)
    mel = torch.randn(B, T2, 80, generator=gen).to(dev)
    tl = torch.full((B,), T1, dtype=torch.int64, device=dev)
    sl = torch.full((B,), T2, dtype=torch.int64, device=dev)
    with torch.no_grad():
        loss, stats, imv, ralpha, mel_pred, _ = model(text, tl, mel, sl)
        perm = torch.randperm(B, generator=gen).to(dev)
        loss2, _, imv2, ralpha2, mel2, _ = model(text[perm], tl, mel[perm], sl)
        sub = slice(5, 9)
        _, _, imv3, _, mel3, _ = model(text[sub], tl[sub], mel[sub], sl[sub])
    assert torch.isfinite(mel_pred).all()
    assert float((mel_pred[perm] - mel2).abs().max()) == 0.0            # bitwise: tiles never mix items and
    assert float((imv[perm] - imv2).abs().max()) == 0.0                 # every row sees the same summation order
    assert abs(float(loss) - float(loss2)) <= 1e-5 * float(loss)
    # item independence across kernels: the 64-item row space runs its stacks on efts_resconv5 with a hi/lo bf16 stream,
    # the 4-item one on efts_gemm with an fp32 stream (2^-17 relative per layer; both sit within 1e-3 of the oracle)
    assert float((mel_pred[sub] - mel3).abs().max()) <= 5e-4
    assert float((ralpha.sum(1) - 1).abs().max()) <= 1e-4
    assert float((imv[:, 1:] - imv[:, :-1]).min()) >= 0.0
    assert float((imv[:, -1] - (T1 - 1)).abs().max()) <= 1e-3


def test_batched_ragged_inference_equals_single_item(golden_dir, model):
    """inference_batch (extension): every item of a ragged batch equals the B=1
    reference-semantics inference of that item alone (and therefore the reference golden)."""
    g = _golden(golden_dir, "inference_lj")
    dev = _dev()
    seqs = [torch.from_numpy(g[f"text{n}"])[0] for n in range(4)]
    T1 = max(len(s) for s in seqs)
    text = torch.zeros(4, T1, dtype=torch.int64)
    for n, s in enumerate(seqs):
        text[n, :len(s)] = s
    lens = torch.tensor([len(s) for s in seqs])
    mel, mel_len, ralpha = model.inference_batch(text.to(dev), lens.to(dev))
    for n, s in enumerate(seqs):
        t2 = int(g[f"t2_{n}"])
        assert int(mel_len[n]) == t2
        one, ra1 = model.inference(s[None].to(dev))
        assert float((mel[n, :t2] - one[0]).abs().max()) <= 2e-4
        assert float(mel[n, t2:].abs().max()) == 0.0 if mel.shape[1] > t2 else True
        assert float((ralpha[n, :len(s), :t2] - ra1[0]).abs().max()) <= 1e-5
        assert float((mel[n, :t2:2].cpu() - torch.from_numpy(g[f"mel_pred{n}"])[0]).abs().max()) <= MEL_TOL


@pytest.mark.parametrize("B,T1,T2,tl,sl", [(1, 9, 33, [9], [33]), (3, 30, 77, [30, 17, 1], [77, 40, 5]),
                                           (2, 130, 300, [130, 64], [300, 299])])
def test_odd_shapes_and_ragged_lengths_vs_oracle(model, B, T1, T2, tl, sl):
    """shapes that are not multiples of any tile (scalar epilogue path, partial K chunks), B = 1, a
    length-1 text and very short mels, T1 > 128 (two key tiles): forward vs the oracle."""
    dev = _dev()
    gen = torch.Generator().manual_seed(B * 1000 + T1 + T2)
    text = torch.randint(0, 76, (B, T1), generator=gen)
    mel = (-4.0 + 2.0 * torch.randn(B, T2, 80, generator=gen)).clamp(-11.5, 2.0)
    tlt, slt = torch.tensor(tl), torch.tensor(sl)
    for b in range(B):
        text[b, tl[b]:] = 0
        mel[b, sl[b]:] = 0
    with torch.no_grad():
        loss, stats, imv, ralpha, mel_pred, _ = model(text.to(dev), tlt.to(dev), mel.to(dev), slt.to(dev))
        o = O.forward(O.fill_params(), text, tlt, mel, slt)
    assert float((mel_pred.cpu() - o["mel_pred"]).abs().max()) <= MEL_TOL
    assert float((ralpha.cpu() - o["reconst_alpha"]).abs().max()) <= 1e-3
    assert float((imv.cpu() - o["imv"]).abs().max()) <= 2e-3
    assert abs(float(loss) - float(o["loss"])) <= 2e-4 * float(o["loss"])


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("B,T1,T2", [(3, 40, 300), (5, 128, 517)])
def test_mel_loss_in_the_head_epilogue_equals_the_loss_launch(precision, B, T1, T2):
    """efts_gemm_args.sqerr_part: the mel head's launch leaves sum (mel_pred - speech)^2 over the valid frames per workgroup and wave and
    efts_losses_from_parts adds them up (fastspeech_loss.py:54-67 behind efficient_tts.py:198-200, :220).  Against the separate loss
    launches on ragged lengths: the outputs of the head bit for bit, the three loss values to fp32 summation order, run-to-run identical."""
    from efficient_tts_amd import EfficientTTSCNN
    g = torch.Generator().manual_seed(B * 1000 + T2)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=precision)
    m.load_state_dict(O.fill_params())
    m = m.cuda().eval()
    m.graphs = False
    text = torch.randint(0, 76, (B, T1), generator=g).cuda()
    speech = torch.randn(B, T2, 80, generator=g).cuda()
    tl = torch.randint(T1 // 2, T1 + 1, (B,), generator=g)
    ml = torch.randint(T2 // 2, T2 + 1, (B,), generator=g)
    tl[0], ml[0] = T1, T2
    speech[1, int(ml[1]):] = float("nan")              # frames past an item's length are never read into the loss
    outs = {}
    with torch.no_grad():
        for fused in (False, True, True):
            m.fuse_mel_loss, m.fuse_mel_loss_min_wgs = fused, 1
            o = m(text, tl.cuda(), speech, ml.cuda())
            torch.cuda.synchronize()
            assert (m._sqerr_parts is not None) == fused
            outs.setdefault(fused, []).append((o[1]._t.clone().cpu(), o[4].clone()))       # (loss, mel_loss, duration_loss) as the device wrote them
    (la, mela), (lb, melb), (lc, melc) = outs[False][0], outs[True][0], outs[True][1]
    assert torch.equal(mela, melb) and torch.equal(melb, melc)
    assert torch.equal(lb, lc)
    assert torch.isfinite(la).all() and ((la - lb).abs() <= 2e-6 * la.abs()).all(), (la, lb)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_narrow_tiles_equal_gemm_kernel(golden_dir, precision):
    """Launches that leave most CUs idle (the text side of a 2-item batch: 260 rows) take 64-column tiles in the automatic
    tiling; forcing the generic 124 x 128 kernel for EVERY launch of the forward must give the same outputs to the last bits
    (same per-element summation order)."""
    from efficient_tts_amd import EfficientTTSCNN, lib as L, ops as P
    g = np.load(os.path.join(golden_dir, "fwd_full.npz"))
    args = [torch.from_numpy(g[k]).cuda() for k in ("text", "text_lengths", "speech", "speech_lengths")]
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=precision)
    m.load_state_dict(O.fill_params())
    m = m.cuda().eval()
    m.graphs = False                                   # the tiling is baked into a captured graph
    outs = {}
    try:
        for tiling in (L.TILING_GENERIC, L.TILING_AUTO):
            P.GEMM_TILING = tiling
            with torch.no_grad():
                o = m(*args)
            torch.cuda.synchronize()
            outs[tiling] = (float(o[0]), o[4].clone(), o[3].clone())
    finally:
        P.GEMM_TILING = L.TILING_AUTO
    a, b = outs[L.TILING_GENERIC], outs[L.TILING_AUTO]
    assert (a[1] - b[1]).abs().max().item() <= 1e-6
    assert (a[2] - b[2]).abs().max().item() <= 1e-6
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0])
    if precision == "bf16x3":
        stride = int(g["mel_pred_stride"])
        assert np.abs(b[1].cpu().numpy()[:, ::stride] - g["mel_pred"]).max() <= 1e-3


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("B,T", [(5, 300), (2, 750)])      # row counts whose last 256-row window stays inside the guard rows
def test_conv5_kernel_equals_gemm_kernel(split, B, T):
    """The 256-row k5 kernel (conv5_kernel: the training step's large forward / dgrad launches) against the 124-row
    gemm_kernel on the same operands: identical summation order per output element, so fp32 stream and operand plane must
    agree to the last bits (ragged length mask, residual, LeakyReLU, both operand formats)."""
    from efficient_tts_amd import lib as L, ops as P
    L.load(); L.require_device()
    dev = _dev()
    C = 512
    torch.manual_seed(B * 1000 + T + split)
    rs = P.Rows(B, T)
    x = torch.randn(B, T, C, device=dev)
    a = P.Plane.for_rows(rs, C, split, dev)
    P.pack_rows(x, None, a, rs)
    xf = P.F32Rows(rs, C, dev); xf.view().copy_(x)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev)
    mask = torch.zeros(rs.rows, device=dev)
    P.row_masks(torch.randint(T // 2, T + 1, (B,), dtype=torch.int32, device=dev), rs, None, mask)
    outs = {}
    for tiling in (L.TILING_GENERIC, L.TILING_WIDE):
        o, pl = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, 2, dev)
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias,
               resid_ptr=xf.ptr, ldr=C, rowmask_ptr=mask.data_ptr(), out_f32_ptr=o.ptr, ldo=C, out_plane=pl, tiling=tiling)
        torch.cuda.synchronize()
        outs[tiling] = (o.buf, pl.buf)
    assert torch.equal(outs[L.TILING_GENERIC][0], outs[L.TILING_WIDE][0])
    assert torch.equal(outs[L.TILING_GENERIC][1], outs[L.TILING_WIDE][1])


def test_inference_graph_path_equals_exact_path(golden_dir, model):
    """inference() with the graph cache on runs as a ragged batch of one on bucketed shapes (T1 up to a multiple of 16, T2 of 64)
    and replays two hipGraphs around the single host sync; it must give the T2 of the exact unmasked B = 1 pass and the same
    mel to ~1e-4, call after call (eager first call, capture, replays), for utterances that fall into different buckets."""
    g = _golden(golden_dir, "inference_lj")
    dev = _dev()
    for n in range(4):
        ids = torch.from_numpy(g[f"text{n}"]).to(dev)
        model.graphs = False
        try:
            exact, ra_exact = model.inference(ids)
        finally:
            model.graphs = True
        outs = [model.inference(ids) for _ in range(3)]          # eager / captured + replayed / replayed
        for mel, ra in outs:
            assert mel.shape == exact.shape and ra.shape == ra_exact.shape and mel.shape[1] == int(g[f"t2_{n}"])
            assert float((mel - exact).abs().max()) <= 2e-4
            assert float((ra - ra_exact).abs().max()) <= 1e-5
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][0], outs[2][0])
        assert float((outs[2][0][0, ::2].cpu() - torch.from_numpy(g[f"mel_pred{n}"])[0]).abs().max()) <= MEL_TOL


@pytest.mark.parametrize("share,want", [(0.0, "graph"), (1e9, "eager")])
def test_forward_call_policy_graph_or_eager_gives_the_same_fresh_tensors(share, want):
    """The teacher-forced forward decides per shape, from its second call, whether it replays a hipGraph or stays on eager launches
    (graphs.GraphCache.run, adaptive).  Forced either way, call after call must return the same numbers as a graph-free model, in
    tensors the later calls do not overwrite; graph_policy="always" captures on the second call as before."""
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.graphs import GraphCache
    dev = _dev()
    torch.manual_seed(3)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01).to(dev).eval()
    B, T1, T2 = 3, 40, 130
    gen = torch.Generator().manual_seed(5)
    text = torch.randint(0, 76, (B, T1), generator=gen).to(dev)
    mel = torch.randn(B, T2, 80, generator=gen).to(dev)
    tl, sl = torch.tensor([40, 33, 9]).to(dev), torch.tensor([130, 77, 20]).to(dev)
    m.graphs = False
    with torch.no_grad():
        ref = m(text, tl, mel, sl)
    m.graphs = True
    m._graph_cache = GraphCache()
    m._graph_cache.EAGER_MAX_HOST_SHARE = share
    with torch.no_grad():
        outs = [m(text, tl, mel, sl) for _ in range(5)]
    ent = next(iter(m._graph_cache.entries.values()))
    assert ent.policy[0] == want and (ent.graph is not None) == (want == "graph")
    for o in outs:
        assert float(o[0]) == float(ref[0])
        for a, b in zip(o[2:5], ref[2:5]):
            assert torch.equal(a, b)
    assert outs[3][4].data_ptr() != outs[4][4].data_ptr()
    m.graph_policy = "always"
    m._graph_cache = GraphCache()
    with torch.no_grad():
        outs = [m(text, tl, mel, sl) for _ in range(2)]
    ent = next(iter(m._graph_cache.entries.values()))
    assert ent.graph is not None and ent.policy is None and torch.equal(outs[1][4], ref[4])


def test_inference_graphs_are_shared_by_the_lengths_of_a_bucket(golden_dir, model):
    """utterances of different lengths inside one T1 bucket replay the SAME two captured graphs (the ids are written into the
    bucket-wide static input, phase 2 reads phase 1's outputs in place): alternating lengths must neither re-capture per call nor
    leak one call's ids / positions into the next, and the tensors handed back must not be overwritten by later calls."""
    g = _golden(golden_dir, "inference_lj")
    dev = _dev()
    base = torch.from_numpy(g["text0"]).to(dev)
    T1 = base.shape[1]
    lo = (T1 - 1) // 16 * 16 + 1                                   # the shortest length of the bucket T1 is in
    ids = [base[:, :n] for n in sorted({max(lo, T1 - 7), T1})]
    assert len(ids) == 2
    model.graphs = False
    try:
        exact = [model.inference(x)[0] for x in ids]
    finally:
        model.graphs = True
    for x in ids * 2:                                                # eager, then captured
        model.inference(x)
    held = {k: (e.graph, e.calls) for k, e in model._infer_cache.entries.items()}
    outs = [model.inference(x)[0] for x in ids * 3]
    for k, (graph, calls) in held.items():
        assert model._infer_cache.entries[k].graph is graph, f"{k} was captured again"
    for i, mel in enumerate(outs):
        assert mel.shape == exact[i % 2].shape
        assert float((mel - exact[i % 2]).abs().max()) <= 2e-4
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[5])      # (still intact after the later calls)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 37, 211), (2, 128, 800), (4, 100, 124)])
def test_soft_index_in_the_gemm_epilogue_equals_the_two_kernel_path(shape):
    """efts_gemm(soft_index=...) -- q.k^T, softmax over the valid keys and its expected key index in one launch
    (efficient_tts.py:390-398, :312) -- against efts_gemm + efts_attn_soft_index on the stored scores, ragged key / query
    lengths; the reductions associate differently (32 lanes x 4 columns against 64 lanes, stride 64), so equal to rounding; and the
    model with the fused launch against the model without it."""
    from efficient_tts_amd import lib as L, ops as P
    B, T1, T2 = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    C = 512
    rs1, rs2 = P.Rows(B, T1), P.Rows(B, T2)
    q = P.Plane.for_rows(rs2, C, 2, dev); P.pack_rows(torch.randn(B, T2, C, device=dev), None, q, rs2)
    k = P.Plane.for_rows(rs1, C, 2, dev); P.pack_rows(torch.randn(B, T1, C, device=dev), None, k, rs1)
    tl = torch.randint(max(1, T1 // 2), T1 + 1, (B,), dtype=torch.int32, device=dev); tl[0] = T1
    ml = torch.randint(max(1, T2 // 2), T2 + 1, (B,), dtype=torch.int32, device=dev); ml[0] = T2
    scores = torch.empty(B, T2, T1, device=dev)
    s_ref, s_fused = torch.empty(B, T2, device=dev), torch.full((B, T2), -1.0, device=dev)
    common = dict(a=q, b_ptr=k.ptr, ldb=k.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q.ld, b_batch_stride=rs1.Tp * k.ld, alpha=P.INV_SQRT(C))
    with P.stream_scope():
        P.gemm(out_f32_ptr=scores.data_ptr(), ldo=T1, out_batch_stride=T2 * T1, **common)
        P.attn_soft_index(scores, T1, tl, ml, s_ref, None, B, T1, T2)
        P.gemm(soft_index=s_fused, key_len=tl, query_len=ml, **common)
        torch.cuda.synchronize()
    assert torch.isfinite(s_fused).all()
    assert (s_fused - s_ref).abs().max().item() <= 2e-4 * T1, (s_fused - s_ref).abs().max().item()
    for b in range(B):
        assert (s_fused[b, int(ml[b]):] == 0).all()
    # fp64 restatement on the stored scores
    want = torch.zeros(B, T2, dtype=torch.float64, device=dev)
    for b in range(B):
        p = torch.softmax(scores[b, :, :int(tl[b])].double(), dim=-1)
        want[b] = (p * torch.arange(int(tl[b]), device=dev, dtype=torch.float64)).sum(-1)
        want[b, int(ml[b]):] = 0
    assert (s_fused.double() - want).abs().max().item() <= 1e-4 * T1
    with pytest.raises(Exception):                                  # more than one column tile: not in the epilogue
        P.gemm(a=q, b_ptr=q.ptr, ldb=q.ld, m=T2, n=T2 if T2 > 128 else 129, batch=B, a_batch_stride=rs2.Tp * q.ld, b_batch_stride=rs2.Tp * q.ld,
               soft_index=s_fused, key_len=tl, query_len=ml)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("shape", [(3, 37, 80), (2, 800, 80), (5, 129, 80), (2, 100, 128), (1, 64, 24)])
def test_frame_linear_equals_pack_rows_plus_gemm(split, shape):
    """efts_frame_linear (the prenet straight from the caller's fp32 frames) against efts_pack_rows + efts_gemm on the same weights:
    bit for bit (same operand rounding, same k order), fp32 output, operand plane and remainder plane; gap rows stay zero."""
    from efficient_tts_amd import lib as L, ops as P
    B, T, cin = shape
    C = 512
    dev = torch.device("cuda:0")
    torch.manual_seed(T + cin)
    rs = P.Rows(B, T)
    x = torch.randn(B, T, cin, device=dev)
    pw = P.PackedWeight(C, cin, 1, split, dev); pw.pack((torch.randn(C, cin, device=dev) * 0.1).contiguous())
    bias = torch.randn(C, device=dev)
    gap = torch.zeros(rs.rows, device=dev); P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
    with P.stream_scope():
        a = P.Plane.for_rows(rs, cin, split, dev); P.pack_rows(x, None, a, rs)
        o_ref = P.F32Rows(rs, C, dev); p_ref = P.Plane.for_rows(rs, C, split, dev)
        l_ref = P.Plane.for_rows(rs, C, 1, dev) if split == 1 else None
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias, rowmask_ptr=gap.data_ptr(),
               out_f32_ptr=o_ref.ptr, ldo=C, out_plane=p_ref)
        if split == 1:
            P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1, bias=bias, rowmask_ptr=gap.data_ptr(),
                   out_plane=P.Plane.for_rows(rs, C, 1, dev), out_plane_lo=l_ref)
        o = P.F32Rows(rs, C, dev); y = P.Plane.for_rows(rs, C, split, dev)
        yl = P.Plane.for_rows(rs, C, 1, dev) if split == 1 else None
        P.frame_linear(x=x, w=pw, bias=bias, act=L.ACT_LEAKY, slope=0.1, rs=rs, y=y, y_lo=yl, y_f32=o)
        torch.cuda.synchronize()
    assert torch.equal(o.buf, o_ref.buf), float((o.buf - o_ref.buf).abs().max())
    bf = lambda pl: pl.buf.view(torch.bfloat16)            # compared as numbers: the masked gap rows of efts_gemm are -0 where the value was negative
    assert torch.equal(bf(y), bf(p_ref))
    live = gap.bool()
    assert torch.equal(y.buf[L.GUARD_LO:L.GUARD_LO + rs.rows][live], p_ref.buf[L.GUARD_LO:L.GUARD_LO + rs.rows][live])       # live rows: byte for byte
    assert (y.buf[L.GUARD_LO:L.GUARD_LO + rs.rows][~live] == 0).all()
    if split == 1:
        assert torch.equal(bf(yl), bf(l_ref))
    with pytest.raises(Exception):
        P.frame_linear(x=torch.randn(B, T, 130, device=dev), w=pw, bias=bias, act=L.ACT_LEAKY, slope=0.1, rs=rs, y=y)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [2, 1])
@pytest.mark.parametrize("taps,cin,cout,B,T", [(5, 512, 512, 1, 86), (5, 512, 512, 1, 601), (3, 512, 512, 3, 70), (1, 512, 512, 2, 131), (5, 512, 512, 1, 1),
                                              (1, 320, 96, 1, 64), (5, 64, 32, 2, 33)])
def test_small_m_tiling_vs_fp64_and_the_ring_kernels(split, taps, cin, cout, B, T):
    """EFTS_TILING_SMALLM (one-utterance row spaces: 64 x 32 tiles, K split across the four waves, fragments straight from global
    memory) against fp64 and against the generic ring kernel on the same operands: same operand rounding, so the two agree to fp32
    summation-order noise; bias, LeakyReLU / ReLU, residual, row mask, fp32 output and both operand-plane formats; gap rows zero."""
    from efficient_tts_amd import lib as L, ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(taps * 1000 + cin + T + 7)
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cout, cin, taps, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, T, cout, generator=g)
    rs = P.Rows(B, T)
    a = P.Plane.for_rows(rs, cin, split, dev)
    P.pack_rows(x.to(dev), None, a, rs)
    pw = P.PackedWeight(cout, cin, taps, split, dev)
    pw.pack(w.to(dev).contiguous())
    resid = P.F32Rows(rs, cout, dev)
    resid.view().copy_(res.to(dev))
    gap = torch.zeros(rs.rows, device=dev)
    P.row_masks(torch.full((B,), T, dtype=torch.int32, device=dev), rs, gap, None)
    act = L.ACT_RELU if taps == 3 else L.ACT_LEAKY
    outs = {}
    for til in (L.TILING_SMALLM, L.TILING_GENERIC):
        out = P.F32Rows(rs, cout, dev)
        outp = P.Plane.for_rows(rs, cout, 3 - split, dev)          # the other plane format than the operands'
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rs.rows, n=cout, act=act, slope=0.1, alpha=0.5,
               bias=bias.to(dev), resid_ptr=resid.ptr, ldr=cout, rowmask_ptr=gap.data_ptr(), out_f32_ptr=out.ptr, ldo=cout, out_plane=outp,
               tiling=til)
        outs[til] = (out, outp)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), None, padding=(taps - 1) // 2) * 0.5 + bias.double()[None, :, None]
    ref = torch.relu(ref) if taps == 3 else torch.nn.functional.leaky_relu(ref, 0.1)
    ref = (res.double() + ref.transpose(1, 2)).float()
    out, outp = outs[L.TILING_SMALLM]
    got = out.view().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= (2e-5 if split == 2 else 2e-2) * scale
    assert float((out.buf - outs[L.TILING_GENERIC][0].buf).abs().max()) <= 3e-6 * scale          # same operands, other summation order
    full = out.buf.cpu()
    assert float(full[:L.GUARD_LO].abs().max()) == 0.0
    assert float(full[L.GUARD_LO:L.GUARD_LO + rs.rows].view(B, rs.Tp, cout)[:, T:].abs().max()) == 0.0
    assert float(full[L.GUARD_LO + rs.rows:].abs().max()) == 0.0
    # the operand plane holds the fp32 output to its format's precision
    pl = outp.buf[L.GUARD_LO:L.GUARD_LO + rs.rows].view(torch.bfloat16)
    if outp.split == 2:
        pl = pl.view(rs.rows, outp.nchunk, 2, 32).float().cpu()
        recon = (pl[:, :, 0] + pl[:, :, 1]).reshape(B, rs.Tp, outp.nchunk * 32)[:, :T, :cout]
        assert float((recon - got).abs().max()) <= 2e-5 * scale
    else:
        recon = pl.float().cpu().view(B, rs.Tp, -1)[:, :T, :cout]
        assert float((recon - got).abs().max()) <= 2.0 ** -8 * scale
    with pytest.raises(ValueError):
        P.gemm(a=a, b_ptr=pw.ptr, ldb=pw.ld, b_tap_stride=pw.tap_stride, taps=taps, m=rs.rows, n=cout - 8, out_f32_ptr=out.ptr, ldo=cout,
               tiling=L.TILING_SMALLM)


def test_graphs_follow_weight_updates():
    """a captured forward / inference graph bakes the address of text-encoder layer 0's tap table
    (efts_embed_conv) into its launches.  After a weight change on a WARM model -- an optimizer step, load_state_dict -- the next
    plain call must give what a graph-free model gives with the new weights (the table is rebuilt in place, and its address is
    part of the graph tags), for the teacher-forced forward and for inference()."""
    from efficient_tts_amd import EfficientTTSCNN
    dev = _dev()
    torch.manual_seed(11)
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01).to(dev).eval()
    assert m.embed_conv
    m.graph_policy = "always"
    with torch.no_grad():
        m.duration_predictor.linear.bias.fill_(1.5)
    B, T1, T2 = 2, 40, 130
    gen = torch.Generator().manual_seed(5)
    text = torch.randint(0, 76, (B, T1), generator=gen).to(dev)
    mel = torch.randn(B, T2, 80, generator=gen).to(dev)
    tl, sl = torch.tensor([40, 33]).to(dev), torch.tensor([130, 77]).to(dev)
    one = text[:1, :37].contiguous()
    with torch.no_grad():
        for _ in range(3):                                     # eager, capture, replay
            m(text, tl, mel, sl)
            m.inference(one)
        assert next(iter(m._graph_cache.entries.values())).graph is not None
        tab0 = m._te0_ptr()
        for rnd in range(2):                                   # two rounds of updates: the table must be rebuilt, not re-allocated
            for p in (m.text_embedding_table.weight, m.text_encoder.layers[0].conv[0].weight_v, m.decoder.layers[2].conv[0].bias):
                p.add_(0.05 * torch.randn_like(p))
            got = m(text, tl, mel, sl)
            got_inf = m.inference(one)
            assert m._te0_ptr() == tab0
            m.graphs = False
            ref = m(text, tl, mel, sl)
            ref_inf = m.inference(one)
            m.graphs = True
            assert float(got[0]) == float(ref[0]), rnd
            assert torch.equal(got[4], ref[4]) and torch.equal(got[2], ref[2]), rnd
            assert got_inf[0].shape == ref_inf[0].shape and float((got_inf[0] - ref_inf[0]).abs().max()) <= 2e-4, rnd


def test_inference_results_survive_later_calls_of_the_same_bucket(golden_dir, model):
    """at B = 1 the trimmed slice of phase 2's static output is already contiguous, so `.contiguous()` handed
    the caller a VIEW that the next utterance of the same (T1, T2) bucket overwrote.  Two DIFFERENT utterances that land in one
    bucket: the tensors of the first must still hold its own result after the second has run."""
    g = _golden(golden_dir, "inference_lj")
    dev = _dev()
    base = torch.from_numpy(g["text0"]).to(dev)
    T1 = base.shape[1]
    model.graphs = False
    try:
        exact0 = model.inference(base)
        t2b = -(-exact0[0].shape[1] // model.T2_BUCKET)
        other, exact1 = None, None
        for k in range(1, T1 - 1):                               # a second utterance of the same length whose mel lands in the same bucket
            cand = base.clone()
            cand[0, k], cand[0, k + 1] = base[0, k + 1], base[0, k]
            if torch.equal(cand, base):
                continue
            e1 = model.inference(cand)
            if -(-e1[0].shape[1] // model.T2_BUCKET) == t2b and (e1[0].shape != exact0[0].shape or not torch.equal(e1[0], exact0[0])):
                other, exact1 = cand, e1
                break
    finally:
        model.graphs = True
    assert other is not None, "no second utterance of the same bucket found"
    for x in (base, other) * 2:                                  # eager, then captured
        model.inference(x)
    a = model.inference(base)
    a_mel, a_al = a[0].clone(), a[1].clone()
    b = model.inference(other)
    c = model.inference(base)
    for t in (a[0], a[1]):
        assert all(t.data_ptr() != u.data_ptr() for u in (b[0], b[1], c[0], c[1]))
    assert torch.equal(a[0], a_mel) and torch.equal(a[1], a_al), "the first call's tensors were overwritten by a later call"
    assert float((a[0] - exact0[0]).abs().max()) <= 2e-4 and float((b[0] - exact1[0]).abs().max()) <= 2e-4
    assert torch.equal(a[0], c[0])
