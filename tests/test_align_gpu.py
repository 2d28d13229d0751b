"""GPU (-m gpu): the fused alignment kernels (csrc/efts_align.hip) against the kernel chain they replace and against fp64.

efts_imv_align == efts_imv_scan + efts_aligned_positions + efts_duration_target, bit for bit.
efts_expand    == efts_reconst_alpha + efts_pack_vt + efts_gemm (split-bf16) to fp32 rounding, and within 2e-5 (relative to
                  the output range) of the fp64 value of reconstruct_align_from_aligned_position + bmm
                  (nntts/models/efficient_tts.py:347-375, :186, :190-194).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _lengths(B, T, g, full_first=True):
    ln = torch.randint(max(1, T // 3), T + 1, (B,), generator=g)
    if full_first:
        ln[0] = T
    return ln.to(torch.int32)


@pytest.mark.parametrize("method1", [True, False])
@pytest.mark.parametrize("B,T1,T2", [(3, 37, 211), (2, 128, 800), (4, 100, 124), (1, 9, 33), (2, 200, 1500), (2, 128, 4100)])
def test_imv_align_equals_the_three_kernel_chain(B, T1, T2, method1):
    from efficient_tts_amd import ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(T1 * 7 + T2)
    tl, ml = _lengths(B, T1, g).to(dev), _lengths(B, T2, g).to(dev)
    # a soft index like the attention's: noisy, roughly increasing over the frames
    sidx = (torch.rand(B, T2, generator=g).cumsum(1) / T2 * T1 * 2 + torch.randn(B, T2, generator=g) * 0.7).to(dev)
    imv0, e0, lde0 = torch.empty(B, T2, device=dev), torch.empty(B, T1, device=dev), torch.empty(B, T1, device=dev)
    P.imv_scan(sidx, tl, ml, imv0, B, T2)
    P.aligned_positions(imv0, tl, ml, 0.5, 1.0, e0, lde0 if method1 else None, B, T1, T2)
    if not method1:
        P.duration_target(e0, tl, ml, 1.0, False, lde0, B, T1)
    imv1, e1, lde1 = torch.full((B, T2), 7.0, device=dev), torch.full((B, T1), 7.0, device=dev), torch.full((B, T1), 7.0, device=dev)
    P.imv_align(sidx, tl, ml, 0.5, 1.0, method1, imv1, e1, lde1, B, T1, T2)
    torch.cuda.synchronize()
    assert torch.equal(imv0, imv1)
    assert torch.equal(e0, e1)
    assert torch.equal(lde0, lde1)
    assert bool((imv1[:, 1:] >= imv1[:, :-1]).logical_or(imv1[:, 1:] == 0).all())      # monotone up to the masked tail


def _expand_ref64(e, v, tl, ml, sigma, T2):
    """fp64 restatement of efficient_tts.py:347-375 + :186 + :190-194 on [B, T1] positions and [B, T1, C] values"""
    B, T1 = e.shape
    e, v = e.double(), v.double()
    out_a = torch.zeros(B, T1, T2, dtype=torch.float64)
    for b in range(B):
        t, m = (int(tl[b]), int(ml[b])) if tl is not None else (T1, T2)
        q = torch.arange(T2, dtype=torch.float64)
        q[m:] = 0.0
        en = -sigma * (q[None, :] - e[b, :t, None]) ** 2
        a = torch.softmax(en, dim=0)
        a[:, m:] = 0.0
        out_a[b, :t] = a
    h = torch.einsum("bij,bic->bjc", out_a, v)
    return out_a, h


@pytest.mark.parametrize("fmt", ["split2", "split1", "f32"])
@pytest.mark.parametrize("B,T1,T2,masked", [(3, 37, 211, True), (2, 128, 800, True), (2, 200, 333, True), (1, 96, 577, False),
                                            (2, 256, 70, True), (4, 16, 31, True)])
def test_expand_vs_fp64_and_the_unfused_chain(B, T1, T2, masked, fmt):
    from efficient_tts_amd import lib as L, ops as P
    dev = _dev()
    C = 512
    g = torch.Generator().manual_seed(T1 * 13 + T2)
    tl = _lengths(B, T1, g) if masked else None
    ml = _lengths(B, T2, g) if masked else None
    # positions like the model's: increasing over the keys, spanning the frames
    e = (torch.rand(B, T1, generator=g) + 0.2).cumsum(1)
    e = e / e[:, -1:] * (T2 - 1)
    v = torch.randn(B, T1, C, generator=g)
    if masked:
        for b in range(B):
            v[b, int(tl[b]):] = 0.0                    # the value projection is zero at padded text (efficient_tts.py:157)
    rs1, rs2 = P.Rows(B, T1), P.Rows(B, T2)
    vf = P.F32Rows(rs1, C, dev)
    vf.view().copy_(v.to(dev))
    e_d = e.to(dev).contiguous()
    tl_d, ml_d = (tl.to(dev), ml.to(dev)) if masked else (None, None)

    # --- fused
    ra1 = torch.full((B, T1, T2), 7.0, device=dev)
    y_f = P.F32Rows(rs2, C, dev) if fmt == "f32" else None
    y_p = P.Plane.for_rows(rs2, C, 2 if fmt == "split2" else 1, dev) if fmt != "f32" else None
    y_l = P.Plane.for_rows(rs2, C, 1, dev) if fmt == "split1" else None
    P.expand(e=e_d, tl=tl_d, ml=ml_d, sigma=0.01, v=vf, rs1=rs1, rs2=rs2, alpha_out=ra1, y_f32=y_f, y=y_p, y_lo=y_l)

    # --- the chain it replaces
    ra0 = torch.empty(B, T1, T2, device=dev)
    ra_p = P.Plane.for_rows(rs2, T1, 2, dev)
    P.reconst_alpha(e_d, tl_d, ml_d, 0.01, ra0, ra_p, B, T1, T2, rs2.Tp)
    vt = P.Plane(B * C, T1, 2, dev)
    P.pack_vt(vf, vt, B, T1, rs1.Tp, C)
    lenm = torch.zeros(rs2.rows, device=dev)
    gap = torch.zeros(rs2.rows, device=dev)
    P.row_masks(ml_d if masked else torch.full((B,), T2, dtype=torch.int32, device=dev), rs2, gap, lenm)
    h0 = P.F32Rows(rs2, C, dev)
    P.gemm(a=ra_p, b_ptr=vt.ptr, ldb=vt.ld, m=T2, n=C, batch=B, a_batch_stride=rs2.Tp * ra_p.ld, b_batch_stride=C * vt.ld,
           rowmask_ptr=lenm.data_ptr(), rowmask_batch_stride=rs2.Tp, out_f32_ptr=h0.ptr, ldo=C, out_batch_stride=rs2.Tp * C)
    torch.cuda.synchronize()

    if fmt == "f32":
        got = y_f.view().cpu()
        full = y_f.buf.cpu()
    elif fmt == "split2":
        pl = y_p.buf[L.GUARD_LO:L.GUARD_LO + rs2.rows].view(torch.bfloat16).view(rs2.rows, y_p.nchunk, 2, 32).float()
        full = (pl[:, :, 0] + pl[:, :, 1]).reshape(B, rs2.Tp, C).cpu()
        got = full[:, :T2]
    else:
        hi = y_p.buf[L.GUARD_LO:L.GUARD_LO + rs2.rows].view(torch.bfloat16).float().view(B, rs2.Tp, C)
        lo = y_l.buf[L.GUARD_LO:L.GUARD_LO + rs2.rows].view(torch.bfloat16).float().view(B, rs2.Tp, C)
        full = (hi + lo).cpu()
        got = full[:, :T2]
    a64, h64 = _expand_ref64(e, v, tl, ml, 0.01, T2)
    scale = float(h64.abs().max())
    tol = 2e-5 * scale + (0.0 if fmt == "f32" else 2.0 ** -16 * scale)      # + the 16-bit hi/lo stream format
    assert float((got.double() - h64).abs().max()) <= tol
    assert float((ra1.cpu().double() - a64).abs().max()) <= 2e-6
    # against the chain: same maths in the same operand format, different summation order of the softmax only
    assert float((ra1 - ra0).abs().max()) <= 1e-6
    assert float((got - h0.view().cpu()).abs().max()) <= 2e-5 * scale + (0.0 if fmt == "f32" else 2.0 ** -16 * scale)
    # frames past an item's mel length are exact zeros; gap rows (t >= T2) are never touched
    if masked:
        for b in range(B):
            assert float(got[b, int(ml[b]):].abs().max() if int(ml[b]) < T2 else 0.0) == 0.0
            assert float(ra1[b, :, int(ml[b]):].abs().max() if int(ml[b]) < T2 else 0.0) == 0.0
            assert float(ra1[b, int(tl[b]):].abs().max() if int(tl[b]) < T1 else 0.0) == 0.0
    if fmt != "f32":
        assert float(full[:, T2:].abs().max()) == 0.0
    assert float((ra1.sum(1).cpu() - (a64.sum(1) > 0.5).float()).abs().max()) <= 1e-5        # columns sum to one where live


def test_expand_rejects_long_texts():
    from efficient_tts_amd import ops as P
    dev = _dev()
    rs1, rs2 = P.Rows(1, 300), P.Rows(1, 64)
    with pytest.raises(ValueError, match="T1 <= 256"):
        P.expand(e=torch.zeros(1, 300, device=dev), tl=None, ml=None, sigma=0.01, v=P.F32Rows(rs1, 512, dev), rs1=rs1, rs2=rs2,
                 y_f32=P.F32Rows(rs2, 512, dev))


def test_hardware_bf16_rounding_equals_the_integer_form():
    """every plane producer rounds with v_cvt_pk_bf16_f32; the integer round-to-nearest-even is the reference form (and what
    torch's .to(bfloat16) does): equal on every finite input, ties and denormals included"""
    from efficient_tts_amd import lib as L, ops as P
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    bits = torch.randint(-2 ** 31, 2 ** 31, (1 << 22,), generator=g, dtype=torch.int64).to(torch.int32)
    hi16 = torch.arange(0, 1 << 16, dtype=torch.int64)
    ties = torch.cat([(hi16 << 16) | 0x8000, (hi16 << 16) | 0x7fff, (hi16 << 16) | 0x8001, hi16 << 16])      # exact ties and their neighbours
    ties = torch.where(ties >= 2 ** 31, ties - 2 ** 32, ties).to(torch.int32)
    x = torch.cat([bits, ties]).view(torch.float32)
    x = x[torch.isfinite(x)].contiguous().to(dev)
    n = x.numel()
    y0 = torch.empty(n, dtype=torch.int16, device=dev)
    y1 = torch.empty(n, dtype=torch.int16, device=dev)
    lib = L.load()
    L.check(lib.efts_bf16_round(x.data_ptr(), y0.data_ptr(), n, 0, P._stream()), "efts_bf16_round")
    L.check(lib.efts_bf16_round(x.data_ptr(), y1.data_ptr(), n, 1, P._stream()), "efts_bf16_round")
    torch.cuda.synchronize()
    want = x.to(torch.bfloat16).view(torch.int16)
    big = x.abs() >= 2.0 ** -126                                  # normal numbers
    assert torch.equal(y1[big], want[big])
    bad = (y0 != y1)
    assert int(bad.sum()) == 0, f"{int(bad.sum())} of {n} differ, e.g. {x[bad][:4].tolist()}"


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("masked", [False, True])
def test_embed_conv_equals_embedding_plus_first_text_layer(precision, masked):
    """efts_embed_conv (embedding + text-encoder layer 0 as table look-ups over the 76 symbols) against efts_embed + the layer's
    efts_gemm launch on the same packed weights: equal to fp32 summation-order noise (same operand rounding: the tap table is built
    with one-tap efts_gemm launches in the model's operand format); masked mode = zero embedding and zero output beyond a length"""
    from efficient_tts_amd import EfficientTTSCNN, lib as L, ops as P
    from oracle import efts_oracle as O
    dev = _dev()
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=precision)
    m.load_state_dict(O.fill_params())
    m = m.to(dev).eval()
    pk = m._weights()
    tab = m._te0_table(pk)
    assert tab is not None and tab.shape == (5, 76, 512)
    B, T = 5, 61
    g = torch.Generator().manual_seed(9)
    text = torch.randint(0, 76, (B, T), generator=g).to(dev)
    lens = torch.tensor([61, 40, 1, 17, 60], dtype=torch.int32, device=dev)
    rs = P.Rows(B, T)
    C = 512
    gap, lenm = torch.zeros(rs.rows, device=dev), torch.zeros(rs.rows, device=dev)
    P.row_masks(lens, rs, gap, lenm)
    # reference chain: embedding (masked mode: zeroed beyond the length), then the layer through efts_gemm
    e_f, x_f, x_p = P.F32Rows(rs, C, dev), P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, m.split, dev)
    if masked:
        P.embed(text, m.text_embedding_table.weight.detach(), e_f, None, rs)
        P.mask_rows(e_f.ptr, lenm.data_ptr(), x_f, x_p, rs.rows, C)
    else:
        P.embed(text, m.text_embedding_table.weight.detach(), x_f, x_p, rs)
    w = pk["text_encoder.0"]
    y_ref, p_ref = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, m.split, dev)
    P.gemm(a=x_p, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=m.slope,
           bias=m.text_encoder.layers[0].conv[0].bias, resid_ptr=x_f.ptr, ldr=C, rowmask_ptr=(lenm if masked else gap).data_ptr(),
           out_f32_ptr=y_ref.ptr, ldo=C, out_plane=p_ref)
    y, p = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, m.split, dev)
    P.embed_conv(text, lens if masked else None, m.text_embedding_table.weight.detach(), tab, m.text_encoder.layers[0].conv[0].bias, m.slope, y, p, rs)
    torch.cuda.synchronize()
    scale = float(y_ref.buf.abs().max())
    assert float((y.buf - y_ref.buf).abs().max()) <= 2e-6 * scale
    assert float(y.buf[:L.GUARD_LO].abs().max()) == 0.0 and float(y.view()[2, 1:].abs().max() if masked else 0.0) == 0.0
    gaprows = y.buf[L.GUARD_LO:L.GUARD_LO + rs.rows].view(B, rs.Tp, C)[:, T:]
    assert float(gaprows.abs().max()) == 0.0
    pa, pb = p.buf.view(torch.bfloat16).float(), p_ref.buf.view(torch.bfloat16).float()
    assert float((pa - pb).abs().max()) <= 2.0 ** -7 * scale          # same values up to one bf16 ulp where the fp32 sums differ in the last bit


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_row_masks_pair_equals_two_row_masks_launches(dtype):
    """efts_row_masks_pair (make_non_pad_mask for both row spaces of a teacher-forced pass + the int32 copies of the caller's int64
    lengths, one launch) against efts_row_masks per row space (nntts/utils/nets_utils.py:170-254, efficient_tts.py:137-139)."""
    from efficient_tts_amd import ops as P
    dev = torch.device("cuda:0")
    B, T1, T2 = 7, 37, 211
    g = torch.Generator().manual_seed(3)
    l1 = torch.randint(1, T1 + 1, (B,), generator=g).to(dtype).to(dev)
    l2 = torch.randint(1, T2 + 1, (B,), generator=g).to(dtype).to(dev)
    rs1, rs2 = P.Rows(B, T1), P.Rows(B, T2)
    ref = [torch.full((r.rows,), -1.0, device=dev) for r in (rs1, rs1, rs2, rs2)]
    got = [torch.full((r.rows,), -1.0, device=dev) for r in (rs1, rs1, rs2, rs2)]
    with P.stream_scope():
        P.row_masks(l1.to(torch.int32), rs1, ref[0], ref[1])
        P.row_masks(l2.to(torch.int32), rs2, ref[2], ref[3])
        o1, o2 = P.row_masks_pair(l1, l2, rs1, rs2, *got)
        torch.cuda.synchronize()
    assert o1.dtype == torch.int32 and torch.equal(o1, l1.to(torch.int32)) and torch.equal(o2, l2.to(torch.int32))
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C", [(2, 800, 512), (3, 70, 128), (1, 128, 64), (2, 97, 192), (2, 40, 512), (2, 300, 80)])
def test_pack_vt_layout_both_kernels(B, T, C):
    """efts_pack_vt: x [B][T][C] fp32 in a row space -> the split-2 plane of x^T ([B * C] rows, K = T: 32 hi + 32 lo bf16 per 128-byte chunk), byte for
    byte against the same layout built with torch -- the 64 x 64 form (C % 64 == 0, T >= 64: 16-byte stores; round 6, the training step's mel-length
    transposes) and the 32 x 32 form (everything else); positions past T inside the last chunk are zero, nothing is written past the row's chunks."""
    from efficient_tts_amd import ops as P
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 100 + T + C)
    x = torch.randn(B, T, C, generator=g) * torch.logspace(-3, 1, C)          # a wide range of magnitudes per channel
    rs = P.Rows(B, T)
    xf = P.F32Rows(rs, C, dev)
    xf.view().copy_(x.to(dev))
    pl = P.Plane(B * C, T, 2, dev)
    pl.buf.fill_(0xAB)                                                       # (the kernel owns every byte of a row's chunks)
    P.pack_vt(xf, pl, B, T, rs.Tp, C)
    torch.cuda.synchronize()
    nchunk = (T + 31) // 32
    xt = torch.zeros(B, C, nchunk * 32)
    xt[:, :, :T] = x.transpose(1, 2)
    hi = xt.to(torch.bfloat16)
    lo = (xt - hi.float()).to(torch.bfloat16)
    want = torch.stack([hi.view(B, C, nchunk, 32), lo.view(B, C, nchunk, 32)], dim=3).reshape(B * C, nchunk * 64)
    got = pl.buf.view(torch.bfloat16).view(B * C, -1)[:, :nchunk * 64].cpu()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
