"""GPU (-m gpu): efts_resconv5 -- one residual k5 convolution layer on hi/lo planes, ResConv1d.forward of the reference
(nntts/layers/efts_modules.py:48-51) -- against efts_gemm (bit-exact: same per-element summation order) and against an
fp64 restatement of the layer; every tile height, multi-tile schedules, ragged row counts, both operand formats and all
input / output stream formats."""
import pytest
import torch

pytestmark = pytest.mark.gpu
C = 512


@pytest.fixture(autouse=True, params=["8-wave", "one-wave-per-SIMD"])
def rc_kernel(request):
    """every case of this file runs on both kernels of efts_resconv5 (include/efts_abi.h efts_resconv5_args.kernel): the 8-wave ping-pong
    kernel (the default) and the one-wave-per-SIMD kernel with the generated main loop (csrc/efts_resconv4.h), which takes over where it
    applies (bf16 planes, 5 taps) -- same tiles, plans and, bit for bit, the same results"""
    from efficient_tts_amd import ops as P
    prev, P.RC_KERNEL = P.RC_KERNEL, (0 if request.param == "8-wave" else 2)
    yield request.param
    P.RC_KERNEL = prev


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def bf16_split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


class Case:
    """x (exactly representable as hi + lo), weights, bias, gap mask of a B x T row space, and the efts_gemm result"""

    def __init__(self, B, T, split, seed=0):
        from efficient_tts_amd import lib as L, ops as P
        L.load(); L.require_device()
        self.L, self.P, self.split = L, P, split
        dev = _dev()
        torch.manual_seed(seed + 1000 * B + T)
        self.rs = rs = P.Rows(B, T)
        x = torch.randn(B, T, C, device=dev)
        hi, lo = bf16_split(x)
        x16 = hi.float() + lo.float()
        hi, lo = bf16_split(x16)                      # bf16(x16) may differ from bf16(x) at rounding ties
        self.x16 = x16
        self.a = P.Plane.for_rows(rs, C, split, dev)
        P.pack_rows(x16, None, self.a, rs)
        self.a_lo = None
        if split == 1:
            self.a_lo = P.Plane.for_rows(rs, C, 1, dev)
            P.pack_rows(lo.float().contiguous(), None, self.a_lo, rs)
        self.xf = P.F32Rows(rs, C, dev); self.xf.view().copy_(x16)
        self.w = (torch.randn(C, C, 5, device=dev) * 0.02).contiguous()
        self.pw = P.PackedWeight(C, C, 5, split, dev); self.pw.pack(self.w)
        self.bias = torch.randn(C, device=dev)
        lens = torch.randint(max(1, T // 2), T + 1, (B,), dtype=torch.int32, device=dev)
        self.mask = torch.zeros(rs.rows, device=dev)
        P.row_masks(lens, rs, None, self.mask)        # a LENGTH mask: rows past an item's length are zeroed too
        self.o_ref, self.p_ref = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, 2, dev)
        P.gemm(a=self.a, b_ptr=self.pw.ptr, ldb=self.pw.ld, b_tap_stride=self.pw.tap_stride, taps=5, m=rs.rows, n=C,
               act=L.ACT_LEAKY, slope=0.1, bias=self.bias, resid_ptr=self.xf.ptr, ldr=C, rowmask_ptr=self.mask.data_ptr(),
               out_f32_ptr=self.o_ref.ptr, ldo=C, out_plane=self.p_ref)
        torch.cuda.synchronize()

    def run(self, mode, plan=None):
        P, rs, dev = self.P, self.rs, _dev()
        o = P.F32Rows(rs, C, dev)
        if mode == "f32":          # fp32 residual in, fp32 + split-2 plane out
            y, yl = P.Plane.for_rows(rs, C, 2, dev), None
            P.resconv5(x=self.a, x_f32_ptr=self.xf.ptr, ldr=C, w=self.pw, m=rs.rows, n=C, bias=self.bias,
                       rowmask_ptr=self.mask.data_ptr(), y_f32_ptr=o.ptr, ldo=C, y=y, plan=plan)
        else:                      # planes in, planes of the same format out (+ fp32 for the comparison)
            y = P.Plane.for_rows(rs, C, self.split, dev)
            yl = P.Plane.for_rows(rs, C, 1, dev) if self.split == 1 else None
            P.resconv5(x=self.a, x_lo=self.a_lo, w=self.pw, m=rs.rows, n=C, bias=self.bias, rowmask_ptr=self.mask.data_ptr(),
                       y_f32_ptr=o.ptr, ldo=C, y=y, y_lo=yl, plan=plan)
        torch.cuda.synchronize()
        return o, y, yl

    def check(self, mode, plan=None):
        o, y, yl = self.run(mode, plan)
        assert torch.equal(o.buf, self.o_ref.buf), f"fp32 differs: {(o.buf - self.o_ref.buf).abs().max().item():.3e}"
        if mode == "f32" or self.split == 2:
            assert torch.equal(y.buf, self.p_ref.buf)
        else:
            rh, rl = bf16_split(self.o_ref.buf)
            n = self.rs.alloc
            assert torch.equal(y.buf.view(torch.bfloat16).view(n, -1)[:, :C], rh)
            assert torch.equal(yl.buf.view(torch.bfloat16).view(n, -1)[:, :C], rl)


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("B,T", [(3, 37), (5, 300), (1, 1), (2, 801)])
@pytest.mark.parametrize("mode", ["f32", "planes"])
def test_resconv5_equals_gemm_automatic_schedule(split, B, T, mode):
    Case(B, T, split).check(mode)


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("classes", [[[2]], [[3]], [[4]], [[5]], [[6]], [[7]], [[8]], [[4], [2]], [[6, 2]], [[7, 6], [6, 7]], [[2, 3, 4, 5]],
                                     [[8, 3, 7]], [[8], [6], [4], [2]], [[5], [3], [7]]])
def test_resconv5_every_tile_height_and_multi_tile_schedules(split, classes):
    """explicit plans: every tile height in half units (odd ones split 4+3, 3+2, 2+1 blocks over the two wave rows), tiles of
    different heights in one workgroup (window buffers / ring slots carried across tiles, the next tile's first operands
    requested before the epilogue), 1 .. 4 classes"""
    c = Case(5, 300, split, seed=7)
    plan = c.P.make_plan(c.rs.rows, classes)
    c.check("planes", plan)
    c.check("f32", plan)


def test_resconv5_vs_fp64_layer():
    """the layer itself against an fp64 restatement: y = (x + LeakyReLU(conv1d_k5(x_hi) + b)) * mask per item"""
    c = Case(3, 70, 2, seed=3)
    o, _, _ = c.run("planes")
    rs = c.rs
    x = c.x16.double()                                                       # [B, T, C]
    hi, lo = bf16_split(c.w)
    w = (hi.double() + lo.double())                                          # bf16x3 operands: 16 mantissa bits of w
    conv = torch.nn.functional.conv1d(x.transpose(1, 2), w, c.bias.double(), padding=2).transpose(1, 2)
    y = x + torch.nn.functional.leaky_relu(conv, 0.1)
    y = y * c.mask.view(rs.B, rs.Tp)[:, :rs.T, None].double()
    got = o.view().double()
    err = (got - y).abs().max().item()
    assert err <= 2e-4, err                                                  # dropped lo*lo term + fp32 accumulation of 2560 products


def test_resconv5_full_size_equals_gemm():
    """BASELINE config 2 row space (64 x 802 rows), both operand formats, the automatic two-class schedule"""
    for split in (1, 2):
        Case(64, 800, split, seed=11).check("planes")


def test_gemm_writes_the_remainder_plane():
    """efts_gemm(out_bf16_lo): hi + lo planes of a split-1 output rebuild the fp32 result to 16 mantissa bits"""
    from efficient_tts_amd import lib as L, ops as P
    c = Case(2, 50, 1, seed=5)
    dev = _dev()
    rs = c.rs
    o, y, yl = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, 1, dev), P.Plane.for_rows(rs, C, 1, dev)
    P.gemm(a=c.a, b_ptr=c.pw.ptr, ldb=c.pw.ld, b_tap_stride=c.pw.tap_stride, taps=5, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=0.1,
           bias=c.bias, resid_ptr=c.xf.ptr, ldr=C, rowmask_ptr=c.mask.data_ptr(), out_f32_ptr=o.ptr, ldo=C, out_plane=y, out_plane_lo=yl)
    torch.cuda.synchronize()
    rh, rl = bf16_split(o.buf)
    assert torch.equal(y.buf.view(torch.bfloat16).view(rs.alloc, -1)[:, :C], rh)
    assert torch.equal(yl.buf.view(torch.bfloat16).view(rs.alloc, -1)[:, :C], rl)


@pytest.mark.parametrize("split", [1, 2])
def test_resconv5_race_screen(split):
    """The ping-pong loop orders its LDS-DMA writes and fragment reads by counted waits and barriers alone; a misplaced one shows
    up as a rare wrong tile, not as a steady failure.  120 launches at the full size (30 240 tiles, 1.2 M (chunk, tap) steps per
    workgroup row), back to back and beside a second stream that keeps the fabric busy with copies, every result compared bit
    for bit with efts_gemm's; then the same under an odd-height multi-tile plan."""
    c = Case(64, 800, split, seed=21)
    P, rs, dev = c.P, c.rs, _dev()
    y = P.Plane.for_rows(rs, C, split, dev)
    yl = P.Plane.for_rows(rs, C, 1, dev) if split == 1 else None
    o = P.F32Rows(rs, C, dev)
    side = torch.cuda.Stream(device=dev)
    junk_a, junk_b = torch.empty(64 << 20, dtype=torch.uint8, device=dev), torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    bad = 0
    for plan in (None, P.make_plan(rs.rows, [[5, 3, 5], [3, 7, 3]])):
        for it in range(60):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk_b.copy_(junk_a)                      # 128 MB of unrelated traffic beside the launch
            o.buf.zero_(); y.buf.zero_()
            P.resconv5(x=c.a, x_lo=c.a_lo, w=c.pw, m=rs.rows, n=C, bias=c.bias, rowmask_ptr=c.mask.data_ptr(), y_f32_ptr=o.ptr, ldo=C,
                       y=y, y_lo=yl, plan=plan)
            bad += int(not torch.equal(o.buf, c.o_ref.buf))
            if split == 2:
                bad += int(not torch.equal(y.buf, c.p_ref.buf))
        torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 120 launches differ from efts_gemm"


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("shapes", [((5, 300), (3, 37)), ((2, 801), (2, 130)), ((3, 37), (5, 300)), ((1, 1), (1, 1)), ((16, 800), (16, 128))])
@pytest.mark.parametrize("classes", [None, [[7, 6], [6, 7]], [[3], [2]], [[8, 2, 5]]])
def test_resconv5_multi_two_layers_in_one_launch(split, shapes, classes):
    """efts_resconv5_multi: two independent layers (own operands, weights, bias, mask, outputs; different row counts) laid end to
    end in ONE persistent launch == each layer through efts_gemm, bit for bit -- for the automatic schedule and explicit ones whose
    tiles get cut at the layer boundary (tile heights change there, the operand streams carried across tiles switch layers)"""
    ca, cb = Case(*shapes[0], split, seed=3), Case(*shapes[1], split, seed=11)
    P, dev = ca.P, _dev()
    outs, layers = [], []
    for c, mode in ((ca, "planes"), (cb, "f32")):
        o = P.F32Rows(c.rs, C, dev)
        if mode == "f32":
            y, yl = P.Plane.for_rows(c.rs, C, 2, dev), None
            kw = dict(x=c.a, x_f32_ptr=c.xf.ptr, ldr=C, w=c.pw, m=c.rs.rows, n=C, bias=c.bias, rowmask_ptr=c.mask.data_ptr(),
                      y_f32_ptr=o.ptr, ldo=C, y=y)
        else:
            y = P.Plane.for_rows(c.rs, C, split, dev)
            yl = P.Plane.for_rows(c.rs, C, 1, dev) if split == 1 else None
            kw = dict(x=c.a, x_lo=c.a_lo, w=c.pw, m=c.rs.rows, n=C, bias=c.bias, rowmask_ptr=c.mask.data_ptr(), y_f32_ptr=o.ptr, ldo=C,
                      y=y, y_lo=yl)
        layers.append(kw)
        outs.append((c, mode, o, y, yl))
    if classes is not None:
        layers[0]["plan"] = P.make_plan(ca.rs.rows + cb.rs.rows, classes)
    P.resconv5_multi(layers)
    torch.cuda.synchronize()
    for c, mode, o, y, yl in outs:
        assert torch.equal(o.buf, c.o_ref.buf), f"fp32 differs: {(o.buf - c.o_ref.buf).abs().max().item():.3e}"
        if mode == "f32" or split == 2:
            assert torch.equal(y.buf, c.p_ref.buf)
        else:
            rh, rl = bf16_split(c.o_ref.buf)
            n = c.rs.alloc
            assert torch.equal(y.buf.view(torch.bfloat16).view(n, -1)[:, :C], rh)
            assert torch.equal(yl.buf.view(torch.bfloat16).view(n, -1)[:, :C], rl)


class Case3(Case):
    """a k3 layer on the same operands: weights [C][C][3], reference = efts_gemm taps 3 (residual layer, or Conv1d + ReLU without
    the residual term and without a mask: the duration predictor's layer, nntts/layers/duration_predictor.py:57)"""

    def __init__(self, B, T, split, seed=0, plain=False):
        super().__init__(B, T, split, seed)
        L, P, rs, dev = self.L, self.P, self.rs, _dev()
        self.plain = plain
        self.w = (torch.randn(C, C, 3, device=dev) * 0.03).contiguous()
        self.pw = P.PackedWeight(C, C, 3, split, dev); self.pw.pack(self.w)
        self.o_ref, self.p_ref = P.F32Rows(rs, C, dev), P.Plane.for_rows(rs, C, 2, dev)
        if plain:
            P.gemm(a=self.a, b_ptr=self.pw.ptr, ldb=self.pw.ld, b_tap_stride=self.pw.tap_stride, taps=3, m=rs.rows, n=C, act=L.ACT_RELU,
                   bias=self.bias, out_f32_ptr=self.o_ref.ptr, ldo=C, out_plane=self.p_ref, tiling=L.TILING_GENERIC)
        else:
            P.gemm(a=self.a, b_ptr=self.pw.ptr, ldb=self.pw.ld, b_tap_stride=self.pw.tap_stride, taps=3, m=rs.rows, n=C, act=L.ACT_LEAKY,
                   slope=0.1, bias=self.bias, resid_ptr=self.xf.ptr, ldr=C, rowmask_ptr=self.mask.data_ptr(), out_f32_ptr=self.o_ref.ptr, ldo=C,
                   out_plane=self.p_ref, tiling=L.TILING_GENERIC)
        torch.cuda.synchronize()

    def kwargs(self, o, y):
        if self.plain:
            return dict(x=self.a, x_lo=self.a_lo, w=self.pw, m=self.rs.rows, n=C, bias=self.bias, slope=0.0, y_f32_ptr=o.ptr, ldo=C, y=y,
                        taps=3, no_residual=True)
        return dict(x=self.a, x_lo=self.a_lo, w=self.pw, m=self.rs.rows, n=C, bias=self.bias, rowmask_ptr=self.mask.data_ptr(),
                    y_f32_ptr=o.ptr, ldo=C, y=y, taps=3)


@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("B,T", [(3, 37), (5, 300), (2, 801), (1, 1)])
@pytest.mark.parametrize("plain", [False, True])
def test_resconv5_k3_layers_equal_gemm(split, B, T, plain):
    """efts_resconv5_args.taps = 3 (a k_size = 3 ResConv1d layer; with no_residual and slope 0 the duration predictor's Conv1d + ReLU)
    == efts_gemm taps 3 on the same operands, bit for bit"""
    c = Case3(B, T, split, seed=5, plain=plain)
    P, dev = c.P, _dev()
    o, y = P.F32Rows(c.rs, C, dev), P.Plane.for_rows(c.rs, C, 2, dev)
    P.resconv5(**c.kwargs(o, y))
    torch.cuda.synchronize()
    live = slice(c.L.GUARD_LO, c.L.GUARD_LO + c.rs.rows)
    assert torch.equal(o.buf[live], c.o_ref.buf[live]), f"fp32 differs: {(o.buf - c.o_ref.buf).abs().max().item():.3e}"
    assert torch.equal(y.buf[live].view(torch.bfloat16), c.p_ref.buf[live].view(torch.bfloat16))


@pytest.mark.parametrize("split", [1, 2])
def test_resconv5_multi_k5_layer_with_a_k3_rider(split):
    """a decoder-like k5 residual layer and the duration predictor's k3 Conv1d + ReLU (no residual, no mask, fp32 output) in one
    grouped launch: each equals its own efts_gemm launch bit for bit"""
    ca, cb = Case(16, 800, split, seed=3), Case3(16, 128, split, seed=11, plain=True)
    P, dev = ca.P, _dev()
    oa, ya = P.F32Rows(ca.rs, C, dev), P.Plane.for_rows(ca.rs, C, 2, dev)
    ob, yb = P.F32Rows(cb.rs, C, dev), P.Plane.for_rows(cb.rs, C, 2, dev)
    ka = dict(x=ca.a, x_lo=ca.a_lo, w=ca.pw, m=ca.rs.rows, n=C, bias=ca.bias, rowmask_ptr=ca.mask.data_ptr(), y_f32_ptr=oa.ptr, ldo=C, y=ya)
    P.resconv5_multi([ka, cb.kwargs(ob, yb)])
    torch.cuda.synchronize()
    assert torch.equal(oa.buf, ca.o_ref.buf)
    live = slice(cb.L.GUARD_LO, cb.L.GUARD_LO + cb.rs.rows)
    assert torch.equal(ob.buf[live], cb.o_ref.buf[live])
    assert torch.equal(yb.buf[live].view(torch.bfloat16), cb.p_ref.buf[live].view(torch.bfloat16))
