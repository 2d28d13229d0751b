"""Thin typed wrappers: torch tensors (device memory + stream plumbing) -> C-ABI calls.

Nothing here computes: every function marshals pointers/sizes into libefts_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import lib as L


# When set to a list, every efts_gemm launch is bracketed by HIP events recorded on the launch
# stream (torch's current stream) and (tag, start, end) is appended; tag = (taps, m, n).
# Default `tiling` of efts_gemm launches (L.TILING_*): AUTO in the product; the equality tests between the kernels flip it
GEMM_TILING = 0
# `kernel` of every efts_resconv5 launch (include/efts_abi.h): 0 the 8-wave kernel (the product), 2 the one-wave-per-SIMD kernel where it applies.
# Part of every graph tag (model.LaunchOptions.tag, train.switch_tag): a captured pass is valid for the kernel it was captured with
RC_KERNEL = 0
PROFILE = None
# optional filter: only launches whose tag equals PROFILE_TAG are bracketed (keeps the host light)
PROFILE_TAG = None
PROFILE_INFO = {}          # tag -> (K = input channels, split, batch) of the bracketed efts_gemm launches (bench.py: FLOPs / bytes of a tag)

_cached_stream = None


class stream_scope:
    """Resolve torch's current HIP stream ONCE for a whole forward / training step (the lookup costs
    ~8 us per call and there are hundreds of launches per step)."""

    def __enter__(self):
        global _cached_stream
        self.prev = _cached_stream
        _cached_stream = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *a):
        global _cached_stream
        _cached_stream = self.prev


class on_stream:
    """Route the C-ABI launches (and torch's own ops) to `stream` inside a stream_scope."""

    def __init__(self, stream: "torch.cuda.Stream"):
        self.stream = stream

    def __enter__(self):
        global _cached_stream
        self.prev = _cached_stream
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        _cached_stream = self.stream.cuda_stream
        return self

    def __exit__(self, *a):
        global _cached_stream
        _cached_stream = self.prev
        self.ctx.__exit__(*a)


def _stream() -> int:
    return _cached_stream if _cached_stream is not None else torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def chunk_k(split: int) -> int:
    """k's per 128-byte chunk of an operand plane"""
    return 64 if split == 1 else 32


class Rows:
    """Geometry of a padded row space (include/efts_abi.h "Row space")."""

    def __init__(self, B: int, T: int, gap: int = L.GAP):
        # gap: zero rows behind every item, >= the widest convolution's padding on this row space (L.GAP = 2 for k5; a model with
        # k_size 7 / 9 / 11 builds its row spaces with 3 / 4 / 5)
        self.B, self.T, self.Tp, self.gap = B, T, T + gap, gap
        self.rows = B * self.Tp
        self.alloc = L.GUARD_LO + roundup(self.rows, L.TILE_M) + L.GUARD_HI


class F32Rows:
    """fp32 [rows, C] stream with zero guard rows; `.ptr` points at row 0."""

    def __init__(self, rs: Rows, c: int, device):
        self.rs, self.c = rs, c
        self.buf = torch.zeros(rs.alloc, c, dtype=torch.float32, device=device)
        self.ptr = self.buf.data_ptr() + L.GUARD_LO * c * 4

    def view(self) -> torch.Tensor:
        """[B, T, C] strided view of the valid rows (no copy)."""
        rs = self.rs
        return self.buf[L.GUARD_LO:L.GUARD_LO + rs.rows].view(rs.B, rs.Tp, self.c)[:, :rs.T]


class Plane:
    """bf16 MFMA operand plane [rows, nchunk*128 B] (split 1: bf16; split 2: hi/lo interleaved)."""

    def __init__(self, nrows_alloc: int, k: int, split: int, device, guard_lo: int = 0):
        self.split, self.k = split, k
        self.nchunk = (k + chunk_k(split) - 1) // chunk_k(split)
        self.ld = self.nchunk * 128
        self.buf = torch.zeros(nrows_alloc, self.ld, dtype=torch.uint8, device=device)
        self.ptr = self.buf.data_ptr() + guard_lo * self.ld

    @staticmethod
    def for_rows(rs: Rows, k: int, split: int, device) -> "Plane":
        return Plane(rs.alloc, k, split, device, guard_lo=L.GUARD_LO)


def gemm(*, a: Plane, a_ptr: Optional[int] = None, b_ptr: int, ldb: int, b_tap_stride: int = 0, taps: int = 1,
         m: int, n: int, batch: int = 1, a_batch_stride: int = 0, b_batch_stride: int = 0, alpha: float = 1.0,
         act: int = L.ACT_NONE, slope: float = 0.0, bias: Optional[torch.Tensor] = None,
         resid_ptr: Optional[int] = None, ldr: int = 0, resid_batch_stride: int = 0,
         rowmask_ptr: Optional[int] = None, rowmask_batch_stride: int = 0,
         out_f32_ptr: Optional[int] = None, ldo: int = 0, out_batch_stride: int = 0,
         out_plane: Optional[Plane] = None, out_plane_ptr: Optional[int] = None, outb_batch_stride: int = 0,
         nchunk: Optional[int] = None, batch2: int = 0, a_batch2_stride: int = 0, b_batch2_stride: int = 0,
         out_batch2_stride: int = 0, dilation: int = 1, plane_act: bool = False, plane_slope: float = 0.0,
         out_plane_lo: Optional[Plane] = None, tiling: Optional[int] = None, sign_mask_ptr: Optional[int] = None,
         soft_index: Optional[torch.Tensor] = None, key_len: Optional[torch.Tensor] = None, query_len: Optional[torch.Tensor] = None,
         drop_p: float = 0.0, drop_seed: int = 0, sqerr_target: Optional[torch.Tensor] = None, sqerr_part: Optional[torch.Tensor] = None) -> None:
    g = L.GemmArgs()
    g.a, g.lda, g.a_batch_stride = (a_ptr if a_ptr is not None else a.ptr), a.ld, a_batch_stride
    g.b, g.ldb, g.b_tap_stride, g.b_batch_stride = b_ptr, ldb, b_tap_stride, b_batch_stride
    g.split, g.taps, g.m, g.n = a.split, taps, m, n
    g.nchunk, g.batch = (a.nchunk if nchunk is None else nchunk), batch
    g.alpha, g.act, g.slope = alpha, act, slope
    g.bias = _p(bias)
    g.resid, g.ldr, g.resid_batch_stride = resid_ptr, ldr, resid_batch_stride
    g.rowmask, g.rowmask_batch_stride = rowmask_ptr, rowmask_batch_stride
    g.out_f32, g.ldo, g.out_batch_stride = out_f32_ptr, ldo, out_batch_stride
    if out_plane is not None:
        g.out_bf16 = out_plane_ptr if out_plane_ptr is not None else out_plane.ptr
        g.ldob, g.out_split = out_plane.ld, out_plane.split
    g.outb_batch_stride = outb_batch_stride
    g.batch2, g.a_batch2_stride, g.b_batch2_stride, g.out_batch2_stride = batch2, a_batch2_stride, b_batch2_stride, out_batch2_stride
    g.dilation, g.plane_act, g.plane_slope = dilation, int(plane_act), plane_slope
    g.out_bf16_lo = None if out_plane_lo is None else out_plane_lo.ptr
    g.tiling = GEMM_TILING if tiling is None else tiling
    g.sign_mask = sign_mask_ptr
    g.drop_p, g.drop_seed = drop_p, drop_seed & 0xFFFFFFFF
    if sqerr_part is not None:                  # sum (out - target)^2 per workgroup and wave in the epilogue; target: fp32 [batch][m][n] contiguous
        g.sqerr_target, g.ld_target, g.target_batch_stride, g.sqerr_part = sqerr_target.data_ptr(), n, m * n, sqerr_part.data_ptr()
    if soft_index is not None:                  # expected key index of the row softmax, computed in the epilogue (int32 lengths per batch item)
        g.soft_index, g.key_len, g.query_len = soft_index.data_ptr(), key_len.data_ptr(), query_len.data_ptr()
    if PROFILE is not None and (PROFILE_TAG is None or PROFILE_TAG == (taps, m, n)):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        L.check(L.load().efts_gemm(C.byref(g), _stream()), "efts_gemm")
        s1.record()
        PROFILE.append(((taps, m, n), s0, s1))
        PROFILE_INFO[(taps, m, n)] = (a.k, a.split, batch)
        return
    L.check(L.load().efts_gemm(C.byref(g), _stream()), "efts_gemm")


def _resconv5_fill(g, *, x: Plane, x_lo: Optional[Plane] = None, x_f32_ptr: Optional[int] = None, ldr: int = 0, w: "PackedWeight",
                   m: int, n: int, bias: Optional[torch.Tensor] = None, slope: float = 0.1, rowmask_ptr: Optional[int] = None,
                   y_f32_ptr: Optional[int] = None, ldo: int = 0, y: Optional[Plane] = None, y_lo: Optional[Plane] = None,
                   plan=None, taps: int = 5, no_residual: bool = False, sign_bits_ptr: Optional[int] = None,
                   act_bwd_sign_ptr: Optional[int] = None, act_bwd_slope: float = 0.0, act_bwd_bias_part: Optional[torch.Tensor] = None) -> None:
    if plan is not None:
        g.plan = plan
    g.kernel = RC_KERNEL
    if act_bwd_sign_ptr is not None:                 # training backward: the activation backward of the layer below in this dgrad launch's epilogue
        g.act_bwd_sign, g.act_bwd_slope = act_bwd_sign_ptr, act_bwd_slope
        if act_bwd_bias_part is not None:
            g.act_bwd_bias_part, g.act_bwd_bias_rows = act_bwd_bias_part.data_ptr(), act_bwd_bias_part.shape[0]
    g.taps, g.no_residual, g.sign_bits = taps, int(no_residual), sign_bits_ptr
    g.x, g.x_lo, g.ldx = x.ptr, (None if x_lo is None else x_lo.ptr), x.ld
    g.x_f32, g.ldr = x_f32_ptr, ldr
    g.w, g.ldw, g.w_tap_stride = w.ptr, w.ld, w.tap_stride
    g.split, g.m, g.n, g.nchunk = x.split, m, n, x.nchunk
    g.bias, g.slope, g.rowmask = _p(bias), slope, rowmask_ptr
    g.y_f32, g.ldo = y_f32_ptr, ldo
    if y is not None:
        g.y, g.ldy, g.y_split = y.ptr, y.ld, y.split
        g.y_lo = None if y_lo is None else y_lo.ptr


def resconv5(**kw) -> None:
    """one residual k5 convolution layer on hi/lo planes (efts_resconv5); plan: explicit tile schedule (make_plan)"""
    g = L.ResConv5Args()
    _resconv5_fill(g, **kw)
    m, n = kw["m"], kw["n"]
    if PROFILE is not None and (PROFILE_TAG is None or PROFILE_TAG == (5, m, n)):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        L.check(L.load().efts_resconv5(C.byref(g), _stream()), "efts_resconv5")
        s1.record()
        PROFILE.append(((5, m, n), s0, s1))
        return
    L.check(L.load().efts_resconv5(C.byref(g), _stream()), "efts_resconv5")


def resconv5_multi(layers) -> None:
    """several independent residual layers of the same geometry in ONE persistent launch (efts_resconv5_multi); `layers`: a list
    of resconv5() keyword dicts, the long layer first"""
    arr = (L.ResConv5Args * len(layers))()
    for g, kw in zip(arr, layers):
        _resconv5_fill(g, **kw)
    tag = (5, sum(kw["m"] for kw in layers), layers[0]["n"])          # (a grouped launch is tagged with its combined rows)
    if PROFILE is not None and (PROFILE_TAG is None or PROFILE_TAG == tag):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        L.check(L.load().efts_resconv5_multi(arr, len(layers), _stream()), "efts_resconv5_multi")
        s1.record()
        PROFILE.append((tag, s0, s1))
        return
    L.check(L.load().efts_resconv5_multi(arr, len(layers), _stream()), "efts_resconv5_multi")


def frame_linear(*, x: torch.Tensor, w: "PackedWeight", bias: Optional[torch.Tensor], act: int, slope: float, rs: Rows,
                 y: Optional[Plane] = None, y_lo: Optional[Plane] = None, y_f32: Optional[F32Rows] = None, max_workgroups: int = 0) -> None:
    """y[b * Tp + t] = act(x[b, t] . W^T + bias) straight from the caller's fp32 frames (efts_frame_linear): x [B, T, cin] contiguous"""
    g = L.FrameLinearArgs()
    g.x, g.w, g.ldw, g.split = x.data_ptr(), w.ptr, w.ld, w.split
    g.bias, g.act, g.slope = _p(bias), act, slope
    g.B, g.T, g.Tp, g.cin, g.n = rs.B, rs.T, rs.Tp, x.shape[2], w.cout
    g.max_workgroups = max_workgroups
    if y_f32 is not None:
        g.y_f32, g.ldo = y_f32.ptr, y_f32.c
    if y is not None:
        g.y, g.ldy, g.y_split = y.ptr, y.ld, y.split
        g.y_lo = None if y_lo is None else y_lo.ptr
    L.check(L.load().efts_frame_linear(C.byref(g), _stream()), "efts_frame_linear")


_bias_rows_cache = {}


def resconv5_bias_rows(m: int, n: int) -> int:
    """rows of the column-sum table a dgrad launch with the fused activation backward fills (efts_resconv5_bias_rows)"""
    if (m, n) not in _bias_rows_cache:
        rc = L.load().efts_resconv5_bias_rows(m, n)
        if rc < 0:
            L.check(rc, "efts_resconv5_bias_rows")
        _bias_rows_cache[(m, n)] = rc
    return _bias_rows_cache[(m, n)]


def resconv5_plan(m: int, n: int, cus: int = 0):
    """the automatic tile schedule as (groups, [(rows, [ni, ...]), ...])"""
    buf = (C.c_int32 * L.RC_PLAN_INTS)()
    rc = L.load().efts_resconv5_plan(m, n, cus, buf, L.RC_PLAN_INTS)
    if rc < 0:
        L.check(rc, "efts_resconv5_plan")
    classes = []
    for c in range(buf[1]):
        q = 2 + c * 10
        classes.append((buf[q], [buf[q + 2 + t] for t in range(buf[q + 1])]))
    return buf[0], classes


_plan_cache = {}


def resconv5_plan_buf(m: int, n: int, cus: int):
    """the automatic schedule for `cus` compute units as a plan buffer efts_resconv5 takes (cached: a host-side table)"""
    key = (m, n, cus)
    if key not in _plan_cache:
        buf = (C.c_int32 * L.RC_PLAN_INTS)()
        rc = L.load().efts_resconv5_plan(m, n, cus, buf, L.RC_PLAN_INTS)
        if rc < 0:
            L.check(rc, "efts_resconv5_plan")
        _plan_cache[key] = buf
    return _plan_cache[key]


def make_plan(m: int, classes):
    """explicit schedule for efts_resconv5: classes = [[h, ...], ...] (tile heights per class in half units of 32 window rows,
    2..8; a tile yields 32 h - 4 rows); enough groups to cover m"""
    buf = (C.c_int32 * L.RC_PLAN_INTS)()
    rows = [sum(32 * ni - 4 for ni in cl) for cl in classes]
    full, rem, groups = m // sum(rows), m % sum(rows), 0
    groups = full * len(classes)
    for r in rows:
        if rem <= 0:
            break
        groups, rem = groups + 1, rem - r
    buf[0], buf[1] = groups, len(classes)
    for c, cl in enumerate(classes):
        q = 2 + c * 10
        buf[q], buf[q + 1] = rows[c], len(cl)
        for t, ni in enumerate(cl):
            buf[q + 2 + t] = ni
    return buf


class PackedWeight:
    """B operand plane [taps][cout][ld] of one Conv1d / Linear weight."""

    def __init__(self, cout: int, cin: int, taps: int, split: int, device):
        self.cout, self.cin, self.taps, self.split = cout, cin, taps, split
        self.nchunk = (cin + chunk_k(split) - 1) // chunk_k(split)
        self.ld = self.nchunk * 128
        self.buf = torch.zeros(taps, cout, self.ld, dtype=torch.uint8, device=device)
        self.ptr = self.buf.data_ptr()
        self.tap_stride = cout * self.ld

    def pack(self, w: torch.Tensor, g: Optional[torch.Tensor] = None, w_out: Optional[torch.Tensor] = None) -> None:
        """w: fp32 [cout, cin, taps] or [cout, cin] contiguous; g: weight-norm gain [cout,1,1] or None."""
        assert w.is_contiguous() and w.dtype == torch.float32
        L.check(L.load().efts_pack_weight(w.data_ptr(), _p(g), _p(w_out), self.ptr, self.ld, self.cout, self.cin,
                                          self.taps, self.split, _stream()), "efts_pack_weight")


def row_masks(lengths_i32: torch.Tensor, rs: Rows, gap: Optional[torch.Tensor], lenmask: Optional[torch.Tensor]) -> None:
    L.check(L.load().efts_row_masks(lengths_i32.data_ptr(), _p(gap), _p(lenmask), rs.B, rs.T, rs.Tp, _stream()),
            "efts_row_masks")


def row_masks_pair(len1: torch.Tensor, len2: torch.Tensor, rs1: Rows, rs2: Rows, gap1, lm1, gap2, lm2):
    """masks of both row spaces + int32 copies of the (int64 or int32, device) length tensors in one launch -> (tl, ml) int32"""
    assert len1.dtype == len2.dtype and len1.dtype in (torch.int64, torch.int32) and len1.is_contiguous() and len2.is_contiguous()
    o1 = torch.empty(rs1.B, dtype=torch.int32, device=len1.device)
    o2 = torch.empty(rs1.B, dtype=torch.int32, device=len1.device)
    L.check(L.load().efts_row_masks_pair(len1.data_ptr(), len2.data_ptr(), int(len1.dtype == torch.int64), o1.data_ptr(), o2.data_ptr(), _p(gap1), _p(lm1),
                                         _p(gap2), _p(lm2), rs1.B, rs1.T, rs1.Tp, rs2.T, rs2.Tp, _stream()), "efts_row_masks_pair")
    return o1, o2


def embed(ids: torch.Tensor, table: torch.Tensor, out: Optional[F32Rows], plane: Optional[Plane], rs: Rows) -> None:
    c = table.shape[1]
    L.check(L.load().efts_embed(ids.data_ptr(), table.data_ptr(), None if out is None else out.ptr,
                                None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                rs.B, rs.T, rs.Tp, c, table.shape[0], 1 if plane is None else plane.split, _stream()),
            "efts_embed")


def embed_conv(ids: torch.Tensor, lens: Optional[torch.Tensor], table: torch.Tensor, tap_table: torch.Tensor, bias: Optional[torch.Tensor],
               slope: float, out: Optional[F32Rows], plane: Optional[Plane], rs: Rows) -> None:
    """embedding + the first residual convolution of the text encoder as table look-ups (efts_embed_conv); tap_table [taps][V][c]"""
    taps, nsym, c = tap_table.shape
    L.check(L.load().efts_embed_conv(ids.data_ptr(), _p(lens), table.data_ptr(), tap_table.data_ptr(), _p(bias), slope,
                                     None if out is None else out.ptr, None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                     rs.B, rs.T, rs.Tp, c, nsym, taps, 1 if plane is None else plane.split, _stream()), "efts_embed_conv")


def pack_rows(x: torch.Tensor, out: Optional[F32Rows], plane: Optional[Plane], rs: Rows) -> None:
    c = x.shape[-1]
    kp = roundup(c, 4) if plane is None else plane.nchunk * chunk_k(plane.split)
    L.check(L.load().efts_pack_rows(x.data_ptr(), None if out is None else out.ptr,
                                    None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                    rs.B, rs.T, rs.Tp, c, kp, 1 if plane is None else plane.split, _stream()),
            "efts_pack_rows")


def act_apply(act, z_ptr: int, resid_ptr: Optional[int], rowmask_ptr: Optional[int], y: Optional[F32Rows], plane: Optional[Plane],
              rows: int, c: int, drop_p: float = 0.0, drop_seed: int = 0) -> None:
    """y = (resid + Dropout(f(z))) * rowmask for a torch.nn activation outside the contraction epilogues (csrc/efts_act.hip);
    act = (EFTS_ACTFN id, p0, p1)"""
    L.check(L.load().efts_act_apply(z_ptr, resid_ptr, rowmask_ptr, act[0], act[1], act[2], None if y is None else y.ptr,
                                    None if plane is None else plane.ptr, 0 if plane is None else plane.ld, 1 if plane is None else plane.split,
                                    rows, c, drop_p, drop_seed & 0xFFFFFFFF, _stream()), "efts_act_apply")


def act_grad(act, g_ptr: int, z_ptr: int, rowmask_ptr: Optional[int], dz: Optional[F32Rows], plane: Optional[Plane], dbias, rows: int, c: int,
             drop_p: float = 0.0, drop_seed: int = 0) -> None:
    """dZ = G * rowmask * Dropout'(.) * f'(z), bias gradient += column sums"""
    L.check(L.load().efts_act_grad(g_ptr, z_ptr, rowmask_ptr, act[0], act[1], act[2], None if dz is None else dz.ptr,
                                   None if plane is None else plane.ptr, 0 if plane is None else plane.ld, 1 if plane is None else plane.split,
                                   _p(dbias), rows, c, drop_p, drop_seed & 0xFFFFFFFF, _stream()), "efts_act_grad")


def attn_soft_index(scores, ld, tl, ml, soft_idx, alpha_out, B, T1, T2) -> None:
    L.check(L.load().efts_attn_soft_index(scores.data_ptr(), ld, tl.data_ptr(), ml.data_ptr(), soft_idx.data_ptr(),
                                          _p(alpha_out), B, T1, T2, _stream()), "efts_attn_soft_index")


def imv_scan(soft_idx, tl, ml, imv, B, T2) -> None:
    L.check(L.load().efts_imv_scan(soft_idx.data_ptr(), tl.data_ptr(), ml.data_ptr(), imv.data_ptr(), B, T2, _stream()),
            "efts_imv_scan")


def aligned_positions(imv, tl, ml, sigma_e, offset, e, lde, B, T1, T2) -> None:
    L.check(L.load().efts_aligned_positions(imv.data_ptr(), tl.data_ptr(), ml.data_ptr(), sigma_e, offset,
                                            e.data_ptr(), _p(lde), B, T1, T2, _stream()), "efts_aligned_positions")


def duration_target(e, tl, ml, offset, method1: bool, lde, B, T1) -> None:
    L.check(L.load().efts_duration_target(e.data_ptr(), tl.data_ptr(), ml.data_ptr(), offset, int(method1), lde.data_ptr(), B, T1, _stream()),
            "efts_duration_target")


def reconst_alpha(e, tl, ml, sigma, alpha_out, plane: Optional[Plane], B, T1, T2, T2p) -> None:
    L.check(L.load().efts_reconst_alpha(e.data_ptr(), _p(tl), _p(ml), sigma, _p(alpha_out),
                                        None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                        B, T1, T2, T2p, _stream()), "efts_reconst_alpha")


def imv_align_fits(T1: int, T2: int) -> bool:
    """efts_imv_align keeps 2 * roundup(T2, 4) + kper + 1 floats in LDS (kper <= T1 keys per workgroup): a sufficient condition for its
    160 KiB limit, so that the callers take the three-kernel chain instead of an EFTS_ESHAPE at the boundary"""
    return (2 * roundup(T2, 4) + T1 + 1) * 4 <= 160 * 1024


def imv_align(soft_idx, tl, ml, sigma_e, offset, method1: bool, imv, e, lde, B, T1, T2) -> None:
    """imv_scan + aligned_positions + duration_target in one launch (efts_imv_align)"""
    L.check(L.load().efts_imv_align(soft_idx.data_ptr(), tl.data_ptr(), ml.data_ptr(), sigma_e, offset, int(method1), imv.data_ptr(),
                                    e.data_ptr(), _p(lde), B, T1, T2, _stream()), "efts_imv_align")


def expand(*, e, tl, ml, sigma: float, v: F32Rows, rs1: Rows, rs2: Rows, alpha_out: Optional[torch.Tensor] = None,
           y_f32: Optional[F32Rows] = None, y: Optional[Plane] = None, y_lo: Optional[Plane] = None) -> None:
    """alpha' generated in registers and contracted with V (efts_expand): H[b * T2p + j] = sum_i alpha'[b, i, j] V[b * T1p + i]"""
    g = L.ExpandArgs()
    g.e, g.text_len, g.mel_len, g.sigma = e.data_ptr(), _p(tl), _p(ml), sigma
    g.v, g.ldv = v.ptr, v.c
    g.B, g.T1, g.T1p, g.T2, g.T2p, g.n = rs1.B, rs1.T, rs1.Tp, rs2.T, rs2.Tp, v.c
    g.alpha_out = _p(alpha_out)
    if y_f32 is not None:
        g.y_f32, g.ldo = y_f32.ptr, y_f32.c
    if y is not None:
        g.y, g.ldy, g.y_split = y.ptr, y.ld, y.split
        g.y_lo = None if y_lo is None else y_lo.ptr
    L.check(L.load().efts_expand(C.byref(g), _stream()), "efts_expand")


def pack_vt(v: F32Rows, plane: Plane, B, T1, T1p, c) -> None:
    L.check(L.load().efts_pack_vt(v.ptr, v.c, plane.ptr, plane.ld, B, T1, T1p, c, _stream()), "efts_pack_vt")


def cumsum_rows(x, y, B, T) -> None:
    L.check(L.load().efts_cumsum_rows(x.data_ptr(), y.data_ptr(), B, T, _stream()), "efts_cumsum_rows")


def duration_positions(dur: torch.Tensor, ld: int, tl, force_delta, method1: bool, e, ml, B, T1) -> None:
    """cumsum of the durations -> positions e, mel lengths ml = round(e[len - 1]) (efts_duration_positions)"""
    L.check(L.load().efts_duration_positions(dur.data_ptr(), ld, tl.data_ptr(), -1.0 if force_delta is None else float(force_delta), int(method1),
                                             e.data_ptr(), ml.data_ptr(), B, T1, _stream()), "efts_duration_positions")


def layernorm_rows(x_ptr, gamma, beta, eps, rowmask_ptr, out_f32_ptr, plane: Optional[Plane], rows, c, drop_p: float = 0.0,
                   drop_seed: int = 0, seed_add_ptr: Optional[int] = None) -> None:
    L.check(L.load().efts_layernorm_rows(x_ptr, gamma.data_ptr(), beta.data_ptr(), eps, rowmask_ptr, out_f32_ptr,
                                         None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                         rows, c, 1 if plane is None else plane.split, drop_p, drop_seed & 0xFFFFFFFF, seed_add_ptr, _stream()),
            "efts_layernorm_rows")


def layernorm_dot(x_ptr, gamma, beta, eps, w, b, rowmask_ptr, mode, offset, out, rows, c, drop_p: float = 0.0,
                  drop_seed: int = 0, seed_add_ptr: Optional[int] = None) -> None:
    L.check(L.load().efts_layernorm_dot(x_ptr, gamma.data_ptr(), beta.data_ptr(), eps, w.data_ptr(), b.data_ptr(),
                                        rowmask_ptr, mode, offset, out.data_ptr(), rows, c, drop_p, drop_seed & 0xFFFFFFFF, seed_add_ptr, _stream()),
            "efts_layernorm_dot")


def store_words(dst: torch.Tensor, words) -> None:
    """up to 8 32-bit words, passed by value, into device memory in stream order (efts_store_words)"""
    arr = (C.c_uint32 * len(words))(*[int(w) & 0xFFFFFFFF for w in words])
    L.check(L.load().efts_store_words(dst.data_ptr(), arr, len(words), _stream()), "efts_store_words")


def mask_rows(x_ptr, rowmask_ptr, out: Optional[F32Rows], plane: Optional[Plane], rows, c) -> None:
    """out[row] = x[row] * rowmask[row] (fp32 and/or operand plane): efts_act_bwd in identity mode"""
    L.check(L.load().efts_act_bwd(x_ptr, None, None, rowmask_ptr, 0.0, 0, None if out is None else out.ptr,
                                  None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                  1 if plane is None else plane.split, None, rows, c, _stream()), "efts_act_bwd")


def losses_workspace(device) -> torch.Tensor:
    return torch.zeros(L.load().efts_losses_workspace_bytes() // 4, dtype=torch.float32, device=device)


def masked_losses(mel_pred_ptr, ldm, speech, ml, dur_pred, lde, tl, out3, ws, B, T1, T1p, T2, T2p, odim) -> None:
    L.check(L.load().efts_masked_losses(mel_pred_ptr, ldm, speech.data_ptr(), ml.data_ptr(), dur_pred.data_ptr(),
                                        lde.data_ptr(), tl.data_ptr(), out3.data_ptr(), ws.data_ptr(),
                                        B, T1, T1p, T2, T2p, odim, _stream()), "efts_masked_losses")


def losses_from_parts(part, n_part, ml, dur_pred, lde, tl, out3, B, T1, T1p, T2, odim) -> None:
    L.check(L.load().efts_losses_from_parts(part.data_ptr(), n_part, ml.data_ptr(), dur_pred.data_ptr(), lde.data_ptr(), tl.data_ptr(),
                                            out3.data_ptr(), B, T1, T1p, T2, odim, _stream()), "efts_losses_from_parts")


INV_SQRT = lambda d: 1.0 / math.sqrt(float(d))  # noqa: E731
