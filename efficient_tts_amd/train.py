"""Training step of EfficientTTSCNN on MI355X: forward with saved activations + hand-written
backward, all through the C ABI (reference: nntts/trainers/efficient_tts_trainer.py:139-160 calling
torch autograd on nntts/models/efficient_tts.py:120-228).

`TrainEngine.forward_backward()` returns the three losses and fills ONE flat fp32 gradient buffer
(views per parameter, laid out in backward-completion order so data-parallel buckets are contiguous
ranges that become final early: mel head + decoder first, text encoder + embedding last).
"""
from __future__ import annotations

import contextlib
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops as O
from .ops import F32Rows, PackedWeight, Plane, Rows

_lib = L.load


def _ptr(t):
    return None if t is None else t.data_ptr()


# Test hooks and tuning constants of the training pass (module-level: bench.py --train-set NAME=INT, tests).  Round 6 removed the decided A/Bs
# (_PACK_SPLIT, _NARROW_TN, _EARLY_PACKS: always on) and did not keep its own (_WGRAD_EARLY, _TEXT_RIDERS, _HEAD_ORDER, _PACK_LATE,
# _WGRAD_GROUP_WGS_ME: all measured slower or equal, profiles/train_ab_r06.txt).  `switch_tag()` is what a captured step
# (step_graph.GraphedStep) is valid for.  Finished A/Bs of earlier rounds are no longer switches: the direct wgrad for taps 1 / 3
# (was _WGRAD_TN_SMALL) and the prenet through efts_frame_linear (was _FRAME_PRENET) are simply the code.
_WGRAD_TN_SPLITS = 8         # > 0: weight gradients on the direct (row-major, stream-K) kernel wherever its tiles fit; 0 = everything through the transposed
                             # planes + split-K efts_gemm (the path odd shapes always take: tests compare the two)
_SIGN_MIN_ROWS = 16384       # row spaces from here on: the forward convolutions of the stacks write the activation's sign words
                             # (efts_gemm `sign_mask`) and efts_act_bwd reads those instead of y and x in fp32 (14 -> 6 B per element);
                             # shorter ones (the text side) keep the narrow tiling, which does not write them.  0 = never
_BIAS_PARTS = 1              # direct-wgrad layers: bias gradient as per-row-block sums finished by the wgrad reduction (0: atomics in act_bwd; tests compare)
_RESCONV_FWD = 3             # residual stacks whose FORWARD runs on efts_resconv5 when their row space is long enough: bit 0 decoder,
                             # bit 1 mel encoder (A/B: bench.py --train-set _RESCONV_FWD=...)
_RESCONV_DGRAD = -1          # ... and whose DGRAD does (same bits as efts_gemm's); -1: decoder (bf16) / decoder + mel encoder (bf16x3), measured best:
                             # 3.72 -> 3.58-3.61 ms per B = 32 step with both switches (bf16), 6.96 -> 6.62-6.64 (bf16x3), tools/train_ab.sh
_WGRAD_WGS = 480             # split-K target: 480 workgroups per wgrad launch measured best (6.30 vs 6.60 ms/step at 640)
_WGRAD_STREAM = 3            # the mel-length weight gradients (mel head, decoder group, mel-encoder group, prenet) + their reductions on a third stream: nothing
                             # on the dgrad chain waits for them, and they fill what that chain leaves idle -- above all the backward of the alignment block,
                             # 0.25 ms of small latency-bound launches that use neither the matrix pipes nor the power budget.  Joined in front of a gradient
                             # bucket's hand-over (data parallel) or of the optimizer.  Bit 0: decoder group, bit 1: head, mel-encoder group, prenet (A/B)
_GV_ON_SIDE = 1              # eager steps: the dV branch of the alignment backward (pack_vt of dH + one batched product) on the text-side stream beside the d alpha' -> ... -> dK chain
_FUSE_ACT_BWD = 1            # stacks whose dgrad runs on efts_resconv5: the activation backward of layer l - 1 in the epilogue of layer l's dgrad launch
                             # (csrc/efts_resconv_bwd.hip) instead of an efts_act_bwd launch of its own (0: separate launches; tests compare)
_WGRAD_GROUP_WGS = 384       # workgroups of a grouped launch (0: two per CU).  Swept 256..512 on the graphed B = 32 step: 3.27-3.32 ms at 384 against 3.33-3.34 at 512,
                             # 3.28-3.34 at 256 (per-layer launches: 3.48-3.50); the launch is bound by the chip, not by its busiest CU


def switch_tag() -> tuple:
    """every hook above, by value: part of the tag of a captured training step"""
    return (_WGRAD_TN_SPLITS, _SIGN_MIN_ROWS, _BIAS_PARTS, _RESCONV_FWD, _RESCONV_DGRAD, _WGRAD_WGS,
            _WGRAD_GROUP_WGS, _FUSE_ACT_BWD, O.RC_KERNEL, _WGRAD_STREAM)


class _TPlane(Plane):
    """transposed operand plane [channels][K = time] for wgrad"""

    def __init__(self, c: int, kpad: int, split: int, device):
        super().__init__(O.roundup(c, 128) + 136, kpad, split, device)
        self.c, self.kpad = c, kpad


class _TPlaneStack:
    """`n` transposed planes back to back (one per conv tap)"""

    def __init__(self, c: int, kpad: int, split: int, n: int, device):
        nrows = O.roundup(c, 128) + 136
        self.split = split
        self.nchunk = (kpad + O.chunk_k(split) - 1) // O.chunk_k(split)
        self.ld = self.nchunk * 128
        self.plane_bytes = nrows * self.ld
        self.buf = torch.zeros(n, nrows, self.ld, dtype=torch.uint8, device=device)
        self.ptr = self.buf.data_ptr()


def grad_layout(model) -> List[Tuple[str, torch.nn.Parameter]]:
    """Parameters in backward-completion order (see module docstring)."""
    named = dict(model.named_parameters())
    order: List[str] = []

    def conv(prefix):
        return [n for n in named if n.startswith(prefix)]

    order += conv("mel_output_layer.")
    for i in reversed(range(len(model.decoder.layers))):
        order += conv(f"decoder.layers.{i}.")
    order += conv("duration_predictor.")
    order += conv("mel_query_fc.")                       # (use_mel_query_fc: its gradient is ready in front of the mel encoder's)
    for i in reversed(range(len(model.mel_encoder.layers))):
        order += conv(f"mel_encoder.layers.{i}.")
    order += conv("mel_prenet.")
    order += conv("text_encoder_value.") + conv("text_encoder_key.")
    for i in reversed(range(len(model.text_encoder.layers))):
        order += conv(f"text_encoder.layers.{i}.")
    order += conv("text_embedding_table.")
    assert sorted(order) == sorted(named), "grad layout must cover every parameter exactly once"
    return [(n, named[n]) for n in order]


class TrainEngine:
    def __init__(self, model):
        self.m = model
        self.layout = grad_layout(model)
        self.named = list(model.named_parameters())     # model.parameters() order (what autograd's Function receives)
        self.params = tuple(p for _, p in self.named)
        self.step_words = None                          # device uint32[4] {lr, 1 - b1^t, sqrt(1 - b2^t) as float bits, 2 * dropout step}: set while
                                                        # a step is captured / replayed as a hipGraph (step_graph.GraphedStep)
        self.step_params = None                         # tuple(model.parameters()) of the running step (autograd.py), saves re-walks
        self.dev = next(model.parameters()).device
        self.numel = sum(p.numel() for _, p in self.layout)
        pad = (-self.numel) % 4
        self.flat = torch.zeros(self.numel + pad, dtype=torch.float32, device=self.dev)
        self.g: Dict[str, torch.Tensor] = {}
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n, p in self.layout:
            self.g[n] = self.flat[off:off + p.numel()].view_as(p)
            self.offsets[n] = (off, off + p.numel())
            off += p.numel()
        # bucket boundaries (element offsets into `flat`): head+decoder | dur+mel side | text side
        self.bucket_ends = [self.offsets[[n for n, _ in self.layout if n.startswith("decoder.layers.0.")][-1]][1],
                            self.offsets[[n for n, _ in self.layout if n.startswith("mel_prenet.")][-1]][1],
                            self.numel]
        self.folded: Dict[str, torch.Tensor] = {}
        self.wt: Dict[str, PackedWeight] = {}
        self._sig = None
        self._ws_tag = ""                       # "s" while work is being enqueued on the side stream
        self.bucket_hook: Optional[Callable[[int], None]] = None
        self.join_reduce: Optional[Callable[[], None]] = None
        self.mark: Optional[Callable[[str], None]] = None      # timing probe of the data-parallel wrapper (bench only)
        self.bound = set()                      # who holds views of `flat` (EftsAdam, DistributedEFTS): see autograd.engine_of

    def stale(self, model) -> bool:
        """True when the model's parameter set no longer matches the layout captured at construction
        (remove_weight_norm / apply_weight_norm / .to(device))."""
        cur = {n: p for n, p in model.named_parameters()}
        return set(cur) != set(self.g) or any(cur[n].shape != self.g[n].shape or cur[n].device != self.dev for n in cur)

    # ------------------------------------------------------------------ weights for the backward
    @staticmethod
    def _pack_phase(name: str, has_t: bool):
        """where a weight's planes are repacked in the training step (model._weights(phase_of=...)): "te" = on the text stream (text encoder,
        duration predictor, key / value), "me" = on the main stream (everything else); forward and dgrad plane in one pass.  Round 6 measured a finer
        split -- only the planes the step's first launches read in front of them, the rest beside / behind the head of the step on the third stream:
        3.23 -> 3.25-3.27 ms (bf16), 5.93-5.97 -> 6.02-6.05 (bf16x3) under bench.py's clock (profiles/train_ab_r06.txt): not kept."""
        return [("te" if name.startswith(("text_encoder.", "dur.", "key", "value")) else "me", has_t, True)]

    def _prepare_weights(self):
        """forward planes, folded fp32 weights and transposed/flipped dgrad planes, all from `model._weights` (a handful
        of grouped launches).  The derived copies follow `model._packed_gen` / `_folded_gen`, the repack counters: the
        fused optimizer updates parameters in place without a version bump, so the signature of the parameters cannot
        be used to detect a change, and an eval forward in between repacks without writing the copies kept here."""
        m = self.m
        dev = self.dev
        for name, conv in m._conv_modules():
            if name not in self.wt:
                taps = conv.kernel_size[0]
                if hasattr(conv, "weight_g") and (conv.out_channels % 64 or conv.in_channels % 64 or taps > 5):
                    self.folded[name] = torch.empty(conv.out_channels, conv.in_channels, taps, device=dev)   # row-kernel shapes only
                self.wt[name] = PackedWeight(conv.in_channels, conv.out_channels, taps, m.split, dev)
        lins = [("key", m.text_encoder_key), ("head", m.mel_output_layer)]
        if not m.share_text_encoder_key_value:
            lins.append(("value", m.text_encoder_value))
        if m.mel_query_fc is not None:
            lins.append(("qfc", m.mel_query_fc))
        for name, lin in lins:
            if name not in self.wt:
                self.wt[name] = PackedWeight(lin.in_features, lin.out_features, 1, m.split, dev)
        pk = m._weights(self.folded, self.wt, self.step_params, phase_of=self._pack_phase)
        if m._folded_gen != m._packed_gen:
            m._packed_sig = None                # the last repack (an eval forward) did not write the copies kept here
            pk = m._weights(self.folded, self.wt, self.step_params, phase_of=self._pack_phase)
        return pk

    # ------------------------------------------------------------------ small wrappers
    def _act_bwd(self, g_ptr, y_ptr, x_ptr, mask_ptr, mode, dz: Optional[F32Rows], plane: Optional[Plane], dbias, rows, c,
                 drop_p: float = 0.0, drop_seed: int = 0):
        L.check(_lib().efts_act_bwd_dropout(g_ptr, y_ptr, x_ptr, mask_ptr, self.m.slope, mode, None if dz is None else dz.ptr,
                                            None if plane is None else plane.ptr, 0 if plane is None else plane.ld,
                                            1 if plane is None else plane.split, _ptr(dbias), rows, c, drop_p, drop_seed & 0xFFFFFFFF,
                                            O._stream()), "efts_act_bwd")

    def _conv_drop(self, k: int):
        """(p, seed) of the train-mode Dropout behind the activation of conv / prenet launch k (efts_modules.py:38-47,
        efficient_tts.py:76-80): one mask per launch and step, regenerated by the backward from the same seed"""
        m = self.m
        if not m.training or m.dropout_rate < 1e-5:
            return 0.0, 0
        return float(m.dropout_rate), m._dropout_seeds(self.drop_calls)[0](k)

    def _seed_base(self) -> int:
        """the model's base seed mixed with the data-parallel rank"""
        return self.m._dropout_base()

    def _wgrad_any(self, ws, dz_f_ptr, dz_p: Optional[Plane], x_f_ptr, x_p: Optional[Plane], cout, cin, taps, rows, out_dw, defer: Optional[list] = None):
        """un-normed weights: the direct kernel when both operand planes exist in one format and the shape fits its tiles.
        defer: a list that collects the direct items of equally shaped layers for ONE grouped launch (the caller passes it to
        _wgrad_group once the operand planes of all of them are final and still intact)"""
        if (_WGRAD_TN_SPLITS > 0 and dz_p is not None and x_p is not None and dz_p.split == x_p.split
                and cout % 128 == 0 and cin % 64 == 0 and taps in (1, 3, 5)):
            item = (dz_p, x_p, None, None, out_dw, None, None, None)
            if defer is not None:
                defer.append(item)
            else:
                self._wgrad_group(ws, [item], cout, cin, rows, taps, dz_p.split)
        else:
            self._wgrad(ws, dz_f_ptr, cout, x_f_ptr, cin, cin, taps, rows, None, None, out_dw, None)

    def _wgrad(self, ws, dz_ptr, cout, x_ptr, ldx, cin, taps, rows, v, g, out_dw, out_dg):
        """dW[co][ci][k] = sum_t dZ[t][co] X[t+k-pad][ci] as `taps` split-K GEMMs on transposed planes"""
        split = self.m.split
        ck = O.chunk_k(split)
        tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
        nch_total = (rows + ck - 1) // ck
        S = max(1, min(32, (_WGRAD_WGS + tiles * taps - 1) // (tiles * taps), nch_total))    # ~_WGRAD_WGS workgroups per launch
        nch = (nch_total + S - 1) // S
        kpad = O.roundup(S * nch * ck, 64)
        tag = self._ws_tag                       # the side stream owns its own scratch (both streams run wgrads at once)
        zt = ws.get(("zt", tag, cout, kpad, split), lambda: _TPlane(cout, kpad, split, self.dev))
        L.check(_lib().efts_pack_t(dz_ptr, cout, zt.ptr, zt.ld, 0, split, rows, cout, 0, 1, kpad, O._stream()), "efts_pack_t")
        part = ws.get(("part", tag, taps, S, cout, cin), lambda: torch.empty(taps, S, cout, cin, device=self.dev))
        pad = (taps - 1) // 2
        xts = ws.get(("xt", tag, cin, kpad, split, taps), lambda: _TPlaneStack(cin, kpad, split, taps, self.dev))
        L.check(_lib().efts_pack_t(x_ptr, ldx, xts.ptr, xts.ld, xts.plane_bytes, split, rows, cin, -pad, taps, kpad, O._stream()),
                "efts_pack_t")
        O.gemm(a=zt, b_ptr=xts.ptr, ldb=xts.ld, m=cout, n=cin, batch=S, nchunk=nch, a_batch_stride=nch * 128, b_batch_stride=nch * 128,
               out_f32_ptr=part.data_ptr(), ldo=cin, out_batch_stride=cout * cin, batch2=taps, b_batch2_stride=xts.plane_bytes,
               out_batch2_stride=S * cout * cin)
        L.check(_lib().efts_wgrad_reduce(part.data_ptr(), S, _ptr(v), _ptr(g), out_dw.data_ptr(), _ptr(out_dg), cout, cin, taps,
                                         O._stream()), "efts_wgrad_reduce")

    def _wgrad_group(self, ws, items, cout, cin, rows, taps: int, split: int, wgs: Optional[int] = None):
        """the direct weight gradients of several layers of one stack (same shape, same row space) as ONE stream-K launch and ONE
        reduction (csrc/efts_wgrad.hip `wgrad_sk_kernel`, csrc/efts_train.hip `wgrad_reduce_sk_kernel`).
        items: (dz_p, x_p, v, g, out_dw, out_dg, bias_part, dbias) per layer (more than the library takes per launch: several launches)"""
        if len(items) > L.WGRAD_MAX_ITEMS:
            for i in range(0, len(items), L.WGRAD_MAX_ITEMS):
                self._wgrad_group(ws, items[i:i + L.WGRAD_MAX_ITEMS], cout, cin, rows, taps, split, wgs)
            return
        lib = _lib()
        n = len(items)
        wgs = _WGRAD_GROUP_WGS if not wgs else wgs
        arr = (L.WgradItem * n)()
        for a, (dz_p, x_p, v, g, dw, dg, bp, db) in zip(arr, items):
            a.dz_plane, a.ldz, a.x_plane, a.ldx = dz_p.ptr, dz_p.ld, x_p.ptr, x_p.ld
            a.v, a.g, a.dw_or_dv, a.dg = _ptr(v), _ptr(g), dw.data_ptr(), _ptr(dg)
            a.bias_part, a.dbias, a.nparts = _ptr(bp), _ptr(db), 0 if bp is None else bp.shape[0]
        nbytes = lib.efts_wgrad_grouped_part_bytes(n, rows, cout, cin, taps, split, wgs)
        if nbytes < 0:
            L.check(-1, "efts_wgrad_grouped_part_bytes")
        part = ws.get(("gpart", self._ws_tag, n, rows, cout, cin, taps, split, wgs), lambda: torch.empty(nbytes // 4, device=self.dev))
        L.check(lib.efts_wgrad_tn_grouped(arr, n, part.data_ptr(), rows, cout, cin, taps, split, wgs, O._stream()), "efts_wgrad_tn_grouped")
        L.check(lib.efts_wgrad_reduce_grouped(arr, n, part.data_ptr(), rows, cout, cin, taps, split, wgs, O._stream()),
                "efts_wgrad_reduce_grouped")

    @staticmethod
    def _narrow_ok(dz_split: int, x_split: int, cout: int, cin: int, ldz: int, ldx: int) -> bool:
        """THE applicability test of _wgrad_narrow (callers that allocate differently for the two paths ask this, not a copy of it)"""
        return bool(_WGRAD_TN_SPLITS > 0 and dz_split == 1 and x_split == 1 and max(cout, cin) % 128 == 0 and min(cout, cin) <= 128
                    and ldz >= max(cout, 128) * 2 and ldx >= max(cin, 128) * 2)

    def _wgrad_narrow(self, ws, tag, dz_p: Plane, x_p: Plane, cout, cin, rows, out_dw) -> bool:
        """weight gradient of a Linear with an 80-channel side (mel head 512 -> 80, prenet 80 -> 512) on the direct kernel: bf16 planes are
        128 columns wide (two 64-channel chunks, zeros beyond the 80th), so the contraction runs on the padded 128 and the result's first
        80 rows / columns are copied out -- instead of two transposed operand copies (efts_pack_t over the mel-length stream) + split-K
        efts_gemm + reduction.  False: not applicable (bf16x3 planes are 96 wide), the caller takes the transposed-plane path."""
        if not self._narrow_ok(dz_p.split, x_p.split, cout, cin, dz_p.ld, x_p.ld):
            return False
        co_p, ci_p = max(cout, 128), max(cin, 128)
        scratch = ws.tensor(f"B{tag}_dw_pad", (co_p, ci_p))
        self._wgrad_group(ws, [(dz_p, x_p, None, None, scratch, None, None, None)], co_p, ci_p, rows, 1, 1)     # 8 tiles, stream-K over the rows
        out_dw.copy_(scratch[:cout, :cin])
        return True

    # ------------------------------------------------------------------ forward with saved activations
    def _stack_fwd(self, ws, tag, blk, pk, rs, x_f, x_p, gap_ptr, last_split):
        m, C = self.m, self.m.n_channels
        saved = []
        layers = getattr(m, blk).layers
        if (_RESCONV_FWD & dict(dec=1, me=2, te=0)[tag]) and m._on_resconv(rs) and self._conv_drop(0)[0] == 0.0 and _WGRAD_TN_SPLITS > 0 and C % 128 == 0:
            # mel-length stack on efts_resconv5 (the inference kernel): the stream between the layers is hi + lo bf16 planes (every
            # hi plane is kept: it is the layer's operand in the wgrad), the activation's sign leaves the epilogue as bit rows
            # (efts_act_bwd mode 5); fp32 only into the first layer (the producer's stream) and out of the last one
            n = len(layers)
            x_lo = None
            for i, layer in enumerate(layers):
                last = i == n - 1
                o_split = last_split if last else m.split
                o_p = ws.plane(f"T{tag}_p{i}", rs, C, o_split)
                o_l = ws.plane(f"T{tag}_l{i & 1}", rs, C, 1) if (o_split == 1 and not last) else None
                o_f = ws.f32(f"T{tag}_f{n - 1}", rs, C) if last else None
                sg = ws.tensor(f"T{tag}_sb{i}", (rs.rows, C // 8), torch.uint8)
                O.resconv5(x=x_p, x_lo=x_lo, x_f32_ptr=x_f.ptr if i == 0 else None, ldr=C, w=pk[f"{blk}.{i}"], taps=m.k_size, m=rs.rows, n=C,
                           bias=layer.conv[0].bias, slope=m.slope, rowmask_ptr=gap_ptr, y_f32_ptr=None if o_f is None else o_f.ptr, ldo=C,
                           y=o_p, y_lo=o_l, sign_bits_ptr=sg.data_ptr())
                saved.append((None, None, x_p, (sg, 5), 0.0, 0))
                x_p, x_lo = o_p, o_l
            return o_f, x_p, saved
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            w = pk[f"{blk}.{i}"]
            o_f = ws.f32(f"T{tag}_f{i}", rs, C)
            o_p = ws.plane(f"T{tag}_p{i}", rs, C, last_split if last else m.split)
            dp, dseed = self._conv_drop(dict(te=10, me=20, dec=30)[tag] + i)
            if m.act_general is not None:
                # a torch.nn activation outside the contraction epilogues (csrc/efts_act.hip): the pre-activation is kept in fp32 for f'(z)
                z = ws.f32(f"T{tag}_z{i}", rs, C)
                O.gemm(a=x_p, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=m.k_size, m=rs.rows, n=C, bias=layer.conv[0].bias,
                       out_f32_ptr=z.ptr, ldo=C)
                O.act_apply(m.act_general, z.ptr, x_f.ptr, gap_ptr, o_f, o_p, rs.rows, C, dp, dseed)
                saved.append((x_f, o_f, x_p, (z, "z"), dp, dseed))
                x_f, x_p = o_f, o_p
                continue
            # under Dropout y - x no longer carries the activation's sign where the element was dropped: always the sign words then
            sg = ws.tensor(f"T{tag}_sg{i}", (rs.rows, C // 8), torch.uint8) if ((0 < _SIGN_MIN_ROWS <= rs.rows or dp > 0) and C % 128 == 0) else None
            O.gemm(a=x_p, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=m.k_size, m=rs.rows, n=C, act=L.ACT_LEAKY, slope=m.slope,
                   bias=layer.conv[0].bias, resid_ptr=x_f.ptr, ldr=C, rowmask_ptr=gap_ptr, out_f32_ptr=o_f.ptr, ldo=C, out_plane=o_p,
                   sign_mask_ptr=None if sg is None else sg.data_ptr(), drop_p=dp, drop_seed=dseed)
            saved.append((x_f, o_f, x_p, None if sg is None else (sg, 4), dp, dseed))
            x_f, x_p = o_f, o_p
        return x_f, x_p, saved

    @contextlib.contextmanager
    def _forked(self, st):
        """launches of the body go to stream `st`, ordered behind everything already enqueued on the current stream (None: no-op)"""
        if st is None:
            yield
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        st.wait_event(ev)
        keep_tag = self._ws_tag
        with O.on_stream(st):
            self._ws_tag = "w"                  # (scratch of its own: the other streams run weight gradients at the same time)
            try:
                yield
            finally:
                self._ws_tag = keep_tag

    def _stack_bwd(self, ws, tag, blk, rs, G: F32Rows, saved, gap_ptr, final_mask_ptr, final_plane: Optional[Plane], wgrad_stream=None):
        """backward through n x (x + leaky(conv(x))); returns the gradient w.r.t. the stack input.
        wgrad_stream: the stack's grouped weight gradients (and their reduction) are enqueued there, behind an event of the current stream
        (their operands -- every layer's dZ plane and input plane -- are final then and not written again in this step); the caller joins"""
        m, C = self.m, self.m.n_channels
        layers = getattr(m, blk).layers
        group = []                                               # (grouped direct wgrads: every layer keeps its dZ plane and bias sums until the stack is through)
        on_rc = bool((_RESCONV_DGRAD if _RESCONV_DGRAD >= 0 else (1 if m.split == 1 else 3)) & dict(dec=1, me=2, te=0)[tag]) and m._on_resconv(rs)
        fused = None                                             # (dZ plane, bias sums) of layer i the dgrad launch of layer i + 1 has already written
        for i in reversed(range(len(layers))):
            x_f, y_f, x_pl, sg, dp, dseed = saved[i]
            conv = layers[i].conv[0]
            pre = f"{blk}.layers.{i}.conv.0."
            direct = _WGRAD_TN_SPLITS > 0 and C % 128 == 0 and x_pl.split == m.split and m.k_size <= 5     # (efts_wgrad_tn_grouped: taps 1 / 3 / 5)
            keep = str(i) if direct else ""
            if fused is not None:
                dz_p, bp = fused
                dz_f = None
            else:
                dz_p = ws.plane(f"B{tag}_dzp{keep}", rs, C, m.split)
                # the direct wgrad and the dgrad both read dZ as the bf16 plane: its fp32 copy is only written for the
                # transposed-plane path
                dz_f = None if direct else ws.f32(f"B{tag}_dz", rs, C)
                # direct path: the bias gradient leaves act_bwd as per-row-block sums and is finished by the wgrad reduction
                # (no same-address atomics: ~8 of 22 us per launch at mel length)
                bp = ws.tensor(f"B{tag}_bp{keep}", ((rs.rows + 63) // 64, C)) if (direct and _BIAS_PARTS) else None
                db, parts = (bp, L.ACT_BWD_BIAS_PARTS) if bp is not None else (self.g[pre + "bias"], 0)
                if sg is not None and sg[1] == "z":                  # general activation: f'(z) from the kept pre-activation, bias gradient by atomics
                    bp = None
                    O.act_grad(m.act_general, G.ptr, sg[0].ptr, gap_ptr, dz_f, dz_p, self.g[pre + "bias"], rs.rows, C, dp, dseed)
                elif sg is not None:                                 # (sign words of efts_gemm: mode 4; sign bits of efts_resconv5: mode 5)
                    self._act_bwd(G.ptr, sg[0].data_ptr(), None, gap_ptr, sg[1] | parts, dz_f, dz_p, db, rs.rows, C, dp, dseed)
                else:
                    self._act_bwd(G.ptr, y_f.ptr, x_f.ptr, gap_ptr, 1 | parts, dz_f, dz_p, db, rs.rows, C, dp, dseed)
            wn = hasattr(conv, "weight_g")
            v_, g_ = (conv.weight_v.detach(), conv.weight_g.detach()) if wn else (None, None)
            dw_, dg_ = (self.g[pre + "weight_v"], self.g[pre + "weight_g"]) if wn else (self.g[pre + "weight"], None)
            if direct:
                group.append((dz_p, x_pl, v_, g_, dw_, dg_, bp, self.g[pre + "bias"]))
            else:
                self._wgrad(ws, dz_f.ptr, C, x_f.ptr, C, C, m.k_size, rs.rows, v_, g_, dw_, dg_)
            wt = self.wt[f"{blk}.{i}"]
            Gn = ws.f32(f"B{tag}_G{i & 1}", rs, C)
            last = i == 0
            fused = None
            if on_rc and dz_p.split == m.split:
                # dgrad on the persistent kernel: G' = (G + conv_T(dZ)) * mask = a residual layer with the transposed weights, no bias
                # and slope 1, fp32 gradient stream in and out (bit-identical to the efts_gemm launch)
                below = saved[i - 1] if i > 0 else None
                if (_FUSE_ACT_BWD and below is not None and below[3] is not None and below[3][1] == 5 and below[4] == 0.0 and direct
                        and _BIAS_PARTS and m.k_size == 5 and below[2].split == m.split):
                    # ... and the activation backward of layer i - 1 on G' while the epilogue holds it: dZ_{i-1} as its operand plane and one
                    # row of column sums per tile (the launch efts_act_bwd would otherwise read G' back for)
                    nz_p = ws.plane(f"B{tag}_dzp{i - 1}", rs, C, m.split)
                    nbp = ws.tensor(f"B{tag}_bq{i - 1}", (O.resconv5_bias_rows(rs.rows, C), C))
                    O.resconv5(x=dz_p, x_f32_ptr=G.ptr, ldr=C, w=wt, taps=5, m=rs.rows, n=C, slope=1.0, rowmask_ptr=gap_ptr, y_f32_ptr=Gn.ptr, ldo=C,
                               y=nz_p, act_bwd_sign_ptr=below[3][0].data_ptr(), act_bwd_slope=m.slope, act_bwd_bias_part=nbp)
                    fused = (nz_p, nbp)
                else:
                    O.resconv5(x=dz_p, x_f32_ptr=G.ptr, ldr=C, w=wt, taps=m.k_size, m=rs.rows, n=C, slope=1.0,
                               rowmask_ptr=final_mask_ptr if last else gap_ptr, y_f32_ptr=Gn.ptr, ldo=C, y=final_plane if last else None)
            else:
                O.gemm(a=dz_p, b_ptr=wt.ptr, ldb=wt.ld, b_tap_stride=wt.tap_stride, taps=m.k_size, m=rs.rows, n=C, resid_ptr=G.ptr, ldr=C,
                       rowmask_ptr=final_mask_ptr if last else gap_ptr, out_f32_ptr=Gn.ptr, ldo=C,
                       out_plane=final_plane if last else None)
            G = Gn
        if group:
            with self._forked(wgrad_stream):
                self._wgrad_group(ws, group, C, C, rs.rows, m.k_size, m.split)
        return G

    def forward_backward(self, text, text_lengths, speech, speech_lengths, gscale: Optional[torch.Tensor] = None,
                         keep: bool = False):
        """One fwd+bwd.  Returns (out3 = [loss, mel_loss, dur_loss] device tensor, aux dict)."""
        with O.stream_scope():
            return self._forward_backward(text, text_lengths, speech, speech_lengths, gscale, keep)

    def _forward_backward(self, text, text_lengths, speech, speech_lengths, gscale, keep):
        m = self.m
        dev = self.dev
        L.require_device()
        B, T1 = text.shape
        T2 = speech.shape[1]
        C, odim, split = m.n_channels, m.odim, m.split
        text = text.contiguous()
        speech = speech.contiguous().float()
        ws = m._workspace(("train", B, T1, T2), dev)
        rs1, rs2 = Rows(B, T1, m.row_gap), Rows(B, T2, m.row_gap)
        gap1, len1 = ws.tensor("gap1", (rs1.rows,)), ws.tensor("len1", (rs1.rows,))
        gap2, len2 = ws.tensor("gap2", (rs2.rows,)), ws.tensor("len2", (rs2.rows,))
        tl_, ml_ = text_lengths.to(dev), speech_lengths.to(dev)
        if tl_.dtype == ml_.dtype and tl_.dtype in (torch.int64, torch.int32) and tl_.is_contiguous() and ml_.is_contiguous():
            tl, ml = O.row_masks_pair(tl_, ml_, rs1, rs2, gap1, len1, gap2, len2)       # both masks + the int32 lengths in one launch
        else:
            tl, ml = tl_.to(torch.int32), ml_.to(torch.int32)
            O.row_masks(tl, rs1, gap1, len1)
            O.row_masks(ml, rs2, gap2, len2)
        # the repack of the operand planes (weight-norm fold + bf16 planes + dgrad planes, ~110 us on one stream): the text-side planes on
        # the stream the text side runs on, the mel-side planes here -- both behind the masks and the previous step's optimizer
        side0 = m._side_stream(dev)
        side0.wait_stream(torch.cuda.current_stream(dev))
        pk = self._prepare_weights()
        m._issue_packs("me")
        with O.on_stream(side0):
            m._issue_packs("te")
        self.flat.zero_()

        # Two HIP streams.  The text-length work (embedding, text encoder, K/V, duration predictor and all of their
        # backward) runs on ~B*T1 = 4k rows: launches of 70-270 workgroups that leave most of the chip idle.  None of it
        # depends on the mel-length work except through K/V (forward) and dK/dV/d(dur) (backward), so it goes to a side
        # stream and fills the idle CUs / tail rounds of the mel-length kernels; events mark the few hand-over points.
        main = torch.cuda.current_stream(dev)
        side = m._side_stream(dev)
        if self.mark is not None:
            self.mark("step_start")
        dp = m.duration_predictor
        ln0, ln1 = dp.conv[0][2], dp.conv[1][2]
        # Dropout(0.1) of the duration predictor is active in train() mode like the reference's
        # (duration_predictor.py:61; the model never forwards its own dropout_rate to it)
        drop_p = float(dp.conv[0][3].p) if m.training else 0.0
        # one mask family per (base seed, data-parallel rank, step counter): replicas draw different masks (the reference's ranks
        # have their own torch RNG streams); the trainer sets the counter to the step count when it loads a checkpoint, so that
        # --resume continues the sequence instead of replaying it
        m.dropout_calls = int(getattr(m, "dropout_calls", 0)) + 1
        self.drop_calls = m.dropout_calls
        self.seed_base = self._seed_base()
        if self.step_words is not None:
            # a step that is being captured (step_graph.GraphedStep): the counter's contribution comes from device word 3 of
            # `step_words` (= 2 * dropout_calls, refreshed in front of every replay), the by-value seeds stay constant
            sadd = self.step_words.data_ptr() + 12
            seed0, seed1 = self.seed_base & 0xFFFFFFFF, (self.seed_base + 1) & 0xFFFFFFFF
        else:
            sadd = None
            seed0, seed1 = (self.seed_base + 2 * self.drop_calls) & 0xFFFFFFFF, (self.seed_base + 2 * self.drop_calls + 1) & 0xFFFFFFFF

        # ============================ forward (efficient_tts.py:144-227), activations kept
        with O.on_stream(side):
            self._ws_tag = "s"
            emb_f, emb_p = ws.f32("Temb_f", rs1, C), ws.plane("Temb_p", rs1, C, split)
            O.embed(text, m.text_embedding_table.weight.detach(), emb_f, emb_p, rs1)
            te_f, te_p, te_saved = self._stack_fwd(ws, "te", "text_encoder", pk, rs1, emb_f, emb_p, gap1.data_ptr(), split)
            key_f, key_p = ws.f32("Tkey_f", rs1, C), ws.plane("Tkey_p", rs1, C, 2)
            val_f, val_p = ws.f32("Tval_f", rs1, C), ws.plane("Tval_p", rs1, C, split)
            shared = m.share_text_encoder_key_value                 # efficient_tts.py:150-153: the value is the key projection
            wk = pk["key"]
            wv = wk if shared else pk["value"]
            O.gemm(a=te_p, b_ptr=wk.ptr, ldb=wk.ld, m=rs1.rows, n=C, bias=m.text_encoder_key.bias, rowmask_ptr=len1.data_ptr(),
                   out_f32_ptr=key_f.ptr, ldo=C, out_plane=key_p)
            O.gemm(a=te_p, b_ptr=wv.ptr, ldb=wv.ld, m=rs1.rows, n=C, bias=(m.text_encoder_key if shared else m.text_encoder_value).bias,
                   rowmask_ptr=len1.data_ptr(), out_f32_ptr=val_f.ptr, ldo=C, out_plane=val_p)
            ev_kv = torch.cuda.Event()
            ev_kv.record(side)
            # duration predictor (efficient_tts.py:219): needs V only
            h1_f, l1_f, l1_p = ws.f32("Tdur_h1", rs1, C), ws.f32("Tdur_l1", rs1, C), ws.plane("Tdur_l1p", rs1, C, split)
            h2_f = ws.f32("Tdur_h2", rs1, C)
            dur = ws.tensor("Tdur_out", (rs1.rows,))
            w0, w1 = pk["dur.0"], pk["dur.1"]
            O.gemm(a=val_p, b_ptr=w0.ptr, ldb=w0.ld, b_tap_stride=w0.tap_stride, taps=3, m=rs1.rows, n=C, act=L.ACT_RELU,
                   bias=dp.conv[0][0].bias, out_f32_ptr=h1_f.ptr, ldo=C)
            O.layernorm_rows(h1_f.ptr, ln0.weight.detach(), ln0.bias.detach(), ln0.eps, gap1.data_ptr(), l1_f.ptr, l1_p, rs1.rows, C,
                             drop_p, seed0, sadd)
            O.gemm(a=l1_p, b_ptr=w1.ptr, ldb=w1.ld, b_tap_stride=w1.tap_stride, taps=3, m=rs1.rows, n=C, act=L.ACT_RELU,
                   bias=dp.conv[1][0].bias, out_f32_ptr=h2_f.ptr, ldo=C)
            O.layernorm_dot(h2_f.ptr, ln1.weight.detach(), ln1.bias.detach(), ln1.eps, dp.linear.weight.detach(),
                            dp.linear.bias.detach(), len1.data_ptr(), 0, float(dp.offset), dur, rs1.rows, C, drop_p, seed1, sadd)
            ev_dur = torch.cuda.Event()
            ev_dur.record(side)
            self._ws_tag = ""

        mel_in_f, mel_in = ws.f32("Tmel_in_f", rs2, odim), ws.plane("Tmel_in", rs2, odim, split)
        O.pack_rows(speech, mel_in_f, mel_in, rs2)                  # (the backward's wgrad of the prenet reads both)
        pre_f, pre_p = ws.f32("Tpre_f", rs2, C), ws.plane("Tpre_p", rs2, C, split)
        wp = pk["prenet"]
        pre_dp, pre_seed = self._conv_drop(40)                       # mel_prenet's Dropout (efficient_tts.py:76-80)
        pre_z = None
        if m.act_general is not None:
            pre_z = ws.f32("Tpre_z", rs2, C)
            O.gemm(a=mel_in, b_ptr=wp.ptr, ldb=wp.ld, m=rs2.rows, n=C, bias=m.mel_prenet[0].bias, out_f32_ptr=pre_z.ptr, ldo=C)
            O.act_apply(m.act_general, pre_z.ptr, None, gap2.data_ptr(), pre_f, pre_p, rs2.rows, C, pre_dp, pre_seed)
        elif pre_dp == 0.0 and m.fuse_prenet and odim % 8 == 0 and odim <= 128 and C % 128 == 0:
            # no Dropout on the prenet (the shipped recipe): straight from the caller's frames, whole-line stores (efts_frame_linear;
            # bit-identical to the launch below)
            O.frame_linear(x=speech, w=wp, bias=m.mel_prenet[0].bias, act=L.ACT_LEAKY, slope=m.slope, rs=rs2, y=pre_p, y_f32=pre_f)
        else:
            O.gemm(a=mel_in, b_ptr=wp.ptr, ldb=wp.ld, m=rs2.rows, n=C, act=L.ACT_LEAKY, slope=m.slope, bias=m.mel_prenet[0].bias,
                   rowmask_ptr=gap2.data_ptr(), out_f32_ptr=pre_f.ptr, ldo=C, out_plane=pre_p, drop_p=pre_dp, drop_seed=pre_seed)
        if m.mel_query_fc is None:
            q_f, q_p, me_saved = self._stack_fwd(ws, "me", "mel_encoder", pk, rs2, pre_f, pre_p, gap2.data_ptr(), 2)
        else:                                                       # efficient_tts.py:163-164: Linear(C, C) in front of the attention
            mh_f, mh_p, me_saved = self._stack_fwd(ws, "me", "mel_encoder", pk, rs2, pre_f, pre_p, gap2.data_ptr(), split)
            q_f, q_p = ws.f32("Tq_f", rs2, C), ws.plane("Tq_p", rs2, C, 2)
            wq = pk["qfc"]
            O.gemm(a=mh_p, b_ptr=wq.ptr, ldb=wq.ld, m=rs2.rows, n=C, bias=m.mel_query_fc.bias, rowmask_ptr=gap2.data_ptr(),
                   out_f32_ptr=q_f.ptr, ldo=C, out_plane=q_p)
        if self.mark is not None:
            self.mark("fwd_mel_encoder_done")

        main.wait_event(ev_kv)                                      # K, V from the side stream
        scale = O.INV_SQRT(C)
        scores = ws.tensor("Tscores", (B, T2, T1))
        O.gemm(a=q_p, b_ptr=key_p.ptr, ldb=key_p.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q_p.ld,
               b_batch_stride=rs1.Tp * key_p.ld, alpha=scale, out_f32_ptr=scores.data_ptr(), ldo=T1, out_batch_stride=T2 * T1)
        sidx, imv = ws.tensor("Tsidx", (B, T2)), ws.tensor("Timv", (B, T2))
        O.attn_soft_index(scores, T1, tl, ml, sidx, None, B, T1, T2)
        e, lde = ws.tensor("Te", (B, T1)), ws.tensor("Tlde", (B, T1))
        if m.fuse_align and O.imv_align_fits(T1, T2):
            O.imv_align(sidx, tl, ml, float(m.sigma_e), float(m.duration_offset), m.delta_e_method_1, imv, e, lde, B, T1, T2)
        else:
            O.imv_scan(sidx, tl, ml, imv, B, T2)
            O.aligned_positions(imv, tl, ml, float(m.sigma_e), float(m.duration_offset), e, lde if m.delta_e_method_1 else None, B, T1, T2)
            if not m.delta_e_method_1:                               # efficient_tts.py:205-213 (the target is detached either way)
                O.duration_target(e, tl, ml, float(m.duration_offset), False, lde, B, T1)
        ralpha = ws.tensor("Tralpha", (B, T1, T2))
        h_f, h_p = ws.f32("Texp_f", rs2, C), ws.plane("Texp_p", rs2, C, split)
        if m._fused_expand(T1):
            # alpha' produced in registers inside the expand contraction (efts_expand); the backward packs its own operands from
            # the fp32 alpha' kept here
            if self.mark is not None:
                self.mark("fwd_alignment_done")
            O.expand(e=e, tl=tl, ml=ml, sigma=float(m.sigma), v=val_f, rs1=rs1, rs2=rs2, alpha_out=ralpha, y_f32=h_f, y=h_p)
        else:
            ra_p = ws.plane("Tra_p", rs2, T1, 2)
            O.reconst_alpha(e, tl, ml, float(m.sigma), ralpha, ra_p, B, T1, T2, rs2.Tp)
            if self.mark is not None:
                self.mark("fwd_alignment_done")
            vt = ws.raw_plane("Tvt", B * C + 136, T1, 2)
            O.pack_vt(val_f, vt, B, T1, rs1.Tp, C)
            O.gemm(a=ra_p, b_ptr=vt.ptr, ldb=vt.ld, m=T2, n=C, batch=B, a_batch_stride=rs2.Tp * ra_p.ld, b_batch_stride=C * vt.ld,
                   rowmask_ptr=len2.data_ptr(), rowmask_batch_stride=rs2.Tp, out_f32_ptr=h_f.ptr, ldo=C, out_batch_stride=rs2.Tp * C,
                   out_plane=h_p, outb_batch_stride=rs2.Tp * h_p.ld)
        d_f, d_p, dec_saved = self._stack_fwd(ws, "dec", "decoder", pk, rs2, h_f, h_p, gap2.data_ptr(), split)
        if self.mark is not None:
            self.mark("fwd_decoder_done")
        mel = ws.f32("Tmel_pred", rs2, odim)
        wh = pk["head"]
        O.gemm(a=d_p, b_ptr=wh.ptr, ldb=wh.ld, m=rs2.rows, n=odim, bias=m.mel_output_layer.bias, rowmask_ptr=len2.data_ptr(),
               out_f32_ptr=mel.ptr, ldo=odim)

        main.wait_event(ev_dur)                                     # predicted durations from the side stream
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        # use_masking=False (fastspeech_loss.py:63-67): means over the padded tensors = the masked sums taken with full lengths
        ml_loss = ml if m.use_masking else torch.full_like(ml, T2)
        tl_loss = tl if m.use_masking else torch.full_like(tl, T1)
        O.masked_losses(mel.ptr, odim, speech, ml_loss, dur, lde, tl_loss, out3, ws.tensor("loss_ws", (1024,)), B, T1, rs1.Tp, T2, rs2.Tp, odim)

        # ============================ backward
        if self.mark is not None:
            self.mark("backward_start")
        g = self.g
        dmel_f = ws.f32("Bdmel_f", rs2, odim)
        dmel_p = ws.plane("Bdmel_p", rs2, odim, split)
        ddur = ws.tensor("Bddur", (rs1.rows,))
        L.check(_lib().efts_loss_bwd(mel.ptr, odim, speech.data_ptr(), ml_loss.data_ptr(), dur.data_ptr(), lde.data_ptr(), tl_loss.data_ptr(),
                                     _ptr(gscale), dmel_f.ptr, None, 0, split, ddur.data_ptr(), B, T1, rs1.Tp, T2, rs2.Tp, odim,
                                     O._stream()), "efts_loss_bwd")
        ev_loss = torch.cuda.Event()
        ev_loss.record(main)
        side.wait_event(ev_loss)                                    # d(dur) is ready
        with O.on_stream(side):
            self._ws_tag = "s"
            # ---- duration predictor (input text_value is NOT detached: efficient_tts.py:219)
            def gname(i, k):
                return f"duration_predictor.conv.{i}.{k}"
            dz2_f, dz2_p = ws.f32("Bdur_dz2", rs1, C), ws.plane("Bdur_dz2p", rs1, C, split)
            L.check(_lib().efts_layernorm_bwd(h2_f.ptr, ln1.weight.data_ptr(), ln1.bias.data_ptr(), ln1.eps, None, ddur.data_ptr(),
                                              dp.linear.weight.data_ptr(), None, dz2_f.ptr, dz2_p.ptr, dz2_p.ld, split,
                                              g[gname(1, "2.weight")].data_ptr(), g[gname(1, "2.bias")].data_ptr(),
                                              g[gname(1, "0.bias")].data_ptr(), g["duration_predictor.linear.weight"].data_ptr(),
                                              g["duration_predictor.linear.bias"].data_ptr(), rs1.rows, C, drop_p, seed1, sadd, O._stream()),
                    "efts_layernorm_bwd")
            dur_items = []                                           # both k3 weight gradients in one grouped launch, below
            self._wgrad_any(ws, dz2_f.ptr, dz2_p, l1_f.ptr, l1_p, C, C, 3, rs1.rows, g[gname(1, "0.weight")], defer=dur_items)
            G1 = ws.f32("Bdur_G1", rs1, C)
            wt = self.wt["dur.1"]
            O.gemm(a=dz2_p, b_ptr=wt.ptr, ldb=wt.ld, b_tap_stride=wt.tap_stride, taps=3, m=rs1.rows, n=C, out_f32_ptr=G1.ptr, ldo=C)
            dz1_f, dz1_p = ws.f32("Bdur_dz1", rs1, C), ws.plane("Bdur_dz1p", rs1, C, split)
            L.check(_lib().efts_layernorm_bwd(h1_f.ptr, ln0.weight.data_ptr(), ln0.bias.data_ptr(), ln0.eps, G1.ptr, None, None,
                                              gap1.data_ptr(), dz1_f.ptr, dz1_p.ptr, dz1_p.ld, split,
                                              g[gname(0, "2.weight")].data_ptr(), g[gname(0, "2.bias")].data_ptr(),
                                              g[gname(0, "0.bias")].data_ptr(), None, None, rs1.rows, C, drop_p, seed0, sadd, O._stream()),
                    "efts_layernorm_bwd")
            self._wgrad_any(ws, dz1_f.ptr, dz1_p, val_f.ptr, val_p, C, C, 3, rs1.rows, g[gname(0, "0.weight")], defer=dur_items)
            if dur_items:
                self._wgrad_group(ws, dur_items, C, C, rs1.rows, 3, split)
            dV_dur = ws.f32("BdV_dur", rs1, C)
            wt = self.wt["dur.0"]
            O.gemm(a=dz1_p, b_ptr=wt.ptr, ldb=wt.ld, b_tap_stride=wt.tap_stride, taps=3, m=rs1.rows, n=C, out_f32_ptr=dV_dur.ptr, ldo=C)
            ev_durb = torch.cuda.Event()
            ev_durb.record(side)
            # operand copies of the alignment backward that depend on forward tensors only (V as an A operand, alpha' as an A operand,
            # K^T, Q^T): this stream idles through the decoder's backward, the main one would run them one after the other in front of
            # their GEMMs
            val_p2 = ws.plane("Bval_p2", rs1, C, 2)
            L.check(_lib().efts_pack_rows(val_f.ptr, None, val_p2.ptr, val_p2.ld, B, rs1.Tp, rs1.Tp, C, C, 2, O._stream()), "efts_pack_rows")
            ra1_p = ws.plane("Bra1_p", rs1, T2, 2)
            O.pack_rows(ralpha, None, ra1_p, rs1)
            kt = ws.raw_plane("Bkt", B * C + 136, T1, 2)
            O.pack_vt(key_f, kt, B, T1, rs1.Tp, C)
            qt = ws.raw_plane("Bqt", B * C + 136, T2, 2)
            O.pack_vt(q_f, qt, B, T2, rs2.Tp, C)
            ev_packs = torch.cuda.Event()
            ev_packs.record(side)
            self._ws_tag = ""
        # mel head (Linear 512->80, masked): bias grad + operand plane, wgrad, dgrad
        if m.use_masking:
            dmel_m = dmel_f                                          # already zero beyond each item's length
            self._act_bwd(dmel_f.ptr, None, None, None, 0, None, dmel_p, g["mel_output_layer.bias"], rs2.rows, odim)
        else:
            # the unmasked loss sees (0 - speech) on padded frames; mel_pred = masked_fill(head output) blocks that gradient (:199-200)
            dmel_m = ws.f32("Bdmel_m", rs2, odim)
            self._act_bwd(dmel_f.ptr, None, None, len2.data_ptr(), 0, dmel_m, dmel_p, g["mel_output_layer.bias"], rs2.rows, odim)
        wst = m._aux_stream(dev) if (_WGRAD_STREAM and m.side_stream) else None
        wst2 = wst if (_WGRAD_STREAM & 2) else None
        with self._forked(wst2):
            if not self._wgrad_narrow(ws, "head", dmel_p, d_p, odim, C, rs2.rows, g["mel_output_layer.weight"]):
                self._wgrad(ws, dmel_m.ptr, odim, d_f.ptr, C, C, 1, rs2.rows, None, None, g["mel_output_layer.weight"], None)
        G = ws.f32("Bdec_Gh", rs2, C)
        wt = self.wt["head"]
        O.gemm(a=dmel_p, b_ptr=wt.ptr, ldb=wt.ld, m=rs2.rows, n=C, rowmask_ptr=gap2.data_ptr(), out_f32_ptr=G.ptr, ldo=C)
        # decoder; its input gradient dH is masked like H (efficient_tts.py:193-194) and also emitted as a split-2 plane
        dH_p = ws.plane("BdH_p", rs2, C, 2)
        dH = self._stack_bwd(ws, "dec", "decoder", rs2, G, dec_saved, gap2.data_ptr(), len2.data_ptr(), dH_p, wgrad_stream=wst if (_WGRAD_STREAM & 1) else None)
        if self.mark is not None:
            self.mark("bwd_decoder_done")
        if self.bucket_hook and wst is None:
            self.bucket_hook(0)

        # ---- expand bmm backward: d alpha' [B,T1,T2] and dV
        main.wait_event(ev_packs)
        dAp = ws.tensor("BdAp", (B, T1, T2))
        O.gemm(a=val_p2, b_ptr=dH_p.ptr, ldb=dH_p.ld, m=T1, n=T2, batch=B, a_batch_stride=rs1.Tp * val_p2.ld,
               b_batch_stride=rs2.Tp * dH_p.ld, out_f32_ptr=dAp.data_ptr(), ldo=T2, out_batch_stride=T1 * T2)
        dHt = ws.raw_plane("BdHt", B * C + 136, T2, 2)              # dH^T per item: [B][C][K = j]
        GV = ws.f32("BGV", rs1, C)
        GV_p = ws.plane("BGV_p", rs1, C, split)

        def dv_branch():                                             # dV = alpha' . dH (+ the duration predictor's dV): transpose of dH + one batched product
            O.pack_vt(dH, dHt, B, T2, rs2.Tp, C)
            O.gemm(a=ra1_p, b_ptr=dHt.ptr, ldb=dHt.ld, m=T1, n=C, batch=B, a_batch_stride=rs1.Tp * ra1_p.ld, b_batch_stride=C * dHt.ld,
                   resid_ptr=dV_dur.ptr, ldr=C, resid_batch_stride=rs1.Tp * C, rowmask_ptr=len1.data_ptr(), rowmask_batch_stride=rs1.Tp,
                   out_f32_ptr=GV.ptr, ldo=C, out_batch_stride=rs1.Tp * C, out_plane=GV_p, outb_batch_stride=rs1.Tp * GV_p.ld)

        ev_gv = None
        if _GV_ON_SIDE and m.side_stream and not torch.cuda.is_current_stream_capturing():
            # nothing between here and dK reads dV: the branch runs on the text-side stream (idle until dK / dV exist, and the consumer of both)
            # beside the chain d alpha' -> e -> pi -> soft index -> scores -> dQ, dK instead of in front of it.  Eager launches only: 3.51-3.63 -> 3.46-3.56 ms per
            # step; captured into the step's hipGraph the extra branch costs 0.1 ms (3.41 vs 3.30 ms, bf16x3 6.55 vs 6.27: the replayed graph's stream
            # assignment loses the overlap of the weight-gradient stream), so a capturing pass keeps the branch on the main stream
            ev_dh = torch.cuda.Event()
            ev_dh.record(main)
            side.wait_event(ev_dh)
            with O.on_stream(side):
                dv_branch()                                          # (dV_dur and alpha' as an operand were produced on this stream)
                ev_gv = torch.cuda.Event()
                ev_gv.record(side)
        else:
            main.wait_event(ev_durb)                                 # dV of the duration predictor (and its gradients: bucket 1)
            dv_branch()

        # ---- alpha' -> e -> pi -> soft index -> scores
        de, dpi, dsx = ws.tensor("Bde", (B, T1)), ws.tensor("Bdpi", (B, T2)), ws.tensor("Bdsx", (B, T2))
        L.check(_lib().efts_alpha_bwd(ralpha.data_ptr(), dAp.data_ptr(), e.data_ptr(), tl.data_ptr(), ml.data_ptr(), float(m.sigma),
                                      ws.tensor("Br", (B, T2)).data_ptr(), de.data_ptr(), B, T1, T2, O._stream()), "efts_alpha_bwd")
        L.check(_lib().efts_e_bwd(imv.data_ptr(), e.data_ptr(), de.data_ptr(), tl.data_ptr(), ml.data_ptr(), float(m.sigma_e),
                                  ws.tensor("Bstats", (2, B, T1)).data_ptr(), dpi.data_ptr(), B, T1, T2, O._stream()), "efts_e_bwd")
        L.check(_lib().efts_imv_bwd(sidx.data_ptr(), imv.data_ptr(), dpi.data_ptr(), tl.data_ptr(), ml.data_ptr(), dsx.data_ptr(), B, T2,
                                    O._stream()), "efts_imv_bwd")
        dS = ws.tensor("BdS", (B, T2, T1))
        dS_p = ws.plane("BdS_p", rs2, T1, 2)
        L.check(_lib().efts_attn_bwd(scores.data_ptr(), T1, sidx.data_ptr(), dsx.data_ptr(), tl.data_ptr(), ml.data_ptr(), dS.data_ptr(),
                                     T1, dS_p.ptr, dS_p.ld, B, T1, T2, rs2.Tp, O._stream()), "efts_attn_bwd")
        # dQ = scale * dS K ; dK = scale * dS^T Q
        GQ = ws.f32("BGQ", rs2, C)
        O.gemm(a=dS_p, b_ptr=kt.ptr, ldb=kt.ld, m=T2, n=C, batch=B, a_batch_stride=rs2.Tp * dS_p.ld, b_batch_stride=C * kt.ld, alpha=scale,
               out_f32_ptr=GQ.ptr, ldo=C, out_batch_stride=rs2.Tp * C)
        dSt = ws.raw_plane("BdSt", B * T1 + 264, T2, 2)             # dS^T: rows (b,i), K = j
        L.check(_lib().efts_pack_vt(dS.data_ptr(), T1, dSt.ptr, dSt.ld, B, T2, T2, T1, O._stream()), "efts_pack_vt")
        GK = ws.f32("BGK", rs1, C)
        GK_p = ws.plane("BGK_p", rs1, C, split)
        shared = m.share_text_encoder_key_value                     # value = key projection: its gradient joins dK here (residual)
        if ev_gv is not None:
            main.wait_event(ev_gv)                                   # (shared: dV is the residual of the next launch; else: one join for everything behind)
        O.gemm(a=dSt, b_ptr=qt.ptr, ldb=qt.ld, m=T1, n=C, batch=B, a_batch_stride=T1 * dSt.ld, b_batch_stride=C * qt.ld, alpha=scale,
               resid_ptr=GV.ptr if shared else None, ldr=C, resid_batch_stride=rs1.Tp * C,
               rowmask_ptr=len1.data_ptr(), rowmask_batch_stride=rs1.Tp, out_f32_ptr=GK.ptr, ldo=C, out_batch_stride=rs1.Tp * C,
               out_plane=GK_p, outb_batch_stride=rs1.Tp * GK_p.ld)

        ev_gk = torch.cuda.Event()
        ev_gk.record(main)
        if self.mark is not None:
            self.mark("bwd_alignment_done")
        side.wait_event(ev_gk)                                      # dK, dV are ready
        if wst is not None and self.bucket_hook:
            # data parallel: bucket 0 (mel head + decoder) is final once the decoder's weight gradients are through; its exchange then
            # overlaps the encoders' backward instead of the alignment block's as well
            main.wait_stream(wst)
            self.bucket_hook(0)

        def kv_param_grads():                                        # bias + weight gradients of the value / key Linears
            sc = ws.f32("Bscratch1" + self._ws_tag, rs1, C)
            kv_items = []                                            # (value and key Linears: one grouped launch)
            if not shared:
                L.check(_lib().efts_act_bwd(GV.ptr, None, None, None, 0.0, 0, sc.ptr, None, 0, 1,
                                            g["text_encoder_value.bias"].data_ptr(), rs1.rows, C, O._stream()), "efts_act_bwd")
                self._wgrad_any(ws, GV.ptr, GV_p, te_f.ptr, te_p, C, C, 1, rs1.rows, g["text_encoder_value.weight"], defer=kv_items)
            L.check(_lib().efts_act_bwd(GK.ptr, None, None, None, 0.0, 0, sc.ptr, None, 0, 1,
                                        g["text_encoder_key.bias"].data_ptr(), rs1.rows, C, O._stream()), "efts_act_bwd")
            self._wgrad_any(ws, GK.ptr, GK_p, te_f.ptr, te_p, C, C, 1, rs1.rows, g["text_encoder_key.weight"], defer=kv_items)
            if kv_items:
                self._wgrad_group(ws, kv_items, C, C, rs1.rows, 1, split)

        with O.on_stream(side):
            self._ws_tag = "s"
            # ---- value / key Linears -> text encoder -> embedding
            kv_param_grads()
            Gt0, Gt = ws.f32("Bte_G0", rs1, C), ws.f32("Bte_G1x", rs1, C)
            wtk = self.wt["key"]
            if shared:                                               # GK already holds dK + dV
                O.gemm(a=GK_p, b_ptr=wtk.ptr, ldb=wtk.ld, m=rs1.rows, n=C, rowmask_ptr=gap1.data_ptr(), out_f32_ptr=Gt.ptr, ldo=C)
            else:
                wtv = self.wt["value"]
                O.gemm(a=GV_p, b_ptr=wtv.ptr, ldb=wtv.ld, m=rs1.rows, n=C, rowmask_ptr=gap1.data_ptr(), out_f32_ptr=Gt0.ptr, ldo=C)
                O.gemm(a=GK_p, b_ptr=wtk.ptr, ldb=wtk.ld, m=rs1.rows, n=C, resid_ptr=Gt0.ptr, ldr=C, rowmask_ptr=gap1.data_ptr(),
                       out_f32_ptr=Gt.ptr, ldo=C)
            Ge = self._stack_bwd(ws, "te", "text_encoder", rs1, Gt, te_saved, gap1.data_ptr(), gap1.data_ptr(), None)
            L.check(_lib().efts_embed_bwd(text.data_ptr(), Ge.ptr, g["text_embedding_table.weight"].data_ptr(), B, T1, rs1.Tp, C,
                                          m.num_symbols, O._stream()), "efts_embed_bwd")
            self._ws_tag = ""

        # ---- mel encoder + prenet
        G_me = GQ
        if m.mel_query_fc is not None:                               # backward of q = Linear(mel_h) (efficient_tts.py:163-164)
            GQ_p = ws.plane("BGQ_p", rs2, C, split)
            self._act_bwd(GQ.ptr, None, None, gap2.data_ptr(), 0, None, GQ_p, g["mel_query_fc.bias"], rs2.rows, C)
            self._wgrad_any(ws, GQ.ptr, GQ_p, mh_f.ptr, mh_p, C, C, 1, rs2.rows, g["mel_query_fc.weight"])
            G_me = ws.f32("BG_mh", rs2, C)
            wtq = self.wt["qfc"]
            O.gemm(a=GQ_p, b_ptr=wtq.ptr, ldb=wtq.ld, m=rs2.rows, n=C, rowmask_ptr=gap2.data_ptr(), out_f32_ptr=G_me.ptr, ldo=C)
        Gm = self._stack_bwd(ws, "me", "mel_encoder", rs2, G_me, me_saved, gap2.data_ptr(), gap2.data_ptr(), None, wgrad_stream=wst2)
        if self.mark is not None:
            self.mark("bwd_mel_encoder_done")
        narrow = self._narrow_ok(split, mel_in.split, C, odim, max(C, 128) * 2, mel_in.ld)        # (the dZ plane allocated below is C wide)
        dzp_f = None if narrow else ws.f32("Bpre_dz", rs2, C)
        dzp_p = ws.plane("Bpre_dzp", rs2, C, split) if narrow else None
        if pre_z is not None:
            O.act_grad(m.act_general, Gm.ptr, pre_z.ptr, gap2.data_ptr(), dzp_f, dzp_p, g["mel_prenet.0.bias"], rs2.rows, C, pre_dp, pre_seed)
        else:
            self._act_bwd(Gm.ptr, pre_f.ptr, None, gap2.data_ptr(), 3, dzp_f, dzp_p, g["mel_prenet.0.bias"], rs2.rows, C, pre_dp, pre_seed)
        with self._forked(wst2):
            if not (narrow and self._wgrad_narrow(ws, "pre", dzp_p, mel_in, C, odim, rs2.rows, g["mel_prenet.0.weight"])):
                assert dzp_f is not None
                self._wgrad(ws, dzp_f.ptr, C, mel_in_f.ptr, odim, odim, 1, rs2.rows, None, None, g["mel_prenet.0.weight"], None)
        if self.bucket_hook:
            if wst is not None:
                main.wait_stream(wst)
            self.bucket_hook(1)

        main.wait_stream(side)                                      # text-side gradients (bucket 2) and everything else enqueued there
        if wst is not None and not self.bucket_hook:
            main.wait_stream(wst)                                   # the weight gradients of the mel-length layers
        if self.bucket_hook:
            self.bucket_hook(2)

        aux = None
        if keep:
            aux = dict(imv=imv, ralpha=ralpha, mel=mel, e=e, dH=dH, dAp=dAp, de=de, dpi=dpi, dsx=dsx, dS=dS, GQ=GQ, GK=GK, GV=GV,
                       Gm=Gm, Ge=Ge, rs1=rs1, rs2=rs2, ddur=ddur)
        return out3, aux
