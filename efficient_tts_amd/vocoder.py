"""HiFi-GAN V1 generator on the MI355X contraction kernel.

Drop-in for `nntts.vocoders.hifigan_model.Generator` (nntts/vocoders/hifigan_model.py:95-150): same
constructor argument (the config object / dict of HiFiGAN_LJ_V1/config.json), same parameter names and
shapes (`state_dict` of the reference loads unchanged, weight_g / weight_v pairs), `forward(mel [B, 80, T])
-> audio [B, 1, T * prod(upsample_rates)]`, `remove_weight_norm()`.  Inference only.

Every Conv1d / ConvTranspose1d is one `efts_gemm` launch on channel-last rows (one row per sample):
  * dilated k = 3 / 7 / 11 convolutions: `taps`, `dilation` of the kernel (the tap offset is a row offset into
    the LDS window, exactly like the acoustic model's k5 convolutions);
  * the residual blocks are pre-activation (`conv(leaky(x))`, :45-52): the PRODUCER of x writes the operand
    plane of leaky(x) (`plane_act`), so no activation pass exists; `xt` between the two convolutions of a
    pair only ever lives as that plane (no fp32 copy);
  * ConvTranspose1d(stride u, kernel 2u, padding u/2) = a 2-tap convolution with u * cout output columns:
    y[n] = x[q] w[r] + x[q-1] w[r+u] with n + u/2 = q u + r; the [T+1][u*cout] result IS the [u(T+1)][cout] row
    space of the next stage (a pointer offset of u/2 rows);
  * the multi-receptive-field mean and the LeakyReLU of the next layer's input: `efts_mean_act_rows`;
  * conv_post + tanh: the kernel's tanh epilogue.
Batches: the B utterances share one row space per stage, item b at rows b * P .. b * P + T_b * up with a pitch of
P = (T + 4) * up rows (the 4 spare mel frames become >= 32 zero rows from the first stage on: more than the largest
halo, (11 - 1) / 2 * 5 = 25).  Every launch carries the per-row validity mask of its stage, so rows past an item's own
length stay zero exactly as the zero padding of a single-utterance run does; `forward(mel, lengths)` therefore equals
B separate calls on the unpadded items.
There is no CPU path.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch.nn.utils import remove_weight_norm, weight_norm

from . import lib as L
from . import ops as O
from .ops import PackedWeight, Plane

LRELU_SLOPE = 0.1
_GUARD = 64            # zero rows in front of every row space: (taps - 1) / 2 * dilation <= 25
_GAP_FRAMES = 4        # spare mel frames between the items of a batch: >= 32 zero rows at every upsampled stage


def _cfg(h, key, default=None):
    return h[key] if isinstance(h, dict) and key in h else getattr(h, key, default)


class _ResBlock1(nn.Module):
    """parameter container with the reference's names (hifigan_model.py:30-43); never called"""

    def __init__(self, channels: int, kernel_size: int, dilation):
        super().__init__()
        pad = lambda d: (kernel_size * d - d) // 2
        self.convs1 = nn.ModuleList([weight_norm(nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=pad(d)))
                                     for d in dilation])
        self.convs2 = nn.ModuleList([weight_norm(nn.Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=pad(1)))
                                     for _ in dilation])
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)


class _Rows:
    """fp32 [rows][c] stream + its operand plane, both with zero guard rows"""

    def __init__(self, rows: int, c: int, split: int, dev, f32: bool = True, plane: bool = True):
        self.rows, self.c = rows, c
        alloc = _GUARD + O.roundup(rows, 128) + 256
        self.f = torch.zeros(alloc, c, dtype=torch.float32, device=dev) if f32 else None
        self.p = Plane(alloc, c, split, dev, guard_lo=_GUARD) if plane else None

    @property
    def fptr(self) -> int:
        return self.f.data_ptr() + _GUARD * self.c * 4


class HiFiGANGenerator(nn.Module):
    def __init__(self, h, precision: str = "bf16x3"):
        super().__init__()
        if str(_cfg(h, "resblock", "1")) != "1":
            raise NotImplementedError("only ResBlock1 (the HiFiGAN_LJ_V1 configuration) is implemented")
        if precision not in ("bf16x3", "bf16"):
            raise ValueError("precision must be 'bf16x3' or 'bf16'")
        self.precision, self.split = precision, 2 if precision == "bf16x3" else 1
        self.upsample_rates = tuple(_cfg(h, "upsample_rates"))
        self.upsample_kernel_sizes = tuple(_cfg(h, "upsample_kernel_sizes"))
        self.res_kernels = tuple(_cfg(h, "resblock_kernel_sizes"))
        self.res_dilations = tuple(tuple(d) for d in _cfg(h, "resblock_dilation_sizes"))
        c0 = int(_cfg(h, "upsample_initial_channel"))
        self.num_mels = int(_cfg(h, "num_mels", 80))
        for u, k in zip(self.upsample_rates, self.upsample_kernel_sizes):
            if k != 2 * u or u % 2:
                raise NotImplementedError("transposed convolutions are implemented for kernel = 2 * stride, even stride")
        self.conv_pre = weight_norm(nn.Conv1d(self.num_mels, c0, 7, 1, padding=3))
        self.ups = nn.ModuleList([weight_norm(nn.ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, u, padding=(k - u) // 2))
                                  for i, (u, k) in enumerate(zip(self.upsample_rates, self.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(self.res_kernels, self.res_dilations):
                self.resblocks.append(_ResBlock1(ch, k, d))
        self.conv_post = weight_norm(nn.Conv1d(ch, 1, 7, 1, padding=3))
        self._packed: Optional[Dict[str, PackedWeight]] = None
        self._packed_dev = None
        self._bias: Dict[str, torch.Tensor] = {}
        self._bufs: Dict[int, dict] = {}
        # per-shape hipGraph of the ~85 launches of a call (round 6; efficient_tts_amd/graphs.py, the acoustic model's cache): one utterance is
        # 2 ms of device work issued by a host that needs about as long for the launches; `graphs = False`: every launch issued eagerly
        self.graphs = True
        self.branch_streams = True          # the residual blocks of a stage on streams of their own (False: one after the other, A/B)
        from .graphs import GraphCache
        object.__setattr__(self, "_graph_cache", GraphCache(capacity=4))

    # ------------------------------------------------------------------ reference API
    def remove_weight_norm(self):
        self._graph_cache.clear()
        for m in [self.conv_pre, self.conv_post, *self.ups]:
            remove_weight_norm(m)
        for rb in self.resblocks:
            for m in [*rb.convs1, *rb.convs2]:
                remove_weight_norm(m)
        self._packed = None

    def load_state_dict(self, *a, **k):
        self._packed = None
        self._graph_cache.clear()
        return super().load_state_dict(*a, **k)

    # ------------------------------------------------------------------ weights -> operand planes (once)
    @staticmethod
    def _folded(m: nn.Module) -> torch.Tensor:
        if hasattr(m, "weight_g"):
            # weight_norm, dim 0 (Conv1d: per cout; ConvTranspose1d: per cin) -- the very op remove_weight_norm bakes in
            return torch._weight_norm(m.weight_v.detach(), m.weight_g.detach(), 0)
        return m.weight.detach()

    def _pack(self, dev) -> Dict[str, PackedWeight]:
        if self._packed is not None and self._packed_dev == dev:
            return self._packed
        self._bufs.clear()                      # workspaces of another device are of no use either
        self._packed_dev = dev
        pk: Dict[str, PackedWeight] = {}
        with O.stream_scope():
            def conv(name, m):
                w = self._folded(m).float().contiguous()
                pk[name] = PackedWeight(w.shape[0], w.shape[1], w.shape[2], self.split, dev)
                pk[name].pack(w)
                self._bias[name] = m.bias.detach().float().contiguous()

            conv("conv_pre", self.conv_pre)
            conv("conv_post", self.conv_post)
            for i, (m, u) in enumerate(zip(self.ups, self.upsample_rates)):
                w = self._folded(m).float()                                  # [cin][cout][2u]
                cin, cout, _ = w.shape
                # B[tap][n = r * cout + co][ci]: tap 0 multiplies x[q-1] (w[.., r + u]), tap 1 multiplies x[q] (w[.., r]), tap 2 = 0
                b = torch.zeros(u * cout, cin, 3, device=dev)
                b[:, :, 0] = w[:, :, u:].permute(2, 1, 0).reshape(u * cout, cin)
                b[:, :, 1] = w[:, :, :u].permute(2, 1, 0).reshape(u * cout, cin)
                pk[f"ups.{i}"] = PackedWeight(u * cout, cin, 3, self.split, dev)
                pk[f"ups.{i}"].pack(b.contiguous())
                self._bias[f"ups.{i}"] = m.bias.detach().float().repeat(u).contiguous()
            for n, rb in enumerate(self.resblocks):
                for d in range(len(rb.dilation)):
                    conv(f"rb{n}.c1.{d}", rb.convs1[d])
                    conv(f"rb{n}.c2.{d}", rb.convs2[d])
        self._packed = pk
        return pk

    # ------------------------------------------------------------------ buffers per mel length
    def _workspace_for(self, B: int, T: int, dev) -> dict:
        key = (B, T)
        if key in self._bufs:
            return self._bufs[key]
        if len(self._bufs) > 2:
            self._bufs.pop(next(iter(self._bufs)))
        sp = self.split
        pitch = T + _GAP_FRAMES                                                # rows per item at mel rate
        rows = B * pitch
        b = {"pitch": pitch, "mel": _Rows(rows, self.num_mels, sp, dev, f32=False),
             "pre": _Rows(rows, self.conv_pre.out_channels, sp, dev, f32=False), "mask": []}
        for i, (m, u) in enumerate(zip(self.ups, self.upsample_rates)):
            cout = m.out_channels
            n_i = rows * u
            b[f"up{i}"] = _Rows((rows + 1) * u, cout, sp, dev, plane=False)  # the [rows + 1][u * cout] GEMM result, seen as rows of cout
            b[f"x{i}"] = _Rows(n_i, cout, sp, dev, f32=False)                # plane of leaky(x_i)
            b[f"t{i}"] = [_Rows(n_i, cout, sp, dev, f32=False) for _ in self.res_kernels]   # plane of leaky(xt), one per residual block (they run side by side)
            b[f"r{i}"] = [[_Rows(n_i, cout, sp, dev) for _ in range(2)] for _ in self.res_kernels]   # ping-pong per residual block
            b[f"s{i}"] = _Rows(n_i, cout, sp, dev, f32=False)                # plane of leaky(mean)
            b["mask"].append(torch.zeros(n_i + 1, dtype=torch.float32, device=dev))
            rows = n_i
        b["mask_mel"] = torch.zeros(B * pitch + 1, dtype=torch.float32, device=dev)
        b["out"] = torch.zeros(rows + 256, 1, dtype=torch.float32, device=dev)
        self._bufs[key] = b
        return b

    def _branch_streams(self, dev, count: int):
        st = getattr(self, "_bstreams", None)
        if st is None or len(st) != count or (count and st[0].device != dev):
            st = [torch.cuda.Stream(device=dev) for _ in range(count)]
            object.__setattr__(self, "_bstreams", st)
        return st

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel [B, num_mels, T] -> audio [B, 1, T * hop].  `lengths` (frames per item, optional): every item is
        synthesised exactly as if it had been passed alone without its padding; samples past lengths[b] * hop are 0."""
        if x.dim() != 3 or x.shape[1] != self.num_mels:
            raise ValueError(f"expected mel [B, {self.num_mels}, T]")
        if not x.is_cuda:
            raise RuntimeError("HiFiGANGenerator runs on an MI355X device only (no CPU path)")
        L.load()
        L.require_device()
        dev = x.device
        pk = self._pack(dev)
        B, _, T = x.shape
        hop = 1
        for u in self.upsample_rates:
            hop *= u
        if lengths is None:
            lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        else:
            lens = lengths.to(device=dev, dtype=torch.int32).clamp(0, T)
        mel = x.transpose(1, 2).float()
        if lengths is not None:                 # what the padding frames hold must not reach conv_pre's halo
            mel = mel * (torch.arange(T, device=dev)[None, :] < lens[:, None])[:, :, None]
        mel = mel.contiguous()

        def body(mel_btc, lens_i32):
            audio = torch.zeros(B, 1, T * hop, dtype=torch.float32, device=dev)
            with O.stream_scope():
                self._run(pk, mel_btc, lens_i32, audio, B, T, hop, dev)
            return (audio,)

        if not self.graphs or torch.cuda.is_current_stream_capturing():
            return body(mel, lens)[0]
        ws = self._workspace_for(B, T, dev)
        # the graph is valid while the buffers its launches point at live: this workspace and the packed planes
        tag = (id(ws), tuple(w.ptr for w in pk.values()), self.split)
        return self._graph_cache.run(("voc", B, T), tag, (mel, lens), body, keepalive=(ws, pk), adaptive=True)[0]

    def _conv(self, pk, name, a: Plane, rows, taps, dil, *, mask=None, out_f=None, ldo=0, out_p=None, plane_slope=None, resid=None,
              ldr=0, act=L.ACT_NONE):
        w = pk[name]
        O.gemm(a=a, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=taps, m=rows, n=w.cout, bias=self._bias[name], act=act,
               resid_ptr=resid, ldr=ldr, rowmask_ptr=None if mask is None else mask.data_ptr(), out_f32_ptr=out_f, ldo=ldo,
               out_plane=out_p, dilation=dil, plane_act=plane_slope is not None, plane_slope=0.0 if plane_slope is None else plane_slope)

    def _run(self, pk, mel_btc: torch.Tensor, lens: torch.Tensor, audio: torch.Tensor, B: int, T: int, hop: int, dev) -> None:
        b = self._workspace_for(B, T, dev)
        lib = L.load()
        sp = self.split
        pitch = b["pitch"]
        st = O._stream()

        def masks(into: torch.Tensor, lengths_i32: torch.Tensor, t_max: int, t_pitch: int):
            L.check(lib.efts_row_masks(lengths_i32.data_ptr(), None, into.data_ptr(), B, t_max, t_pitch, st), "efts_row_masks")

        mel_p = b["mel"].p
        L.check(lib.efts_pack_rows(mel_btc.data_ptr(), None, mel_p.ptr, mel_p.ld, B, T, pitch, self.num_mels,
                                   mel_p.nchunk * O.chunk_k(sp), sp, st), "efts_pack_rows")
        masks(b["mask_mel"], lens, T, pitch)
        # conv_pre (:118); its only consumer applies leaky first (:120) -> the plane of leaky(x), no fp32 copy
        rows = B * pitch
        self._conv(pk, "conv_pre", mel_p, rows, 7, 1, mask=b["mask_mel"], out_p=b["pre"].p, plane_slope=LRELU_SLOPE)
        cur_p, n, up_total = b["pre"].p, 0, 1
        for i, u in enumerate(self.upsample_rates):
            cout = self.ups[i].out_channels
            up_total *= u
            n_i = rows * u
            mask = b["mask"][i]
            masks(mask, lens * up_total, T * up_total, pitch * up_total)
            up = b[f"up{i}"]
            # ConvTranspose1d as a 2-tap convolution over the input rows + 1 (the row after the last one is a zero guard row)
            self._conv(pk, f"ups.{i}", cur_p, rows + 1, 3, 1, out_f=up.fptr, ldo=u * cout)
            x_f = up.fptr + (u // 2) * cout * 4                               # y[n] = row n + u/2 of the flattened result
            x_p = b[f"x{i}"].p
            # plane of leaky(x): x * (x > 0 ? 1 : slope), the identity-residual LeakyReLU mode of efts_act_bwd; the mask drops
            # the transposed convolution's cropped border samples, which land in the zero rows between items
            L.check(lib.efts_act_bwd(x_f, x_f, None, mask.data_ptr(), LRELU_SLOPE, 3, None, x_p.ptr, x_p.ld, sp, None, n_i, cout, st),
                    "efts_act_bwd")
            # The residual blocks of a stage (kernel sizes 3 / 7 / 11) are independent until their mean (:123-128): each runs as a chain of its own
            # on its own stream (round 6).  At one utterance a launch is 50-200 workgroups for 15-30 us: eighteen of them in a row per stage were
            # latency, not work (profiles/rocprofv3_r06_vocoder_summary.txt); two chains' workgroups fit a CU's LDS side by side.
            finals: List[int] = [0] * len(self.res_kernels)
            main = torch.cuda.current_stream(dev)
            side = self._branch_streams(dev, len(self.res_kernels) - 1) if self.branch_streams else []
            ev_x = torch.cuda.Event()
            ev_x.record(main)
            joins = []
            order = sorted(range(len(self.res_kernels)), key=lambda j: -self.res_kernels[j])       # the longest chain is issued first
            for slot, j in enumerate(order):
                k = self.res_kernels[j]
                rb = self.resblocks[n + j]
                own = side[slot] if slot < len(side) else None                                   # the last (shortest) chain stays on the main stream
                if own is not None:
                    own.wait_event(ev_x)
                with (O.on_stream(own) if own is not None else contextlib.nullcontext()):
                    r_f, r_p, pp = x_f, x_p, b[f"r{i}"][j]
                    tp = b[f"t{i}"][j].p
                    for d_i, d in enumerate(rb.dilation):
                        self._conv(pk, f"rb{n + j}.c1.{d_i}", r_p, n_i, k, d, mask=mask, out_p=tp, plane_slope=LRELU_SLOPE)  # :47-48 (+ :49)
                        nxt = pp[d_i & 1]
                        self._conv(pk, f"rb{n + j}.c2.{d_i}", tp, n_i, k, 1, mask=mask, out_f=nxt.fptr, ldo=cout, out_p=nxt.p,
                                   plane_slope=LRELU_SLOPE, resid=r_f, ldr=cout)                                        # :50-51
                        r_f, r_p = nxt.fptr, nxt.p
                    finals[j] = r_f
                    if own is not None:
                        ev = torch.cuda.Event()
                        ev.record(own)
                        joins.append(ev)
            for ev in joins:
                main.wait_event(ev)
            n += len(self.res_kernels)
            last = i == len(self.upsample_rates) - 1
            s_p = b[f"s{i}"].p
            L.check(lib.efts_mean_act_rows(finals[0], finals[1] if len(finals) > 1 else None, finals[2] if len(finals) > 2 else None,
                                           cout, 1.0 / len(finals), 0.01 if last else LRELU_SLOPE, None, 0, s_p.ptr, s_p.ld, sp, n_i, cout,
                                           st), "efts_mean_act_rows")                                              # :128 (+ :120 / :129)
            cur_p, rows = s_p, n_i
        self._conv(pk, "conv_post", cur_p, rows, 7, 1, mask=b["mask"][-1], out_f=b["out"].data_ptr(), ldo=1, act=L.ACT_TANH)   # :130-131
        audio[:, 0].copy_(b["out"][:rows, 0].view(B, pitch * hop)[:, :T * hop])


def load_hifigan_generator(device, config_path: str, checkpoint_path: str, precision: str = "bf16x3") -> HiFiGANGenerator:
    """The reference's loader (hifigan_model.py:18-28) with explicit paths: JSON config + a checkpoint whose
    "generator" entry is the reference Generator's state_dict (the published `generator_v1` is not shipped with the
    reference repository)."""
    import json
    with open(config_path) as f:
        cfg = json.load(f)
    gen = HiFiGANGenerator(cfg, precision=precision)
    state = torch.load(checkpoint_path, map_location="cpu")
    gen.load_state_dict(state["generator"] if "generator" in state else state)
    gen = gen.to(device).eval()
    gen.remove_weight_norm()
    return gen
