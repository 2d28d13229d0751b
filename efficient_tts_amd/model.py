"""EfficientTTSCNN on MI355X: the host-side mirror of the reference model class.

Same class name, ctor kwargs, ``forward`` / ``inference`` signatures and returns, and
``state_dict`` keys as ``nntts.models.EfficientTTSCNN`` (reference
nntts/models/efficient_tts.py:23-418), so ``getattr(pkg.models, config["model_name"])(**config["model_params"])``
(nntts/bin/train.py:173-185) and reference checkpoints work unchanged.  All arithmetic runs in
hand-written HIP kernels behind the C ABI of ``libefts_hip.so`` (include/efts_abi.h); PyTorch
only owns device memory, streams and the parameter containers.  There is no torch/CPU
fallback: without the library or a gfx950 device the model raises.
"""
from __future__ import annotations

import dataclasses
import os
import logging
from typing import Dict, Optional, Tuple

import torch

from . import lib as L
from . import ops as O
from .graphs import GraphCache, PadTo
from .ops import F32Rows, PackedWeight, Plane, Rows, roundup


# --------------------------------------------------------------------------------------
# Parameter containers.  torch.nn modules are used ONLY to hold/initialise parameters
# with the reference's names and default init order; their forward() is never called.
# --------------------------------------------------------------------------------------
class _ResConv1d(torch.nn.Module):                     # efts_modules.py:19-51 (names only)
    def __init__(self, n_channels, k_size, act, act_params, dropout_rate):
        super().__init__()
        mods = [torch.nn.Conv1d(n_channels, n_channels, kernel_size=k_size, padding=(k_size - 1) // 2),
                getattr(torch.nn, act)(**act_params)]
        if dropout_rate >= 1e-5:
            mods.append(torch.nn.Dropout(dropout_rate))
        self.conv = torch.nn.Sequential(*mods)


class _ResConvBlock(torch.nn.Module):                  # efts_modules.py:54-99 (names only)
    def __init__(self, num_layers, n_channels, k_size, act, act_params, dropout_rate, use_weight_norm):
        super().__init__()
        self.num_layers = num_layers
        self.layers = torch.nn.Sequential(*[_ResConv1d(n_channels, k_size, act, act_params, dropout_rate)
                                            for _ in range(num_layers)])
        if use_weight_norm:
            for m in self.modules():
                if isinstance(m, torch.nn.Conv1d):
                    torch.nn.utils.weight_norm(m)


class _DurationPredictor(torch.nn.Module):             # duration_predictor.py:28-64 (names only)
    def __init__(self, idim, n_layers, n_chans, kernel_size=3, dropout_rate=0.1, offset=1.0):
        super().__init__()
        self.offset, self.n_layers = offset, n_layers
        self.conv = torch.nn.ModuleList()
        for _ in range(n_layers):
            self.conv.append(torch.nn.Sequential(
                torch.nn.Conv1d(n_chans, n_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2),
                torch.nn.ReLU(),
                torch.nn.LayerNorm(n_chans, eps=1e-12),       # layer_norm.py:14-17
                torch.nn.Dropout(dropout_rate)))
        self.linear = torch.nn.Linear(n_chans, 1)


class LazyStats(dict):
    """``stats`` of the reference forward (efficient_tts.py:225-227) without the three
    per-step host syncs: values stay on the device until a key is read."""

    def __init__(self, out3: torch.Tensor):
        super().__init__(loss=None, mel_loss=None, duration_loss=None)
        self._t, self._v = out3, None

    def _vals(self):
        if self._v is None:
            self._v = self._t.tolist()
        return dict(loss=self._v[0], mel_loss=self._v[1], duration_loss=self._v[2])

    def __getitem__(self, k):
        return self._vals()[k]

    def get(self, k, default=None):
        return self._vals().get(k, default)

    def __iter__(self):
        return iter(("loss", "mel_loss", "duration_loss"))

    def keys(self):
        return self._vals().keys()

    def items(self):
        return self._vals().items()

    def values(self):
        return self._vals().values()

    def __repr__(self):
        return repr(self._vals())


class _Workspace:
    """Named device buffers for one (B, T1, T2) shape; zero-initialised once, so guard and gap
    rows stay zero (kernels never write them with non-zero values)."""

    _serial = 0

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, object] = {}
        _Workspace._serial += 1
        self.serial = _Workspace._serial          # identifies the allocation: captured graphs point into these buffers

    def f32(self, name: str, rs: Rows, c: int) -> F32Rows:
        key = ("f", name, rs.B, rs.T, c)
        if key not in self.bufs:
            self.bufs[key] = F32Rows(rs, c, self.device)
        return self.bufs[key]

    def plane(self, name: str, rs: Rows, k: int, split: int) -> Plane:
        key = ("p", name, rs.B, rs.T, k, split)
        if key not in self.bufs:
            self.bufs[key] = Plane.for_rows(rs, k, split, self.device)
        return self.bufs[key]

    def raw_plane(self, name: str, nrows: int, k: int, split: int) -> Plane:
        key = ("rp", name, nrows, k, split)
        if key not in self.bufs:
            self.bufs[key] = Plane(nrows, k, split, self.device)
        return self.bufs[key]

    def get(self, key, make):
        """generic cached object (e.g. transposed planes of the training step)"""
        if key not in self.bufs:
            self.bufs[key] = make()
        return self.bufs[key]

    def tensor(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        key = ("t", name, tuple(shape), dtype)
        if key not in self.bufs:
            self.bufs[key] = torch.zeros(*shape, dtype=dtype, device=self.device)
        return self.bufs[key]


PRECISIONS = {"bf16": 1, "bf16x3": 2}


@dataclasses.dataclass
class LaunchOptions:
    """Everything that changes WHICH launches a pass consists of (not their results): the A/B and test hooks of the model.  One object,
    so that the tags of the captured graphs are derived from it (`tag()`) instead of being enumerated by hand at every capture site
    -- an option that is missing from a tag replays a graph captured under the other setting.  Every field is also readable / writable
    as an attribute of the model (`model.resconv = False`); the finished A/Bs of earlier rounds (text layers as riders of the
    mel-encoder launches, CU halves for the prenet, duration-predictor riders) are no longer options: their winners are the code."""
    resconv: bool = True            # long row spaces: residual stacks on efts_resconv5 (False: efts_gemm + fp32 stream)
    resconv_min_rows: int = 16384   # ... from this many rows on (shorter ones keep the fp32 stream + efts_gemm, whose 124-row tiles fill the chip better)
    side_stream: bool = True        # text-length work on a second HIP stream beside the mel-length kernels
    fuse_prenet: bool = True        # the prenet straight from the fp32 frames (efts_frame_linear); False: efts_pack_rows + efts_gemm
    fuse_soft_index: bool = True    # T1 <= 128: q.k^T, softmax and soft index in one launch (False: scores stored, efts_attn_soft_index)
    fuse_align: bool = True         # imv scan + aligned positions + duration target in one launch (efts_imv_align)
    fuse_expand: bool = True        # T1 <= 256: alpha' generated in registers inside the expand contraction (efts_expand)
    fuse_mel_loss: bool = True      # the loss's mel term in the epilogue of the mel head's launch (efts_gemm_args.sqerr_part) where that launch fills the chip
    fuse_mel_loss_min_wgs: int = 0  # ... i.e. from this many 128-row tiles on (0: the number of CUs; tests lower it)
    embed_conv: bool = True         # eval paths: embedding + text-encoder layer 0 as table look-ups (efts_embed_conv)
    small_m: bool = True            # free-running inference on short row spaces: the K-split small-M tiling of efts_gemm
    small_m_rows: int = 1024        # ... up to this many rows

    def tag(self) -> tuple:
        return dataclasses.astuple(self) + (O.RC_KERNEL, O.GEMM_TILING)       # (the launch-level switches of ops.py change the launches too)


class EfficientTTSCNN(torch.nn.Module):
    """EFTS-CNN acoustic model (drop-in for nntts.models.EfficientTTSCNN).

    Extra keyword (not in the reference): ``precision`` selects the MFMA operand mode of the
    Conv1d/Linear stacks: "bf16x3" (split-bf16, fp32-class accuracy; default) or "bf16".
    The alignment block (QK^T, expand) always runs in bf16x3 / fp32.
    """

    def __init__(self, num_symbols: int, odim: int = 80, symbol_embedding_dim: int = 512, n_channels: int = 512,
                 n_text_encoder_layer: int = 5, n_mel_encoder_layer: int = 3, n_decoder_layer: int = 6,
                 n_duration_layer: int = 2, k_size: int = 5, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True, dropout_rate=0.1,
                 use_masking: bool = False, use_weighted_masking: bool = False, duration_offset=1.0, sigma=0.01,
                 sigma_e=0.5, delta_e_method_1=True, share_text_encoder_key_value=False, use_mel_query_fc=False,
                 precision: str = "bf16x3"):
        super().__init__()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        # LeakyReLU (every reference config) and ReLU (= slope 0) live in the epilogues of the contraction kernels; the other pointwise
        # torch.nn activations run as a separate elementwise launch behind the contraction (csrc/efts_act.hip)
        act_general = None
        if nonlinear_activation == "ReLU" and set(nonlinear_activation_params) - {"inplace"}:
            # the reference builds torch.nn.ReLU(**params) (efts_modules.py:32-35): anything but `inplace` is a TypeError there
            raise TypeError(f"ReLU.__init__() got an unexpected keyword argument {sorted(set(nonlinear_activation_params) - {'inplace'})[0]!r}")
        if nonlinear_activation not in ("LeakyReLU", "ReLU") or set(nonlinear_activation_params) - {"negative_slope", "inplace"}:
            act_general = L.actfn(nonlinear_activation, dict(nonlinear_activation_params))
            if act_general is None:
                raise NotImplementedError(f"nonlinear_activation={nonlinear_activation!r}({nonlinear_activation_params}): the library has forms of "
                                          f"{sorted(L.ACTFN)} with their scalar parameters (learnable or non-pointwise modules have none)")
        if symbol_embedding_dim != n_channels:
            raise ValueError("symbol_embedding_dim must equal n_channels (the reference adds them residually)")
        if k_size % 2 == 0 or not 1 <= k_size <= 11:
            raise NotImplementedError("k_size must be odd and <= 11 (efts_gemm's tap counts; the guard rows in front of a row space hold "
                                      "a padding of up to 5).  5, the value of every reference config, and 3 run the mel-length stacks on "
                                      "efts_resconv5; 1, 7, 9 and 11 on efts_gemm")
        if use_weighted_masking:
            raise NotImplementedError("FastSpeechLoss(use_weighted_masking=True) is not implemented (no shipped config selects it)")
        if n_channels % 256 or odim > 128:
            raise NotImplementedError("n_channels must be a multiple of 256 and odim <= 128")
        self.precision = precision
        self.split = PRECISIONS[precision]
        self.odim, self.n_channels, self.num_symbols = odim, n_channels, num_symbols
        self.duration_offset, self.sigma, self.sigma_e = duration_offset, sigma, sigma_e
        self.delta_e_method_1 = delta_e_method_1
        self.share_text_encoder_key_value = share_text_encoder_key_value
        self.use_masking = bool(use_masking)        # False (the reference ctor default): the two losses are means over the padded tensors
        self.k_size = int(k_size)
        self.row_gap = max(L.GAP, (self.k_size - 1) // 2)           # zero rows between the items of a row space (include/efts_abi.h "Row space")
        # LeakyReLU's own default slope is 0.01 (torch.nn.LeakyReLU); ReLU = slope 0 through the same epilogue
        self.slope = float(nonlinear_activation_params.get("negative_slope", 0.01)) if nonlinear_activation == "LeakyReLU" else 0.0
        self.act_general = act_general                                # (EFTS_ACTFN id, p0, p1) or None = the fused (Leaky)ReLU
        self.dropout_rate = dropout_rate
        a, ap = nonlinear_activation, nonlinear_activation_params
        # construction order == reference (efficient_tts.py:57-112) so a given torch seed
        # yields the reference's default initialisation
        self.text_embedding_table = torch.nn.Embedding(num_symbols, symbol_embedding_dim)
        self.text_encoder = _ResConvBlock(n_text_encoder_layer, n_channels, k_size, a, ap, dropout_rate, use_weight_norm)
        self.text_encoder_key = torch.nn.Linear(n_channels, n_channels)
        if not share_text_encoder_key_value:        # (:72-75) shared: the value IS the key projection
            self.text_encoder_value = torch.nn.Linear(n_channels, n_channels)
        self.mel_prenet = torch.nn.Sequential(torch.nn.Linear(odim, n_channels), getattr(torch.nn, a)(**ap),
                                              torch.nn.Dropout(dropout_rate))
        self.mel_encoder = _ResConvBlock(n_mel_encoder_layer, n_channels, k_size, a, ap, dropout_rate, use_weight_norm)
        self.mel_query_fc = torch.nn.Linear(n_channels, n_channels) if use_mel_query_fc else None      # (:90-95)
        self.decoder = _ResConvBlock(n_decoder_layer, n_channels, k_size, a, ap, dropout_rate, use_weight_norm)
        self.mel_output_layer = torch.nn.Linear(n_channels, odim)
        self.duration_predictor = _DurationPredictor(n_channels, n_duration_layer, n_channels, offset=duration_offset)
        self.opt = LaunchOptions()          # launch-affecting options (A/B and test hooks): also plain attributes of the model, see below
        self._free_running = False          # set while inference() / inference_batch() enqueue their launches
        self.graphs = True                  # plain eval calls replay a per-shape hipGraph (False: every kernel launched eagerly)
        self._graph_cache = GraphCache()
        self.graph_policy = "auto"          # teacher-forced forward: "auto" = a shape replays a graph only where launch latency is not
                                            # hidden by the device time anyway (graphs.GraphCache.run, adaptive); "always" = every shape
        self._len1 = {}
        self._infer_cache = GraphCache(capacity=64)      # free-running inference: two phases per (bucketed) shape
        self.dropout_seed = 0x5EED          # base seed of the train-mode dropout masks (mixed with the data-parallel rank and the step counter)
        self._drop_now = None               # set while a train()-mode gradient-free forward enqueues its launches: (conv p, k -> seed, duration p, (seed0, seed1))
        self.dropout_calls = 0              # training steps taken so far: the position in the mask sequence (the trainer restores it from the step count on --resume)
        self._packed: Dict[str, PackedWeight] = {}
        self._packed_sig = None
        self._ptr_sig = 0
        self._packed_gen = 0                # bumped by every repack: what derived caches (TrainEngine) compare
        self._folded_gen = -1               # the repack that last wrote the training engine's folded fp32 copies
        self._ws: Dict[Tuple, _Workspace] = {}
        self._ws_infer: Dict[Tuple, _Workspace] = {}
        self._ws_train: Dict[Tuple, _Workspace] = {}

    # ------------------------------------------------------------------ weight norm (efficient_tts.py:400-418)
    def remove_weight_norm(self):
        for m in self.modules():
            if isinstance(m, torch.nn.Conv1d) and hasattr(m, "weight_g"):
                torch.nn.utils.remove_weight_norm(m)
                logging.debug(f"Weight norm is removed from {m}.")
        self._packed_sig = None

    def apply_weight_norm(self):
        for m in self.modules():
            if isinstance(m, torch.nn.Conv1d) and not hasattr(m, "weight_g"):
                torch.nn.utils.weight_norm(m)
        self._packed_sig = None

    # ------------------------------------------------------------------ packed weights
    def _conv_modules(self):
        out = []
        for blk in ("text_encoder", "mel_encoder", "decoder"):
            for i, layer in enumerate(getattr(self, blk).layers):
                out.append((f"{blk}.{i}", layer.conv[0]))
        for i, seq in enumerate(self.duration_predictor.conv):
            out.append((f"dur.{i}", seq[0]))
        return out


    def _weights(self, folded: Optional[Dict[str, torch.Tensor]] = None,
                 wt: Optional[Dict[str, PackedWeight]] = None, params=None, phase_of=None) -> Dict[str, PackedWeight]:
        """B operand planes of every Conv1d/Linear; repacked (weight-norm fold fused) whenever
        a parameter changed (optimizer step, load_state_dict, .to()).  Training engine extras, produced by the same
        launches: `folded[name]` (fp32 [cout][cin][taps]) receives the folded weight g * v / ||v|| of a weight-normed
        conv, `wt[name]` the transposed + tap-flipped dgrad plane.  Equally shaped weights go through ONE grouped
        call (`efts_pack_weights_grouped`, a device-side item table) instead of a launch each.
        `_packed_sig` cannot see in-place updates made by the fused optimizer kernel (no version bump), which is
        why EftsAdam resets it and why consumers of derived data compare `_packed_gen`, never the signature.
        `phase_of(name, has_dgrad_plane) -> [(phase, with_dgrad_plane, with_forward_plane), ...]` (round 6, the training pass): the repack in PHASES.  Nothing is
        launched here then; `_issue_packs(phase)` enqueues a phase's grouped launches on the CURRENT stream, and the caller places the phases
        where their first consumer needs them (the training pass: the text side's planes on the text stream, the rest on the main one; a phase
        entry with with_forward_plane = False writes the dgrad plane only).  Every phase of a repack must be issued before the next one is scheduled."""
        sig = tuple((p.data_ptr(), p._version) for p in (params if params is not None else self.parameters()))
        if sig == self._packed_sig:
            return self._packed
        self._ptr_sig = hash(tuple(a for a, _ in sig))       # parameter storage identity (captured graphs hold these pointers)
        dev = self.text_embedding_table.weight.device
        pk = self._packed
        folded = folded or {}
        wt = wt or {}
        mods = [(name, conv, conv.kernel_size[0]) for name, conv in self._conv_modules()]
        lins = [("key", self.text_encoder_key), ("prenet", self.mel_prenet[0]), ("head", self.mel_output_layer)]
        if not self.share_text_encoder_key_value:
            lins.append(("value", self.text_encoder_value))
        if self.mel_query_fc is not None:
            lins.append(("qfc", self.mel_query_fc))
        mods += [(name, lin, 1) for name, lin in lins]
        groups: Dict[Tuple, list] = {}
        for name, mod, taps in mods:
            cout, cin = mod.weight_v.shape[:2] if hasattr(mod, "weight_g") else mod.weight.shape[:2]
            if name not in pk or pk[name].buf.device != dev:
                pk[name] = PackedWeight(cout, cin, taps, self.split, dev)
            for phase, with_t, with_f in ([(None, name in wt, True)] if phase_of is None else phase_of(name, name in wt)):
                groups.setdefault((cout, cin, taps, bool(with_t), phase, bool(with_f)), []).append((name, mod))
        lib = L.load()
        table_rows, launches = [], []
        for (cout, cin, taps, with_t, phase, with_f), members in groups.items():
            first = len(table_rows)
            tiled = cout % 64 == 0 and cin % 64 == 0 and taps <= 5     # the library's one-pass path: no folded copy needed
            for name, mod in members:
                if hasattr(mod, "weight_g"):
                    w, g = mod.weight_v.detach(), mod.weight_g.detach()
                    fo = None if tiled else folded.get(name)
                    if with_t and not tiled and fo is None:
                        raise L.EftsError(f"{name}: the row pack kernels need a folded fp32 copy for the dgrad plane")
                else:
                    w, g, fo = mod.weight.detach(), None, None
                assert w.is_contiguous()
                # (a NULL forward plane: the item only writes its dgrad plane -- the forward plane was packed by an earlier phase)
                table_rows.append((w.data_ptr(), 0 if g is None else g.data_ptr(), 0 if fo is None else fo.data_ptr(),
                                   pk[name].ptr if with_f else 0, wt[name].ptr if with_t else 0))
            ref = pk[members[0][0]]
            launches.append((first, len(members), ref.ld, wt[members[0][0]].ld if with_t else 0, cout, cin, taps, int(with_t), phase))
        key = tuple(table_rows)
        tables = getattr(self, "_pack_tables", None)
        if tables is None:
            tables = {}
            object.__setattr__(self, "_pack_tables", tables)
        if key not in tables:
            # Pointers are stable across steps: built once per item list.  Tables are KEPT (the eval forward and the training engine
            # alternate between two lists -- with and without the dgrad planes -- and a training step captured as a hipGraph keeps
            # launching with the table it was captured with: a replaced table would be freed memory under that graph).  Least
            # recently used ones go beyond 8: those belong to parameter storages that no longer exist (re-homed / re-created
            # parameters), which a captured step can no longer be replayed with either (its tag holds the storage signature)
            while len(tables) >= 8:
                tables.pop(next(iter(tables)))
            # (scale workspace: one region per phase -- the launches of a phase run in order on one stream, different phases may overlap)
            phases = sorted({t[8] for t in launches}, key=str)
            region = max(n * co for _, n, _, _, co, _, _, _, _ in launches)
            tables[key] = (torch.tensor(table_rows, dtype=torch.int64, device=dev), torch.empty(len(phases) * region, device=dev),
                           {ph: i * region * 4 for i, ph in enumerate(phases)})
        else:
            tables[key] = tables.pop(key)                                # most recently used last
        table, scale, region_of = tables[key]
        base = table.data_ptr()
        pending: Dict = {}
        for first, n, ld, ld_t, cout, cin, taps, with_t, phase in launches:
            pending.setdefault(phase, []).append((base + first * 40, n, scale.data_ptr() + region_of[phase], ld, ld_t, cout, cin, taps, self.split, with_t))
        object.__setattr__(self, "_pack_pending", pending)
        if phase_of is None:
            self._issue_packs(None)
        self._packed_sig = sig
        self._packed_gen += 1
        if wt:
            self._folded_gen = self._packed_gen      # this repack also wrote the training engine's derived copies
        return pk

    def _issue_packs(self, phase) -> None:
        """enqueue the grouped repack launches of `phase` (scheduled by the last `_weights` call) on the current stream"""
        lib = L.load()
        for args in self._pack_pending.pop(phase, []):
            L.check(lib.efts_pack_weights_grouped(*args, O._stream()), "efts_pack_weights_grouped")

    def _te0_table(self, pk) -> Optional[torch.Tensor]:
        """tap_table [k_size][num_symbols][C] of text-encoder layer 0: tap k's weights applied to every symbol's embedding, in the
        operand format the model runs in (k_size one-tap efts_gemm launches over the embedding rows); rebuilt when the weights were
        re-packed.  None when the look-up form does not apply."""
        if not self.embed_conv or len(self.text_encoder.layers) == 0 or self.n_channels % 32 or self._drop(0)[0] > 0.0 or self.k_size > 5 or self.act_general:
            return None                      # (a Dropout mask sits between layer 0's activation and its residual add: no look-up form)
        if getattr(self, "_te0_gen", None) == self._packed_gen and getattr(self, "_te0_tab", None) is not None:
            return self._te0_tab
        table = self.text_embedding_table.weight.detach()
        V, C, K = table.shape[0], self.n_channels, self.k_size
        dev = table.device
        rsv = Rows(1, V)
        # ONE table (and one staging plane) per (device, geometry), rebuilt IN PLACE like the packed planes: captured graphs bake the
        # table's address into their efts_embed_conv launch, so a fresh allocation per repack would leave them reading a freed or
        # stale table after any weight update; the stream orders the rebuild in front of the next replay
        geo = (dev, K, V, C, self.split)
        if getattr(self, "_te0_geo", None) != geo:
            object.__setattr__(self, "_te0_geo", geo)
            object.__setattr__(self, "_te0_tab", torch.empty(K, V, C, dtype=torch.float32, device=dev))
            object.__setattr__(self, "_te0_plane", Plane.for_rows(rsv, C, self.split, dev))
            object.__setattr__(self, "_te0_ids", torch.arange(V, device=dev)[None])
        tab, plane = self._te0_tab, self._te0_plane
        with O.stream_scope():
            O.embed(self._te0_ids, table, None, plane, rsv)
            w = pk["text_encoder.0"]
            for k in range(K):
                O.gemm(a=plane, b_ptr=w.ptr + k * w.tap_stride, ldb=w.ld, m=V, n=C, out_f32_ptr=tab[k].data_ptr(), ldo=C, tiling=L.TILING_GENERIC)
        object.__setattr__(self, "_te0_gen", self._packed_gen)
        return tab

    def _te0_ptr(self) -> int:
        """address of the tap table (part of every graph tag: a new table means a new capture)"""
        tab = getattr(self, "_te0_tab", None)
        return 0 if (tab is None or not self.embed_conv) else tab.data_ptr()

    def _embed_te0(self, ws, pk, text, rs1: Rows, lens_i32: Optional[torch.Tensor], tab: torch.Tensor):
        """embedding + text-encoder layer 0 in one gather launch -> (fp32 stream, operand plane) of layer 0's output"""
        C = self.n_channels
        x_f, x_p = ws.f32("te_f0", rs1, C), ws.plane("te_p0", rs1, C, self.split)
        O.embed_conv(text, lens_i32, self.text_embedding_table.weight.detach(), tab, self.text_encoder.layers[0].conv[0].bias, self.slope,
                     x_f, x_p, rs1)
        return x_f, x_p

    # ------------------------------------------------------------------ train-mode Dropout (counter-based masks)
    def _dropout_base(self) -> int:
        """the base seed of this process's mask family: the model's seed mixed with the data-parallel rank (the reference's ranks have
        their own torch RNG streams)"""
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:                                  # noqa: BLE001
            rank = 0
        return (int(self.dropout_seed) + 0x9E3779B1 * rank) & 0xFFFFFFFF

    def _dropout_seeds(self, calls: int):
        """(conv(k) -> seed of the Dropout behind conv / prenet launch k, seed of duration-predictor layer 0, of layer 1) for the
        `calls`-th train-mode pass of this process: ONE definition for the fused training pass (train.TrainEngine) and the
        gradient-free train()-mode forward, which therefore draw the same masks at the same counter"""
        base = self._dropout_base()

        def conv(k: int) -> int:
            return (base * 2654435761 + calls * 1000003 + k * 7919 + 12345) & 0xFFFFFFFF
        return conv, (base + 2 * calls) & 0xFFFFFFFF, (base + 2 * calls + 1) & 0xFFFFFFFF

    def _drop(self, k: int):
        """(p, seed) of the Dropout behind conv / prenet launch k in the pass in progress ((0, 0): none)"""
        st = getattr(self, "_drop_now", None)
        return (0.0, 0) if st is None or st[0] <= 0.0 else (st[0], st[1](k))

    def _side_stream(self, device) -> "torch.cuda.Stream":
        if not self.side_stream:
            return torch.cuda.current_stream(device)
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, "_side", st)
        return st

    def _aux_stream(self, device) -> "torch.cuda.Stream":
        """a third stream (the training pass: the decoder's grouped weight gradients beside the alignment block's backward)"""
        if not self.side_stream:
            return torch.cuda.current_stream(device)
        st = getattr(self, "_aux", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, "_aux", st)
        return st

    def _workspace(self, key, device, pin=()) -> _Workspace:
        """The buffers of one shape.  Bounded LRU pools (each entry holds full activations): 4 teacher-forced / training
        shapes, 64 free-running ones (one or a few utterances, a few MB each).  `pin`: workspaces of the call in progress,
        never evicted -- their buffers are about to be read (the value projection of phase 1) or captured."""
        if key[0] in ("infb", "infb2", "inf", "inf2"):
            pool, cap = self._ws_infer, 64
        elif key[0] == "train":
            # training shapes have a pool of their own: an evaluation pass over more than four shapes must not evict the workspace a
            # captured training step (step_graph.GraphedStep) points into -- that forces a recapture and keeps two multi-GB
            # workspaces alive at once
            pool, cap = self._ws_train, 4
        else:
            pool, cap = self._ws, 4
        ws = pool.pop(key, None)
        if ws is None:
            for old in list(pool):
                if len(pool) < cap:
                    break
                if not any(pool[old] is q for q in pin):
                    del pool[old]
            ws = _Workspace(device)
        pool[key] = ws                                   # most recently used last
        return ws

    # ------------------------------------------------------------------ building blocks

    def _til(self, rows: int):
        """efts_gemm tiling for a 512-column launch over `rows` rows: the small-M kernel (64 x 32 tiles, K split across the waves:
        ~4x shorter dependent chain per layer) for one-utterance row spaces of the free-running path, else the library's rules"""
        if self._free_running and self.small_m and rows <= self.SMALL_M_ROWS and self.n_channels % 32 == 0 and self.k_size <= 5:
            return L.TILING_SMALLM
        return None

    def _on_resconv(self, rs: Rows) -> bool:
        if self._drop(0)[0] > 0.0:           # conv Dropout in the pass in progress: efts_gemm's epilogue carries the masks
            return False
        return self.resconv and rs.rows >= self.RESCONV_MIN_ROWS and self.n_channels % 256 == 0 and self.k_size in (3, 5) and self.act_general is None

    def _stream_in(self, ws, tag, rs: Rows):
        """Buffers the producer of a residual stack's input writes: (fp32 stream or None, operand plane, lo plane or None).
        A stack on efts_resconv5 reads its residual from the planes, so no fp32 copy is written at all."""
        C = self.n_channels
        pl = ws.plane(f"{tag}_p", rs, C, self.split)
        if self._on_resconv(rs):
            return None, pl, (ws.plane(f"{tag}_l", rs, C, 1) if self.split == 1 else None)
        return ws.f32(f"{tag}_f", rs, C), pl, None

    def _res_layer_args(self, ws, tag, blk, pk, rs: Rows, i: int, n: int, x_f32, x_pl: Plane, x_lo, gap_ptr, last_split: int, last_f32: bool):
        """keyword set of efts_resconv5 for layer i of a stack on hi / lo planes, and the buffers it writes (`x_f32`: the fp32
        stream to take the residual from instead of the planes, or None)"""
        C = self.n_channels
        last = i == n - 1
        o_split = last_split if last else self.split
        y = ws.plane(f"{tag}_p{i & 1}", rs, C, o_split)
        y_lo = ws.plane(f"{tag}_l{i & 1}", rs, C, 1) if (o_split == 1 and not last) else None
        o_f32 = ws.f32(f"{tag}_f{i & 1}", rs, C) if (last and last_f32) else None
        kw = dict(x=x_pl, x_lo=x_lo, x_f32_ptr=None if x_f32 is None else x_f32.ptr, ldr=C, w=pk[f"{blk}.{i}"], taps=self.k_size,
                  m=rs.rows, n=C, bias=getattr(self, blk).layers[i].conv[0].bias, slope=self.slope,
                  rowmask_ptr=gap_ptr, y_f32_ptr=None if o_f32 is None else o_f32.ptr, ldo=C, y=y, y_lo=y_lo)
        return kw, o_f32, y, y_lo

    def _res_stack(self, ws, tag, blk, pk, rs: Rows, x_f32: Optional[F32Rows], x_pl: Plane, gap_ptr, last_split: int,
                   last_f32: bool, x_lo: Optional[Plane] = None, rider=None, after=None, start: int = 0):
        """n x ( x + LeakyReLU(conv1d_k5(x)) ) on the row space (efts_modules.py:48-51,77-79).
        `rider(i)`: optional, returns the efts_resconv5 keyword set of an independent layer of the same geometry that shares
        layer i's persistent launch (or None); `after(i)`: optional, called when layer i's launch has been enqueued; `start`: first
        layer to run (x is then the output of layer start - 1)."""
        n = len(getattr(self, blk).layers)
        C = self.n_channels
        if self._on_resconv(rs):
            # The stream between the layers is a pair of bf16 planes (hi = the next layer's MFMA operand, lo = the
            # remainder; split 2 planes carry both): 4 B read + 4 B written per element and layer instead of 4 + 6..8 B.
            o_f32 = None
            for i in range(start, n):
                kw, o_f32, y, y_lo = self._res_layer_args(ws, tag, blk, pk, rs, i, n, x_f32 if i == start else None, x_pl, x_lo, gap_ptr, last_split, last_f32)
                extra = rider(i) if rider is not None else None
                if extra is not None:
                    O.resconv5_multi([kw, extra])
                else:
                    O.resconv5(**kw)
                if after is not None:
                    after(i)
                x_pl, x_lo = y, y_lo
            return o_f32, x_pl
        assert rider is None and after is None
        kbase = dict(te=10, me=20, dec=30).get(tag, 50)      # launch index of the Dropout masks (same numbering as the fused training pass)
        for i in range(start, n):
            last = i == n - 1
            w = pk[f"{blk}.{i}"]
            o_split = last_split if last else self.split
            o_f32 = ws.f32(f"{tag}_f{i & 1}", rs, C) if (not last or last_f32) else None
            o_pl = ws.plane(f"{tag}_p{i & 1}", rs, C, o_split)
            dp_, dseed = self._drop(kbase + i)               # efts_modules.py:38-47: Dropout behind the activation, in front of the residual add
            if self.act_general is not None:                 # any other torch.nn activation: pre-activation in fp32, then one elementwise launch
                z = ws.f32(f"{tag}_z", rs, C)
                O.gemm(a=x_pl, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=self.k_size, m=rs.rows, n=C,
                       bias=getattr(self, blk).layers[i].conv[0].bias, out_f32_ptr=z.ptr, ldo=C, tiling=self._til(rs.rows))
                o_f32 = ws.f32(f"{tag}_f{i & 1}", rs, C)
                O.act_apply(self.act_general, z.ptr, x_f32.ptr, gap_ptr, o_f32, o_pl, rs.rows, C, dp_, dseed)
                x_f32, x_pl = o_f32, o_pl
                continue
            O.gemm(a=x_pl, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=self.k_size, m=rs.rows, n=C,
                   act=L.ACT_LEAKY, slope=self.slope, bias=getattr(self, blk).layers[i].conv[0].bias,
                   resid_ptr=x_f32.ptr, ldr=C, rowmask_ptr=gap_ptr,
                   out_f32_ptr=None if o_f32 is None else o_f32.ptr, ldo=C, out_plane=o_pl, tiling=self._til(rs.rows),
                   drop_p=dp_, drop_seed=dseed)
            x_f32, x_pl = o_f32, o_pl
        return x_f32, x_pl

    def _text_side(self, ws, pk, text, rs1: Rows, gap1, len1, on_key=None, vt: Optional[Plane] = None):
        """embed -> text encoder -> key (split-2 plane, masked), value (fp32 + plane, masked)
        (efficient_tts.py:144-157 / :246-255).  `on_key()` is called as soon as the key projection is enqueued (the q.k^T launch
        of the other stream waits for that, not for the value); `vt`: also pack V^T for the alpha'.V launch here."""
        C = self.n_channels
        tab = self._te0_table(pk)
        if tab is not None:                                    # embedding + layer 0 as table look-ups; padded ids are real symbols here
            x_f, x_p = self._embed_te0(ws, pk, text, rs1, None, tab)
            start = 1
        else:
            x_f = ws.f32("emb_f", rs1, C)
            x_p = ws.plane("emb_p", rs1, C, self.split)
            O.embed(text, self.text_embedding_table.weight.detach(), x_f, x_p, rs1)
            start = 0
        if start < len(self.text_encoder.layers):
            _, h_p = self._res_stack(ws, "te", "text_encoder", pk, rs1, x_f, x_p, gap1.data_ptr(), self.split, False, start=start)
        else:
            h_p = x_p
        key_p = self._key_proj(ws, pk, rs1, h_p, gap1, len1)
        if on_key is not None:
            on_key()
        val_f, val_p = self._value_proj(ws, pk, rs1, h_p, gap1, len1, vt)
        return key_p, val_f, val_p

    def _key_proj(self, ws, pk, rs1: Rows, h_p: Plane, gap1, len1) -> Plane:
        """text_encoder_key, zero at padded text (efficient_tts.py:149, :155-156): split-2 operand plane of q.k^T"""
        C = self.n_channels
        key_p = ws.plane("key_p", rs1, C, 2)
        wk = pk["key"]
        O.gemm(a=h_p, b_ptr=wk.ptr, ldb=wk.ld, m=rs1.rows, n=C, bias=self.text_encoder_key.bias,
               rowmask_ptr=(gap1 if len1 is None else len1).data_ptr(), out_plane=key_p, tiling=self._til(rs1.rows))
        return key_p

    def _value_proj(self, ws, pk, rs1: Rows, h_p: Plane, gap1, len1, vt: Optional[Plane] = None):
        """text_encoder_value (the key projection again when shared, :150-153), zero at padded text (:157): fp32 + operand plane"""
        C = self.n_channels
        val_f = ws.f32("val_f", rs1, C)
        val_p = ws.plane("val_p", rs1, C, self.split)
        shared = self.share_text_encoder_key_value
        wv = pk["key"] if shared else pk["value"]
        vbias = self.text_encoder_key.bias if shared else self.text_encoder_value.bias
        O.gemm(a=h_p, b_ptr=wv.ptr, ldb=wv.ld, m=rs1.rows, n=C, bias=vbias,
               rowmask_ptr=(gap1 if len1 is None else len1).data_ptr(), out_f32_ptr=val_f.ptr, ldo=C, out_plane=val_p, tiling=self._til(rs1.rows))
        if vt is not None:
            O.pack_vt(val_f, vt, rs1.B, rs1.T, rs1.Tp, C)
        return val_f, val_p

    def _duration(self, ws, pk, rs1: Rows, val_p: Plane, gap1, out_mask_ptr, mode: int) -> torch.Tensor:
        """DurationPredictor._forward (duration_predictor.py:66-88); its Dropout(0.1) behind each LayerNorm (:61) is applied when the
        pass in progress is a train()-mode one (`_drop_now`), else identity."""
        C = self.n_channels
        dp = self.duration_predictor
        h_f = ws.f32("dur_f", rs1, C)
        x_p = val_p
        for i, seq in enumerate(dp.conv):
            w = pk[f"dur.{i}"]
            O.gemm(a=x_p, b_ptr=w.ptr, ldb=w.ld, b_tap_stride=w.tap_stride, taps=3, m=rs1.rows, n=C, act=L.ACT_RELU,
                   bias=seq[0].bias, out_f32_ptr=h_f.ptr, ldo=C, tiling=self._til(rs1.rows))
            x_p = self._duration_norm(ws, rs1, i, gap1, out_mask_ptr, mode)
        return ws.tensor("dur_out", (rs1.rows,))

    def _duration_norm(self, ws, rs1: Rows, i: int, gap1, out_mask_ptr, mode: int) -> Optional[Plane]:
        """LayerNorm behind conv i (duration_predictor.py:58-61): -> the next conv's operand plane, or (last layer) + Linear(C, 1) ->
        the predicted log-durations in ws "dur_out" """
        C = self.n_channels
        dp = self.duration_predictor
        ln = dp.conv[i][2]
        h_f = ws.f32("dur_f", rs1, C)
        st = getattr(self, "_drop_now", None)
        dur_p, dur_seed = (0.0, 0) if st is None else (st[2], st[3][min(i, 1)] if i < 2 else (st[3][1] + i) & 0xFFFFFFFF)
        if i + 1 < len(dp.conv):
            x_p = ws.plane(f"dur_p{i}", rs1, C, self.split)
            O.layernorm_rows(h_f.ptr, ln.weight.detach(), ln.bias.detach(), ln.eps, gap1.data_ptr(), None, x_p, rs1.rows, C, dur_p, dur_seed)
            return x_p
        O.layernorm_dot(h_f.ptr, ln.weight.detach(), ln.bias.detach(), ln.eps, dp.linear.weight.detach(), dp.linear.bias.detach(),
                        out_mask_ptr, mode, float(dp.offset), ws.tensor("dur_out", (rs1.rows,)), rs1.rows, C, dur_p, dur_seed)
        return None

    def _fused_expand(self, T1: int) -> bool:
        return self.fuse_expand and T1 <= 256 and self.n_channels % 128 == 0

    def _expand_decode(self, ws, pk, B, T1, rs1: Rows, rs2: Rows, val_f: F32Rows, e, tl, ml, ralpha, len2_ptr, gap2,
                       vt: Optional[Plane] = None, rider=None, after=None, loss_target: Optional[torch.Tensor] = None):
        """Gaussian re-alignment from e, bmm(V^T, alpha') -> decoder -> mel head (efficient_tts.py:184-200 / :270-284).
        `ralpha` [B, T1, T2] receives alpha' (the API tensor); tl / ml: int32 lengths or None (no masks, :270-274);
        `vt`: V^T already packed (unfused path only)."""
        C = self.n_channels
        h_f, h_p, h_l = self._stream_in(ws, "exp", rs2)
        if self._fused_expand(T1):
            # alpha' is the MFMA A operand, produced in registers; V is read as fp32: no alpha'^T plane, no V^T planes
            O.expand(e=e, tl=tl, ml=ml, sigma=float(self.sigma), v=val_f, rs1=rs1, rs2=rs2, alpha_out=ralpha, y_f32=h_f, y=h_p, y_lo=h_l)
        else:
            ra_plane = ws.plane("ra_p", rs2, T1, 2)
            O.reconst_alpha(e, tl, ml, float(self.sigma), ralpha, ra_plane, B, T1, rs2.T, rs2.Tp)
            if vt is None:
                vt = ws.raw_plane("vt", B * C, T1, 2)
                O.pack_vt(val_f, vt, B, T1, rs1.Tp, C)
            O.gemm(a=ra_plane, b_ptr=vt.ptr, ldb=vt.ld, m=rs2.T, n=C, batch=B, a_batch_stride=rs2.Tp * ra_plane.ld,
                   b_batch_stride=C * vt.ld, rowmask_ptr=len2_ptr, rowmask_batch_stride=rs2.Tp,
                   out_f32_ptr=None if h_f is None else h_f.ptr, ldo=C, out_batch_stride=rs2.Tp * C, out_plane=h_p,
                   outb_batch_stride=rs2.Tp * h_p.ld, out_plane_lo=h_l)
        _, d_p = self._res_stack(ws, "dec", "decoder", pk, rs2, h_f, h_p, gap2.data_ptr(), self.split, False, x_lo=h_l, rider=rider, after=after)
        # mel head (:198-200), written straight into the [B, T2, odim] tensor the caller gets (one item per batch entry of the
        # launch: no row-space copy of mel_pred, no clone)
        mel = torch.empty(B, rs2.T, self.odim, dtype=torch.float32, device=h_p.buf.device)
        wh = pk["head"]
        loss_kw = {}
        if loss_target is not None and self.fuse_mel_loss and self.use_masking and len2_ptr is not None and self.odim % 4 == 0 and self.odim <= 128:
            # the mel term of the loss (:220, fastspeech_loss.py:54-67) in the head's epilogue: the tile is in LDS, the target frames arrive
            # where a residual would, the row mask IS the loss mask -- one launch less at the end of every forward.  Only where AUTO picks the
            # generic tiling anyway (one workgroup per CU or more): a short launch keeps its 64-column tiles and the separate loss launch
            mt = (rs2.T + 127) // 128
            if B * mt >= (self.fuse_mel_loss_min_wgs or torch.cuda.get_device_properties(mel.device).multi_processor_count):
                part = ws.tensor("sqerr_part", (B * mt * 4,))
                loss_kw = dict(sqerr_target=loss_target, sqerr_part=part)
                object.__setattr__(self, "_sqerr_parts", (part, B * mt * 4))
        O.gemm(a=d_p, b_ptr=wh.ptr, ldb=wh.ld, m=rs2.T, n=self.odim, batch=B, a_batch_stride=rs2.Tp * d_p.ld, bias=self.mel_output_layer.bias,
               rowmask_ptr=len2_ptr if len2_ptr is not None else gap2.data_ptr(), rowmask_batch_stride=rs2.Tp,
               out_f32_ptr=mel.data_ptr(), ldo=self.odim, out_batch_stride=rs2.T * self.odim, **loss_kw)
        return mel

    def _require(self, t: torch.Tensor):
        if not t.is_cuda:
            raise L.EftsError("EfficientTTSCNN runs on an MI355X only: inputs/parameters must be on a ROCm device "
                              "(there is no CPU fallback)")
        L.require_device()

    # ------------------------------------------------------------------ forward (efficient_tts.py:120-228)
    def forward(self, text: torch.Tensor, text_lengths: torch.Tensor, speech: torch.Tensor,
                speech_lengths: torch.Tensor):
        """Teacher-forced forward.  Returns (loss, stats, imv[B,T2], reconst_alpha[B,T1,T2],
        mel_pred[B,T2,odim], speech) exactly like the reference (:228)."""
        training_path = torch.is_grad_enabled() and (self.text_embedding_table.weight.requires_grad or any(p.requires_grad for p in self.parameters()))
        self._require(text)
        if training_path:
            from .autograd import training_forward
            return training_forward(self, text, text_lengths, speech, speech_lengths)
        if self.training:
            # train() mode without gradients (a validation pass that forgot eval(), a probe): the reference applies its Dropouts here --
            # the duration predictor's 0.1 always (duration_predictor.py:61), ResConv1d's and the prenet's when dropout_rate > 0
            # (efts_modules.py:38-47, efficient_tts.py:76-80).  Same counter-based masks as the fused training pass would draw for the
            # next step (`_forward_impl`: `dropout_calls` itself is left alone); the seeds are by-value launch arguments, so no graph
            return self._forward_impl(text, text_lengths, speech, speech_lengths)[0]
        if not self.graphs or torch.cuda.is_current_stream_capturing():
            return self._forward_impl(text, text_lengths, speech, speech_lengths)[0]
        # per-shape hipGraph: the launches of this shape are replayed as one graph (efficient_tts_amd/graphs.py)
        pk = self._weights()                                  # (re)packing stays outside the graph
        self._te0_table(pk)                                   # ... and so does the tap table of text-encoder layer 0
        dev = text.device
        key = ("fwd", tuple(text.shape), tuple(speech.shape), text.dtype, speech.dtype, text_lengths.dtype, speech_lengths.dtype)
        ws = self._workspace(("fwd", text.shape[0], text.shape[1], speech.shape[1]), dev)
        # the graph is valid while the buffers its launches point at live: this workspace, the packed planes, the parameters
        tag = (ws.serial, self._ptr_sig, tuple(w.ptr for w in pk.values()), self.opt.tag(), self._te0_ptr())

        def body(t, tl, sp, sl):
            (_, stats, imv, ralpha, mel_pred, _), _ = self._forward_impl(t, tl, sp, sl)
            return stats._t, imv, ralpha, mel_pred

        out3, imv, ralpha, mel_pred = self._graph_cache.run(key, tag, (text, text_lengths.to(dev), speech, speech_lengths.to(dev)), body, keepalive=ws,
                                                            adaptive=self.graph_policy == "auto")
        return out3[0], LazyStats(out3), imv, ralpha, mel_pred, speech

    def _forward_impl(self, text, text_lengths, speech, speech_lengths, keep: bool = False):
        prev = getattr(self, "_drop_now", None)
        if self.training:
            # a gradient-free pass in train() mode (a probe, a validation pass that forgot eval()) does NOT advance `dropout_calls`: that counter
            # sequences the masks of the optimisation steps (GraphedStep's step words, the trainer's --resume), and a resumed run must
            # reproduce the uninterrupted one whatever was probed in between.  Probes draw the masks of the step that comes next, then of
            # the ones behind it, from a counter of their own that restarts whenever a real step has moved the sequence on
            base = int(self.dropout_calls)
            if getattr(self, "_probe_base", None) != base:
                object.__setattr__(self, "_probe_base", base)
                object.__setattr__(self, "_probe_calls", 0)
            conv, s0, s1 = self._dropout_seeds(base + 1 + self._probe_calls)
            object.__setattr__(self, "_probe_calls", self._probe_calls + 1)
            conv_p = float(self.dropout_rate) if self.dropout_rate >= 1e-5 else 0.0
            object.__setattr__(self, "_drop_now", (conv_p, conv, float(self.duration_predictor.conv[0][3].p), (s0, s1)))
        else:
            object.__setattr__(self, "_drop_now", None)
        try:
            with O.stream_scope():
                return self._forward_body(text, text_lengths, speech, speech_lengths, keep)
        finally:
            object.__setattr__(self, "_drop_now", prev)

    def _forward_body(self, text, text_lengths, speech, speech_lengths, keep: bool = False):
        dev = text.device
        B, T1 = text.shape
        T2 = speech.shape[1]
        C = self.n_channels
        text = text.contiguous()
        speech = speech.contiguous().float()
        pk = self._weights()
        ws = self._workspace(("fwd", B, T1, T2), dev)
        rs1, rs2 = Rows(B, T1, self.row_gap), Rows(B, T2, self.row_gap)
        gap1, len1 = ws.tensor("gap1", (rs1.rows,)), ws.tensor("len1", (rs1.rows,))
        gap2, len2 = ws.tensor("gap2", (rs2.rows,)), ws.tensor("len2", (rs2.rows,))
        tl_d, ml_d = text_lengths.to(dev), speech_lengths.to(dev)
        masks_done = tl_d.dtype == ml_d.dtype and tl_d.dtype in (torch.int64, torch.int32) and tl_d.is_contiguous() and ml_d.is_contiguous()
        if masks_done:
            # :137-139 for both row spaces + the int32 copies of the lengths in ONE launch (two casts + two mask launches before)
            tl, ml = O.row_masks_pair(tl_d, ml_d, rs1, rs2, gap1, len1, gap2, len2)
        else:
            tl, ml = tl_d.to(torch.int32), ml_d.to(torch.int32)
        # The text side (masks, embed, 5 convs, K/V) and the duration predictor do not depend on the mel side (prenet, 3 convs).
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)
        if not masks_done:
            O.row_masks(ml, rs2, gap2, len2)                                      # :139 (in FRONT of the fork: the prenet on the side stream reads gap2)
        side.wait_stream(main)
        vt = None if self._fused_expand(T1) else ws.raw_plane("vt", B * C, T1, 2)
        nt, nm = len(self.text_encoder.layers), len(self.mel_encoder.layers)
        merged = self._on_resconv(rs2) and nt >= 1 and nm >= 1
        v_ready = torch.cuda.Event()

        def prenet(max_wgs=0):                                                    # :161
            pre_f, pre_p, pre_l = self._stream_in(ws, "pre", rs2)
            wp = pk["prenet"]
            pre_dp, pre_seed = self._drop(40)                                      # mel_prenet's Dropout (:76-80)
            if self.act_general is not None:
                mel_in = ws.plane("mel_in", rs2, self.odim, self.split)
                O.pack_rows(speech, None, mel_in, rs2)
                z = ws.f32("pre_z", rs2, C)
                O.gemm(a=mel_in, b_ptr=wp.ptr, ldb=wp.ld, m=rs2.rows, n=C, bias=self.mel_prenet[0].bias, out_f32_ptr=z.ptr, ldo=C)
                O.act_apply(self.act_general, z.ptr, None, gap2.data_ptr(), pre_f, pre_p, rs2.rows, C, pre_dp, pre_seed)
            elif pre_dp == 0.0 and self.fuse_prenet and self.odim % 8 == 0 and self.odim <= 128 and C % 128 == 0:
                # straight from the caller's fp32 frames: no operand plane of the mel input, one launch (bit-identical on every frame)
                O.frame_linear(x=speech, w=wp, bias=self.mel_prenet[0].bias, act=L.ACT_LEAKY, slope=self.slope, rs=rs2,
                               y=pre_p, y_lo=pre_l, y_f32=pre_f, max_workgroups=max_wgs)
            else:
                mel_in = ws.plane("mel_in", rs2, self.odim, self.split)
                O.pack_rows(speech, None, mel_in, rs2)
                O.gemm(a=mel_in, b_ptr=wp.ptr, ldb=wp.ld, m=rs2.rows, n=C, act=L.ACT_LEAKY, slope=self.slope,
                       bias=self.mel_prenet[0].bias, rowmask_ptr=gap2.data_ptr(), out_f32_ptr=None if pre_f is None else pre_f.ptr,
                       ldo=C, out_plane=pre_p, out_plane_lo=pre_l, drop_p=pre_dp, drop_seed=pre_seed)
            return pre_f, pre_p, pre_l

        def mel_stack(pre_f, pre_p, pre_l, rider=None):                            # :162-164
            if self.mel_query_fc is None:
                _, q_p = self._res_stack(ws, "me", "mel_encoder", pk, rs2, pre_f, pre_p, gap2.data_ptr(), 2, False, x_lo=pre_l, rider=rider)
                return q_p
            _, mh_p = self._res_stack(ws, "me", "mel_encoder", pk, rs2, pre_f, pre_p, gap2.data_ptr(), self.split, False, x_lo=pre_l, rider=rider)
            q_p = ws.plane("q_p", rs2, C, 2)
            wq = pk["qfc"]
            O.gemm(a=mh_p, b_ptr=wq.ptr, ldb=wq.ld, m=rs2.rows, n=C, bias=self.mel_query_fc.bias, rowmask_ptr=gap2.data_ptr(), out_plane=q_p)
            return q_p

        if merged:
            # The last min(nt, nm) text-encoder layers ride in the persistent launches of the mel-encoder layers (one grouped
            # efts_resconv5_multi launch per pair: the text rows are scheduled behind the mel rows at the long launch's efficiency).
            # A persistent launch owns every CU's LDS, so a text-length launch of its own beside it would get the 4 spare CUs:
            # only the first nt - nm text layers run by themselves (efts_resconv5 on the short row space), beside the HBM-bound
            # prenet on the second stream.
            nr = min(nt - (1 if self._te0_table(pk) is not None else 0), nm)       # (layer 0 may be table look-ups: it never rides)
            ns = nt - nr                                                           # text layers that run by themselves first
            pre_ready, te_done = torch.cuda.Event(), torch.cuda.Event()
            # The first text layers and the prenet share the chip by halves: both kinds of workgroup take a whole CU (LDS), the
            # prenet is bound by HBM -- which half the CUs saturate -- and a text-length layer by streaming its weights, so the
            # prenet's grid is capped at half the CUs and the text layers are scheduled onto the other half.
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            share = ns > (1 if self._te0_table(pk) is not None else 0) and cus >= 64
            with O.on_stream(side):
                pre = prenet(cus // 2 if share else 0)                             # (r6: 128 / 144 / 160 / 176 CUs for the prenet measured: no difference)
                pre_ready.record(side)
            te_plan = O.resconv5_plan_buf(rs1.rows, C, cus // 2 - 2) if share else None
            if not masks_done:
                O.row_masks(tl, rs1, gap1, len1)                                  # :137
            tab = self._te0_table(pk)
            if tab is not None:                                                   # :144 + layer 0 of :148 as table look-ups
                x_f, x_p = self._embed_te0(ws, pk, text, rs1, None, tab)
                first = 1
            else:
                x_f = ws.f32("emb_f", rs1, C)
                x_p = ws.plane("emb_p", rs1, C, self.split)
                O.embed(text, self.text_embedding_table.weight.detach(), x_f, x_p, rs1)      # :144
                first = 0
            tstate = dict(x_f=x_f, x_p=x_p, x_lo=None)

            def text_layer(i):                                                    # efts_resconv5 keyword set of text layer i (:148)
                kw, _, y, y_lo = self._res_layer_args(ws, "te", "text_encoder", pk, rs1, i, nt, tstate["x_f"], tstate["x_p"], tstate["x_lo"],
                                                      gap1.data_ptr(), self.split, False)
                tstate.update(x_f=None, x_p=y, x_lo=y_lo)
                return kw

            for i in range(first, ns):
                O.resconv5(plan=te_plan, **text_layer(i))
            main.wait_event(pre_ready)
            q_p = mel_stack(*pre, rider=lambda i: text_layer(ns + i - (nm - nr)) if i >= nm - nr else None)
            te_done.record(main)
            key_p = self._key_proj(ws, pk, rs1, tstate["x_p"], gap1, len1)          # :149, :155-156 (q.k^T is next on this stream)
            side.wait_event(te_done)
            with O.on_stream(side):                                               # the value projection beside the key projection, then the
                val_f, val_p = self._value_proj(ws, pk, rs1, tstate["x_p"], gap1, len1, vt)   # :150-157   duration predictor (:219;
                v_ready.record(side)                                              # needed by the loss only).  (Its k3 convolutions as riders of
                dur = self._duration(ws, pk, rs1, val_p, gap1, len1.data_ptr(), 0)    # decoder launches: measured slower, 1.665 vs 1.633 ms, r3.)
        else:
            # second HIP stream: the text-side launches fill the tail rounds of the mel-length kernels
            k_ready = torch.cuda.Event()
            with O.on_stream(side):
                if not masks_done:
                    O.row_masks(tl, rs1, gap1, len1)                              # :137
                key_p, val_f, val_p = self._text_side(ws, pk, text, rs1, gap1, len1, on_key=lambda: k_ready.record(side), vt=vt)  # :144-157
                v_ready.record(side)
                dur = self._duration(ws, pk, rs1, val_p, gap1, len1.data_ptr(), 0)    # :219
            q_p = mel_stack(*prenet())
            main.wait_event(k_ready)

        sidx = ws.tensor("sidx", (B, T2))
        imv = torch.empty(B, T2, dtype=torch.float32, device=dev)              # returned to the caller: written in place, no copy
        alpha = ws.tensor("alpha", (B, T1, T2)) if keep else None
        if T1 <= 128 and not keep and self.fuse_soft_index:
            # :390-398, :168, :312 in one launch: a 128-column tile holds whole score rows, so the softmax over the keys and its
            # expected index are taken from the staged tile and the 4 B T2 T1 bytes of scores are neither written nor re-read
            O.gemm(a=q_p, b_ptr=key_p.ptr, ldb=key_p.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q_p.ld,
                   b_batch_stride=rs1.Tp * key_p.ld, alpha=O.INV_SQRT(C), soft_index=sidx, key_len=tl, query_len=ml)
        else:
            scores = ws.tensor("scores", (B, T2, T1))                             # :390 q.k/sqrt(D)
            O.gemm(a=q_p, b_ptr=key_p.ptr, ldb=key_p.ld, m=T2, n=T1, batch=B, a_batch_stride=rs2.Tp * q_p.ld,
                   b_batch_stride=rs1.Tp * key_p.ld, alpha=O.INV_SQRT(C), out_f32_ptr=scores.data_ptr(), ldo=T1,
                   out_batch_stride=T2 * T1)
            O.attn_soft_index(scores, T1, tl, ml, sidx, alpha, B, T1, T2)         # :391-398, :168, :312
        e, lde = ws.tensor("e", (B, T1)), ws.tensor("lde", (B, T1))
        if self.fuse_align and O.imv_align_fits(T1, T2):
            O.imv_align(sidx, tl, ml, float(self.sigma_e), float(self.duration_offset), self.delta_e_method_1, imv, e, lde, B, T1, T2)   # :314-345, :203-216
        else:
            O.imv_scan(sidx, tl, ml, imv, B, T2)                                  # :314-323
            O.aligned_positions(imv, tl, ml, float(self.sigma_e), float(self.duration_offset), e, lde if self.delta_e_method_1 else None,
                                B, T1, T2)                                        # :178-180, :203-216
            if not self.delta_e_method_1:
                O.duration_target(e, tl, ml, float(self.duration_offset), False, lde, B, T1)   # :205-213
        ralpha = torch.empty(B, T1, T2, dtype=torch.float32, device=dev)

        main.wait_event(v_ready)
        object.__setattr__(self, "_sqerr_parts", None)
        mel = self._expand_decode(ws, pk, B, T1, rs1, rs2, val_f, e, tl, ml, ralpha, len2.data_ptr(), gap2, vt=vt,
                                  loss_target=speech)                                                        # :184-200
        main.wait_stream(side)                                                     # duration predictor done

        out3 = torch.empty(3, dtype=torch.float32, device=dev)                     # :220-227
        # use_masking=False (fastspeech_loss.py:63-67): plain means over the padded tensors = the masked sums taken with full lengths
        # (mel_pred, dur_pred and log_delta_e are zero beyond each item's length, speech is whatever the caller padded with)
        ml_loss = ml if self.use_masking else torch.full_like(ml, T2)
        tl_loss = tl if self.use_masking else torch.full_like(tl, T1)
        if self._sqerr_parts is not None:                                          # the mel head left the squared-error partial sums
            O.losses_from_parts(self._sqerr_parts[0], self._sqerr_parts[1], ml_loss, dur, lde, tl_loss, out3, B, T1, rs1.Tp, T2, self.odim)
        else:
            O.masked_losses(mel.data_ptr(), self.odim, speech, ml_loss, dur, lde, tl_loss, out3, ws.tensor("loss_ws", (1024,)), B, T1,
                            rs1.Tp, T2, T2, self.odim)
        mel_pred = mel
        ret = (out3[0], LazyStats(out3), imv, ralpha, mel_pred, speech)
        extra = dict(e=e, log_delta_e=lde, dur_pred=dur.view(B, rs1.Tp)[:, :T1], ws=ws) if keep else None
        return ret, extra

    # ------------------------------------------------------------------ inference (efficient_tts.py:230-285)
    @torch.no_grad()
    def inference(self, text: torch.Tensor, text_lengths: torch.Tensor = None):
        """Free-running synthesis of ONE utterance: returns (mel_pred[1,T2,odim], reconst_alpha[1,T1,T2])."""
        prev, self._free_running = self._free_running, True
        try:
            return self._inference_impl(text, text_lengths)
        finally:
            self._free_running = prev

    def _inference_impl(self, text: torch.Tensor, text_lengths: torch.Tensor = None):
        self._require(text)
        if text.shape[0] != 1:
            raise ValueError("inference() takes one utterance, like the reference (efficient_tts.py:361); "
                             "use inference_batch() for B > 1")
        if self.graphs and not torch.cuda.is_current_stream_capturing():
            # same arithmetic as a ragged batch of one (every layer masked by the length; equal to the unmasked B = 1 pass
            # below to ~1e-4), on bucketed shapes whose launches replay as two hipGraphs around the one host sync
            # (text_lengths is accepted and ignored, like the reference's inference(): efficient_tts.py:233, :243 -- every position
            # of `text` is synthesised; ragged batches go through inference_batch())
            key = (text.device, text.shape[1])
            tl1 = self._len1.get(key)
            if tl1 is None:                                  # one int32 [T1] per length: no fill + cast launches per call
                if len(self._len1) > 4096:
                    self._len1.clear()
                tl1 = self._len1[key] = torch.full((1,), text.shape[1], dtype=torch.int32, device=text.device)
            mel, _, ralpha = self._inference_batch_impl(text, None, None, tl_i32=tl1, want_lengths=False)
            return mel, ralpha
        dev = text.device
        T1, C = text.shape[1], self.n_channels
        pk = self._weights()
        ws = self._workspace(("inf", 1, T1), dev)
        rs1 = Rows(1, T1, self.row_gap)
        full = torch.full((1,), T1, dtype=torch.int32, device=dev)
        gap1 = ws.tensor("gap1", (rs1.rows,))
        O.row_masks(full, rs1, gap1, None)
        _, val_f, val_p = self._text_side(ws, pk, text.contiguous(), rs1, gap1, None)         # :246-255
        delta = self._duration(ws, pk, rs1, val_p, gap1, gap1.data_ptr(), 1)      # :258  clamp(exp(.)-offset, 0)
        e = ws.tensor("e", (1, T1))
        O.cumsum_rows(delta[:T1], e, 1, T1)                                       # :260
        t2 = int(torch.round(e[0, -1]).item())                                    # :361 (host sync, as the reference)
        if not self.delta_e_method_1:                                             # :261-265 + trim_e (:362-363): positions start at 0
            e = e - delta[:T1].view(1, T1)
        if t2 <= 0:
            raise ValueError("predicted total duration rounds to 0 frames")
        rs2 = Rows(1, t2, self.row_gap)
        ws2 = self._workspace(("inf2", 1, T1, t2), dev, pin=(ws,))
        gap2 = ws2.tensor("gap2", (rs2.rows,))
        O.row_masks(torch.full((1,), t2, dtype=torch.int32, device=dev), rs2, gap2, None)
        ralpha = torch.empty(1, T1, t2, dtype=torch.float32, device=dev)
        mel = self._expand_decode(ws2, pk, 1, T1, rs1, rs2, val_f, e, None, None, ralpha, None, gap2)   # :270-284
        return mel, ralpha

    # ------------------------------------------------------------------ batched ragged inference (extension)
    T1_BUCKET, T2_BUCKET = 16, 64       # free-running inference runs on shapes rounded up to these multiples (graph / workspace reuse)

    def _infer_text(self, ws, text, tl, force_delta):
        """phase 1 (no host sync): embed -> text encoder -> value -> duration predictor -> aligned positions e = cumsum(durations)
        and the mel length of every item (efficient_tts.py:246-260).  text [B, T1] int64 (positions >= tl are ignored), tl int32 [B]."""
        dev = text.device
        B, T1 = text.shape
        C = self.n_channels
        pk = self._weights()
        rs1 = Rows(B, T1, self.row_gap)
        gap1, len1 = ws.tensor("gap1", (rs1.rows,)), ws.tensor("len1", (rs1.rows,))
        O.row_masks(tl, rs1, gap1, len1)
        # embedding with padded positions zeroed, then every layer masked by the item length
        tab = self._te0_table(pk)
        if tab is not None:                                    # embedding + layer 0 as table look-ups, zero beyond each item's length
            x_f, x_p = self._embed_te0(ws, pk, text.contiguous(), rs1, tl, tab)
            start = 1
        else:
            e_f = ws.f32("emb_raw", rs1, C)
            O.embed(text.contiguous(), self.text_embedding_table.weight.detach(), e_f, None, rs1)
            x_f, x_p = ws.f32("emb_f", rs1, C), ws.plane("emb_p", rs1, C, self.split)
            O.mask_rows(e_f.ptr, len1.data_ptr(), x_f, x_p, rs1.rows, C)
            start = 0
        if start < len(self.text_encoder.layers):
            _, h_p = self._res_stack(ws, "te", "text_encoder", pk, rs1, x_f, x_p, len1.data_ptr(), self.split, False, start=start)
        else:
            h_p = x_p
        val_f, val_p = ws.f32("val_f", rs1, C), ws.plane("val_p", rs1, C, self.split)
        shared = self.share_text_encoder_key_value            # (:252-253)
        wv = pk["key"] if shared else pk["value"]
        O.gemm(a=h_p, b_ptr=wv.ptr, ldb=wv.ld, m=rs1.rows, n=C, bias=(self.text_encoder_key if shared else self.text_encoder_value).bias,
               rowmask_ptr=len1.data_ptr(), out_f32_ptr=val_f.ptr, ldo=C, out_plane=val_p, tiling=self._til(rs1.rows))
        delta = self._duration(ws, pk, rs1, val_p, len1, len1.data_ptr(), 1)          # zero beyond each length
        # durations -> positions e = cumsum, mel lengths round(e[len - 1]) (:260, :270), positions from 0 for method 2 (:261-265)
        e = torch.empty(B, T1, dtype=torch.float32, device=dev)
        ml = torch.empty(B, dtype=torch.int32, device=dev)
        O.duration_positions(delta, rs1.Tp, tl, force_delta, self.delta_e_method_1, e, ml, B, T1)
        return e, ml

    def _infer_mel(self, ws, ws2, e, tl, ml, T2: int):
        """phase 2: Gaussian re-alignment from e, expand, decoder, mel head on a [B, T2] row space (efficient_tts.py:270-284);
        reads the value projection phase 1 left in `ws` (the (B, T1) workspace the caller holds: never looked up again by key,
        a pool eviction in between would hand back a fresh zero-filled one)."""
        dev = e.device
        B, T1 = e.shape
        C = self.n_channels
        pk = self._weights()
        rs1, rs2 = Rows(B, T1, self.row_gap), Rows(B, T2, self.row_gap)
        val_f = ws.f32("val_f", rs1, C)
        gap2, len2 = ws2.tensor("gap2", (rs2.rows,)), ws2.tensor("len2", (rs2.rows,))
        O.row_masks(ml, rs2, gap2, len2)
        ralpha = torch.empty(B, T1, T2, dtype=torch.float32, device=dev)
        mel = self._expand_decode(ws2, pk, B, T1, rs1, rs2, val_f, e, tl, ml, ralpha, len2.data_ptr(), len2)
        return mel, ralpha

    @torch.no_grad()
    def inference_batch(self, text: torch.Tensor, text_lengths: torch.Tensor, force_delta: Optional[float] = None):
        """Free-running synthesis of B utterances at once -- an extension the reference cannot do
        (its inference() is B == 1 only: efficient_tts.py:361).  Every item is computed exactly as if it
        were alone: positions beyond an item's own length are kept at zero after EVERY layer (true
        zero padding, the opposite of the teacher-forced forward's leakage semantics), durations are
        accumulated per item and each item gets its own mel length T2_b = round(sum of durations).

        Returns (mel_pred [B, max T2_b, odim] zero-padded, mel_lengths [B] int64, reconst_alpha
        [B, T1, max T2_b]).  One host sync (max T2_b), like the reference's single `.item()`.

        With `graphs` on, the two phases run on bucketed shapes (T1 up to a multiple of 16, T2 of 64: padding is masked
        like any other ragged tail) and replay per-shape hipGraphs from the second call of a shape on.

        force_delta (benchmark hook): the duration predictor still runs, but every
        valid phoneme then gets this many frames, so a synthetic batch yields a known, equal T2."""
        prev, self._free_running = self._free_running, True
        try:
            return self._inference_batch_impl(text, text_lengths, force_delta)
        finally:
            self._free_running = prev

    def _inference_batch_impl(self, text, text_lengths, force_delta, tl_i32: Optional[torch.Tensor] = None, want_lengths: bool = True):
        """tl_i32: the lengths already as an int32 device tensor (inference() keeps one per length); want_lengths=False: the caller
        drops the mel lengths (inference()), so they are not converted"""
        self._require(text)
        with O.stream_scope():
            dev = text.device
            B, T1 = text.shape
            tl = tl_i32 if tl_i32 is not None else text_lengths.to(device=dev, dtype=torch.int32)
            graphs = self.graphs and not torch.cuda.is_current_stream_capturing()
            T1b = roundup(T1, self.T1_BUCKET) if graphs else T1
            pk = self._weights()
            self._te0_table(pk)                               # (built outside the graphs, like the packed planes)
            wsig = (self._ptr_sig, tuple(w.ptr for w in pk.values()), self.opt.tag(), self._te0_ptr())
            ws = self._workspace(("infb", B, T1b), dev)
            if graphs:
                def phase1(t, l):
                    with O.stream_scope():              # resolved INSIDE the capture: the launches must go to the capturing stream
                        return self._infer_text(ws, t, l, force_delta) + (l,)
                # The graph's static outputs -- and its static copy of the lengths -- are handed on as they are: phase 2 reads them in
                # place (`refs`: no copies into inputs of its own; a re-captured phase 1 means new buffers, hence a new phase-2 capture).
                # The ids go straight into the zero-filled bucket-wide static input (positions beyond an item's length are ignored).
                ids = PadTo(text, (B, T1b)) if T1b != T1 else text.contiguous()
                e, ml, tl = self._infer_cache.run(("text", B, T1b, force_delta), (ws.serial, wsig), (ids, tl), phase1, keepalive=ws, clone=False)
            else:
                e, ml = self._infer_text(ws, text, tl, force_delta)
            t2 = int((ml if B == 1 else ml.max()).item())                                  # the one host sync
            if t2 <= 0:
                raise ValueError("predicted total durations round to 0 frames")
            T2b = roundup(t2, self.T2_BUCKET) if graphs else t2
            ws2 = self._workspace(("infb2", B, T1b, T2b), dev, pin=(ws,))
            if graphs:
                def phase2(e_, l, m):
                    with O.stream_scope():
                        return self._infer_mel(ws, ws2, e_, l, m, T2b)
                trim = T2b != t2 or T1b != T1
                mel, ralpha = self._infer_cache.run(("mel", B, T1b, T2b), (ws.serial, ws2.serial, wsig), (), phase2, keepalive=(ws, ws2),
                                                    refs=(e, tl, ml), clone=not trim)
                if trim:                                       # (the trimmed copies are the fresh tensors the caller gets)
                    # .clone(), not .contiguous(): at B == 1 (or t2 == T2b) the slice of the graph's static output is already
                    # contiguous and .contiguous() would hand the caller a VIEW the next call of this bucket overwrites
                    mel, ralpha = mel[:, :t2].clone(), ralpha[:, :T1, :t2].clone()
                if want_lengths:
                    ml = ml.to(torch.int64)                    # (a new tensor: `ml` is the text graph's static output)
            else:
                mel, ralpha = self._infer_mel(ws, ws2, e, tl, ml, T2b)
                ml = ml.to(torch.int64) if want_lengths else ml
            return mel, ml, ralpha


def _opt_property(name: str):
    return property(lambda self: getattr(self.opt, name), lambda self, v: setattr(self.opt, name, type(getattr(LaunchOptions(), name))(v)))


for _f in dataclasses.fields(LaunchOptions):
    setattr(EfficientTTSCNN, _f.name, _opt_property(_f.name))
EfficientTTSCNN.RESCONV_MIN_ROWS = _opt_property("resconv_min_rows")      # (the names tests and tools of earlier rounds use)
EfficientTTSCNN.SMALL_M_ROWS = _opt_property("small_m_rows")
