"""On-device log-mel front-end: audio -> the `speech` tensor EfficientTTSCNN.forward() takes.

Replaces, for a whole batch on the GPU, what the reference does per item on CPU dataloader workers:
`mel_spectrogram` (nntts/datasets/meldataset.py:49-82) called from `TextMelLoader.get_mel`
(nntts/datasets/taco2_data.py:66-76), plus the mel padding of `TextMelCollate` (:122-139).
Pipeline (csrc/efts_frontend.hip): frame_pack (reflect pad + hann window -> bf16x3 operand planes)
-> efts_gemm against a real-DFT plane (MFMA) -> logmel (magnitude, Slaney mel filterbank, log clamp).
Round 6: the DFT is split by decimation in time (`radix` >= 2): one batched efts_gemm of `radix` real
(n_fft / radix)-point DFTs -- radix times fewer FLOPs than the dense n_fft-point product (26.9 instead of 107.6 GFLOP per
64 x 800 frames) -- recombined by `radix` complex multiply-adds per bin in the logmel kernel.  And for the reference's own configuration
(n_fft 1024, hop 256) the default is `radix=0`: ONE launch, `efts_logmel_fft`, audio in and log-mels out, the STFT as an fp32 FFT in
registers and LDS (two real frames per complex 1024-point FFT) -- no operand plane, no spectrum in memory.
No CPU fallback: the HIP library is required.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import lib as L
from . import ops as O
from .ops import F32Rows, PackedWeight, Plane, Rows


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """The filterbank the reference obtains from librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)
    (meldataset.py:65; librosa >= 0.8 defaults htk=False, norm='slaney'), computed here from the
    published definition so that the package has no librosa dependency: Slaney's mel scale (linear at
    200/3 Hz per mel below 1 kHz, logarithmic with 27 mels per factor 6.4 above), triangles between
    neighbouring centre frequencies on the rFFT bin grid, each scaled to unit area in Hz."""
    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), f * 3.0 / 200.0)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)), m * 200.0 / 3.0)

    bins = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, bins.size))
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        rise = (bins - lo) / (ce - lo)
        fall = (hi - bins) / (hi - ce)
        fb[m] = np.clip(np.minimum(rise, fall), 0.0, None) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


class LogMelFrontend:
    """mel, frames = LogMelFrontend(device)(audio, lengths)

    audio: [B, L] float32 in [-1, 1] (or int16, scaled by 1/32768 as TextMelLoader.get_mel does), on the
    device or the host; lengths: [B] sample counts.  Returns mel [B, T, 80] float32 (zero past each item's
    frame count, T = max frames) and frames [B] int64 -- the (speech, speech_lengths) of the model."""

    def __init__(self, device, sampling_rate: int = 22050, n_fft: int = 1024, hop_size: int = 256, win_size: int = 1024,
                 num_mels: int = 80, fmin: float = 0.0, fmax: float = 8000.0, max_wav_value: float = 32768.0, radix: Optional[int] = None):
        if win_size != n_fft:
            raise ValueError("win_size must equal n_fft (the reference's configuration)")
        fft_ok = n_fft == 1024 and hop_size == 256 and num_mels <= 80
        if radix is None:
            radix = 0 if fft_ok else 4
        if radix == 0 and not fft_ok:
            raise ValueError("radix 0 (the fused FFT launch) is built for n_fft 1024, hop 256, at most 80 mel bins")
        if radix < 0 or (radix and (n_fft % radix or (n_fft // radix) % 32)):
            raise ValueError("radix must divide n_fft into whole 32-sample operand chunks (1 = the dense n_fft-point product, 0 = the fused FFT launch)")
        self.radix = radix
        self.dev = torch.device(device)
        self.n_fft, self.hop, self.n_mels, self.n_bins = n_fft, hop_size, num_mels, n_fft // 2 + 1
        self.max_wav_value = max_wav_value
        L.load()
        L.require_device()
        fb = slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        rng = np.zeros((num_mels, 2), dtype=np.int32)
        for m in range(num_mels):
            nz = np.nonzero(fb[m])[0]
            rng[m] = (nz[0], nz[-1] + 1) if nz.size else (0, 0)
        self.basis = torch.from_numpy(fb).to(self.dev).contiguous()
        self.ranges = torch.from_numpy(rng).to(self.dev).contiguous()
        self.window = torch.hann_window(win_size, dtype=torch.float32).to(self.dev)           # periodic hann, meldataset.py:67
        if radix == 0:
            self.dft = self.twiddle = None
            self.n_out = self.ld_spec = 0
        elif radix == 1:
            # real DFT as a B operand plane: rows 0..n_bins-1 = cos(2 pi f k / N), rows n_bins.. = -sin
            k = np.arange(n_fft, dtype=np.float64)[None, :]
            f = np.arange(self.n_bins, dtype=np.float64)[:, None]
            ang = 2.0 * math.pi * f * k / n_fft
            dft = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0).astype(np.float32)
            self.n_out = 2 * self.n_bins
            self.ld_spec = O.roundup(self.n_out, 4)
            self.twiddle = None
            with O.stream_scope():
                self.dft = PackedWeight(self.n_out, n_fft, 1, 2, self.dev)
                self.dft.pack(torch.from_numpy(dft).to(self.dev).contiguous())
        else:
            # the real M-point DFT of one sub-sequence, M = n_fft / radix: M independent real outputs -- rows 0 .. M/2: cos(2 pi g j / M)
            # (re Y[0 .. M/2]), rows M/2 + g, g = 1 .. M/2 - 1: -sin(2 pi g j / M) (im Y[1 .. M/2 - 1]; im Y[0] = im Y[M/2] = 0)
            M = n_fft // radix
            j = np.arange(M, dtype=np.float64)[None, :]
            gc = np.arange(M // 2 + 1, dtype=np.float64)[:, None]
            gs = np.arange(1, M // 2, dtype=np.float64)[:, None]
            dft = np.concatenate([np.cos(2.0 * math.pi * gc * j / M), -np.sin(2.0 * math.pi * gs * j / M)], axis=0).astype(np.float32)
            assert dft.shape == (M, M)
            self.n_out, self.ld_spec, self.sub = M, n_fft, M
            # twiddle[p][f] = (cos, sin)(2 pi p f / n_fft): X[f] = sum_p (cos - i sin) Y_p[f mod M]
            pf = np.arange(radix, dtype=np.float64)[:, None] * np.arange(self.n_bins, dtype=np.float64)[None, :]
            tw = np.stack([np.cos(2.0 * math.pi * pf / n_fft), np.sin(2.0 * math.pi * pf / n_fft)], axis=-1).astype(np.float32)
            self.twiddle = torch.from_numpy(tw).to(self.dev).contiguous()
            with O.stream_scope():
                self.dft = PackedWeight(M, M, 1, 2, self.dev)
                self.dft.pack(torch.from_numpy(dft).to(self.dev).contiguous())
        self._ws = {}
        self._pins = {}

    _RING = 8

    def _to_device_async(self, lh, fh) -> torch.Tensor:
        """lengths and frame counts as ONE int32 [2, B] device tensor, copied from a pinned staging buffer without blocking the host (a pageable
        copy waits for the kernels queued in front of it: the host could never run ahead of the fused launch).  A ring of staging buffers per
        batch size, each guarded by the event of the copy that last read it."""
        B = len(lh)
        ring = self._pins.setdefault(B, dict(next=0, slots=[]))
        if len(ring["slots"]) < self._RING:
            ring["slots"].append([torch.empty(2, B, dtype=torch.int32).pin_memory(), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"]]
            ring["next"] = (ring["next"] + 1) % self._RING
            if slot[1] is not None:
                slot[1].synchronize()
        buf = slot[0]
        buf[0] = torch.tensor(lh, dtype=torch.int32)
        buf[1] = torch.tensor(fh, dtype=torch.int32)
        dev_t = buf.to(self.dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        slot[1] = ev
        return dev_t

    def frames_of(self, lengths: torch.Tensor) -> torch.Tensor:
        return torch.div(lengths.to(torch.int64), self.hop, rounding_mode="floor")

    @torch.no_grad()
    def __call__(self, audio: torch.Tensor, lengths: torch.Tensor, max_frames: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if audio.dim() != 2:
            raise ValueError("audio must be [B, L]")
        B = audio.shape[0]
        pcm16 = audio.dtype == torch.int16 and self.radix == 0
        if pcm16:
            audio = audio.to(self.dev).contiguous()             # the fused launch scales int16 PCM at its loads (taco2_data.py:70): no conversion pass
        else:
            if audio.dtype == torch.int16:
                audio = audio.to(self.dev).to(torch.float32) / self.max_wav_value              # taco2_data.py:70
            audio = audio.to(self.dev, torch.float32).contiguous()
        # host side of a call: the checks on plain ints, ONE small host-to-device copy (lengths and frame counts together), one output allocation
        # (three CPU-tensor reductions and three copies took 60 us per call -- as long as the fused launch itself)
        lh = lengths.detach().to("cpu", torch.int64).tolist()
        if min(lh) <= (self.n_fft - self.hop) // 2:
            raise ValueError("every item must be longer than the reflect padding (n_fft - hop) / 2")
        if max(lh) > audio.shape[1]:
            raise ValueError("lengths exceed the audio buffer")
        fh = [l // self.hop for l in lh]
        T = max(fh) if max_frames is None else int(max_frames)
        both = self._to_device_async(lh, fh)
        li, fi = both[0], both[1]
        frames_d = fi.to(torch.int64)
        if self.radix == 0:
            out = torch.empty(B, T, self.n_mels, dtype=torch.float32, device=self.dev)
            with O.stream_scope():
                if pcm16:
                    L.check(L.load().efts_logmel_fft_pcm16(audio.data_ptr(), audio.shape[1], 1.0 / self.max_wav_value, li.data_ptr(), self.window.data_ptr(),
                                                           self.basis.data_ptr(), self.ranges.data_ptr(), out.data_ptr(), B, T, self.n_fft, self.hop, self.n_mels,
                                                           O._stream()), "efts_logmel_fft_pcm16")
                else:
                    L.check(L.load().efts_logmel_fft(audio.data_ptr(), audio.shape[1], li.data_ptr(), self.window.data_ptr(), self.basis.data_ptr(),
                                                     self.ranges.data_ptr(), out.data_ptr(), B, T, self.n_fft, self.hop, self.n_mels, O._stream()),
                            "efts_logmel_fft")
            return out, frames_d
        rs = Rows(B, T)
        key = (B, T)
        if key not in self._ws:
            if len(self._ws) > 4:
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = (Plane.for_rows(rs, self.n_fft, 2, self.dev), F32Rows(rs, self.ld_spec, self.dev))
        fr, spec = self._ws[key]
        out = torch.empty(B, T, self.n_mels, dtype=torch.float32, device=self.dev)
        lib = L.load()
        with O.stream_scope():
            if self.radix == 1:
                L.check(lib.efts_frame_pack(audio.data_ptr(), audio.shape[1], li.data_ptr(), self.window.data_ptr(), fr.ptr, fr.ld,
                                            B, T, rs.Tp, self.n_fft, self.hop, 2, O._stream()), "efts_frame_pack")
                O.gemm(a=fr, b_ptr=self.dft.ptr, ldb=self.dft.ld, m=rs.rows, n=self.n_out, out_f32_ptr=spec.ptr, ldo=self.ld_spec)
                L.check(lib.efts_logmel(spec.ptr, self.ld_spec, self.basis.data_ptr(), self.ranges.data_ptr(), fi.data_ptr(),
                                        out.data_ptr(), B, T, rs.Tp, self.n_bins, self.n_mels, O._stream()), "efts_logmel")
            else:
                M = self.sub
                L.check(lib.efts_frame_pack_dit(audio.data_ptr(), audio.shape[1], li.data_ptr(), self.window.data_ptr(), fr.ptr, fr.ld,
                                                B, T, rs.Tp, self.n_fft, self.hop, 2, self.radix, O._stream()), "efts_frame_pack_dit")
                # `radix` real M-point DFTs as ONE batched product: item p reads columns p * M .. of every row, writes columns p * M .. of the spectrum row
                O.gemm(a=fr, b_ptr=self.dft.ptr, ldb=self.dft.ld, m=rs.rows, n=M, batch=self.radix, nchunk=M // O.chunk_k(2),
                       a_batch_stride=M // O.chunk_k(2) * 128, out_f32_ptr=spec.ptr, ldo=self.ld_spec, out_batch_stride=M)
                L.check(lib.efts_logmel_dit(spec.ptr, self.ld_spec, self.basis.data_ptr(), self.ranges.data_ptr(), fi.data_ptr(),
                                            self.twiddle.data_ptr(), out.data_ptr(), B, T, rs.Tp, self.n_bins, self.n_mels, self.radix, O._stream()),
                            "efts_logmel_dit")
        return out, frames_d
