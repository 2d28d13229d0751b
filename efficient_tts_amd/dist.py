"""Data-parallel training over RCCL/xGMI: one process per GPU, gradients averaged with a bucketed
all-reduce that overlaps the backward pass (replaces torch DistributedDataParallel of the reference,
nntts/bin/train.py:210-216; payload 20 587 601 fp32 = 82.35 MB per step).

The engine lays its flat gradient buffer out in backward-completion order (efficient_tts_amd/train.py),
so a bucket is a contiguous slice that is FINAL as soon as its stage of the backward has been
enqueued: [mel head + decoder] -> [duration predictor + mel encoder + prenet] -> [K/V + text encoder +
embedding].  Each bucket's all-reduce is issued on a side stream right after that stage while the
compute stream continues with the next stage; the optimizer waits on the last bucket only.
xGMI is point-to-point (7 links/GPU): three 25-35 MB buckets keep each ring/tree transfer long
enough to be link-bound rather than latency-bound.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class BucketReducer:
    """Sum-reduce contiguous slices of one flat tensor across the group, asynchronously, in call order.

    algo "allreduce": one `all_reduce` per bucket (RCCL picks ring / tree).
    algo "rs_ag"    : `reduce_scatter_tensor` + `all_gather_into_tensor` per bucket (xGMI is
                      point-to-point, 7 links per GPU; a ring all-reduce of 82 MB is bound by ONE link (~0.94 ms), the direct
                      reduce-scatter + all-gather exchange uses all of them (~0.135 ms)).  The bucket is staged through a
                      scratch buffer padded to a multiple of the world size (two device copies of the bucket, ~10 us).
    timing          : HIP events around every bucket's collective on the communication stream, and on the compute stream
                      when the first bucket is handed over / when the optimizer starts waiting: `stats()`."""

    def __init__(self, flat: torch.Tensor, bucket_ends: List[int], group=None, algo: str = "allreduce", timing: bool = False):
        if algo not in ("allreduce", "rs_ag"):
            raise ValueError("algo must be 'allreduce' or 'rs_ag'")
        self.flat, self.ends, self.group, self.algo, self.timing = flat, list(bucket_ends), group, algo, timing
        self.starts = [0] + self.ends[:-1]
        self.world = dist.get_world_size(group)
        self.works = []
        self.comm_stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
        self.pad, self.shard = [], []
        if algo == "rs_ag":
            for s0, e0 in zip(self.starts, self.ends):
                chunk = (e0 - s0 + self.world - 1) // self.world
                self.pad.append(torch.zeros(chunk * self.world, dtype=flat.dtype, device=flat.device))
                self.shard.append(torch.empty(chunk, dtype=flat.dtype, device=flat.device))
        self.events = {}                  # timing: name -> torch.cuda.Event of the current step
        self.pending = False              # buckets handed over since the last finish()

    def _event(self, name: str, stream=None) -> None:
        # (not while the step is being captured as a hipGraph: timing events are an eager-loop probe)
        if self.timing and self.comm_stream is not None and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.flat.device))
            self.events[name] = ev

    def mark(self, name: str) -> None:
        """an event on the COMPUTE stream (the training engine calls mark("backward_start"))"""
        self._event(name)

    def _collective(self, i: int) -> None:
        """bucket i's exchange, enqueued on the CURRENT stream (the communication stream on a GPU).  On a GPU the collectives are
        issued as stream-ordered SYNCHRONOUS ops (async_op=False): the NCCL backend then enqueues them on the stream they are called
        on, so the communication stream stays a plain fork of the compute stream -- the only dependency shape ROCm 7.2 captures safely
        (a forked stream waiting for an event of ANOTHER forked stream -- what the backend's internal stream of an async op is --
        crashes hipStreamEndCapture: tools/attic/gpu_probe_capture4.py) -- and the whole step, collectives included, can be one hipGraph.
        Neither the host nor the compute stream waits: only the communication stream is ordered behind the collective."""
        view = self.flat[self.starts[i]:self.ends[i]]
        on_gpu = self.comm_stream is not None
        if self.algo == "allreduce":
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=not on_gpu)
            if not on_gpu:
                self.works.append(w)
            return
        pad, shard, n = self.pad[i], self.shard[i], view.numel()
        pad[:n].copy_(view)                                       # the tail beyond n stays zero
        if on_gpu:
            dist.reduce_scatter_tensor(shard, pad, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(pad, shard, group=self.group)
        else:
            dist.reduce_scatter_tensor(shard, pad, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            dist.all_gather_into_tensor(pad, shard, group=self.group, async_op=True).wait()
        view.copy_(pad[:n])

    def reduce(self, i: int) -> None:
        self.pending = True
        if self.comm_stream is not None:
            if i == 0:
                self._event("first_bucket_ready")
            # the bucket is final once everything already enqueued on the compute stream has run
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm_stream):
                self._event(f"b{i}_start", self.comm_stream)
                self._collective(i)
                self._event(f"b{i}_end", self.comm_stream)
        else:
            self._collective(i)

    def finish(self) -> None:
        """make the compute stream (or the host, on CPU) wait for every outstanding bucket"""
        if not self.pending:
            return
        self.pending = False
        self._event("wait_start")
        for w in self.works:
            w.wait()
        self.works = []
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)
        self._event("wait_end")

    def stats(self) -> dict:
        """after a synchronize: milliseconds of the last step (timing=True, GPU only)"""
        ev = self.events
        if not self.timing or "wait_end" not in ev:
            return {}
        out = dict(algo=self.algo, world=self.world,
                   bucket_mb=[(e0 - s0) * self.flat.element_size() / 1e6 for s0, e0 in zip(self.starts, self.ends)],
                   bucket_ms=[ev[f"b{i}_start"].elapsed_time(ev[f"b{i}_end"]) for i in range(len(self.ends)) if f"b{i}_end" in ev],
                   exposed_ms=ev["wait_start"].elapsed_time(ev["wait_end"]))
        if "backward_start" in ev:
            out["backward_ms"] = ev["backward_start"].elapsed_time(ev["wait_start"])
            out["first_bucket_after_ms"] = ev["backward_start"].elapsed_time(ev["first_bucket_ready"])
        return out


class DistributedEFTS(torch.nn.Module):
    """DDP stand-in with the attributes the reference trainer uses (`.module`, call-through)."""

    def __init__(self, module, group=None, algo: str = "allreduce", timing: bool = False, force_reducer: bool = False):
        """force_reducer: run the bucketed exchange even in a group of ONE rank (tests: the RCCL code path on a single GPU; a
        one-rank sum is the identity, so the step must equal the wrapper-free one bit for bit)"""
        super().__init__()
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        from .autograd import engine_of
        self.engine = engine_of(module)
        self.engine.bound.add("DistributedEFTS")
        self.reducer: Optional[BucketReducer] = None
        if self.world > 1 or (force_reducer and dist.is_initialized()):
            self.reducer = BucketReducer(self.engine.flat, self.engine.bucket_ends, group, algo=algo, timing=timing)
            self.engine.bucket_hook = self.reducer.reduce
            self.engine.join_reduce = self.reducer.finish
            self.engine.mark = self.reducer.mark if timing else None
            # identical initial parameters on every rank (DDP broadcasts rank 0's)
            for p in module.parameters():
                dist.broadcast(p.data, src=0, group=group)
            module._packed_sig = None           # in-place through .data: no version bump, so drop any packed operand planes

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def finish_reduce(self) -> None:
        if self.reducer is not None:
            self.reducer.finish()
