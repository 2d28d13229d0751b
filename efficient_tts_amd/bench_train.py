"""bench.py --workload train32: the data-parallel training step (BASELINE configs 3 and 4):
fwd + bwd + clip(1.0) + Adam-amsgrad + WarmupLR at batch 32 per GPU, (T1, T2) = (128, 800), with the
gradient all-reduce over RCCL overlapped with backward when world > 1
(reference: nntts/trainers/efficient_tts_trainer.py:139-160 under nntts/bin/train.py:210-216).

`measure_train` is the measurement (also the `train32` sub-record of the default bench line); `run_train` prints it.
With world > 1 the run validates itself before it is timed: the process-group backend must be "nccl" (= RCCL), two optimizer
steps are taken and every rank's parameters are compared BIT FOR BIT with rank 0's, and RCCL's own log (NCCL_DEBUG=INFO into
per-rank files, set up by bench.py before the process group exists) is searched for the topology / algorithm lines, which go
into the `dp` record -- so that the first run on a real 8-GPU node explains its scaling number by itself."""
import glob
import json
import os
import time

import torch

TRAIN_FLOP_PER_ITEM = 3 * 21.43e9


def _params_fingerprint(model, dev) -> torch.Tensor:
    """an exact fingerprint of all parameters: wrap-around int64 sums of their bit patterns, plain and position-weighted"""
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    for p in model.parameters():
        bits = p.detach().reshape(-1).view(torch.int32).to(torch.int64)
        acc[0] += bits.sum()
        acc[1] += (bits * (torch.arange(bits.numel(), device=dev, dtype=torch.int64) % 8191 + 1)).sum()
    return acc


def _rccl_log_lines(limit: int = 12):
    """what RCCL said about the rings / trees / algorithms it uses (NCCL_DEBUG_FILE of this job, rank 0's view)"""
    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return None
    out = []
    for f in sorted(glob.glob(pat.replace("%h", "*").replace("%p", "*")))[:2]:
        try:
            for ln in open(f, errors="replace"):
                if any(k in ln for k in ("Algo", "algo", "Proto", "Connected all", "Channel", "nranks", "NCCL version", "RCCL version", "P2P", "XGMI", "xgmi")):
                    out.append(ln.strip()[-200:])
                    if len(out) >= limit:
                        return out
        except OSError:
            pass
    return out or None


XGMI_LINKS, XGMI_LINK_GBPS = 7, 153.0        # per GPU, SURVEY.md section 5 (point-to-point: a ring is bound by ONE link)


def exchange_model(stats: dict, with_ms: float, without_ms: float, world: int) -> dict:
    """what the data-parallel line says about its own collectives (pure arithmetic: unit-tested on the CPU): the step with and without
    the exchange under the same clock, their ratio as the scaling efficiency of this run, and per bucket the all-reduce BUS bandwidth
    2 (N - 1) / N * bytes / time against one xGMI link (what a ring can reach) and against all seven (the direct exchange's ceiling)"""
    out = dict(step_ms_with_exchange=with_ms, step_ms_no_exchange=without_ms,
               efficiency=(without_ms / with_ms) if with_ms > 0 else None,
               xgmi=dict(links_per_gpu=XGMI_LINKS, link_gbps=XGMI_LINK_GBPS, all_links_gbps=XGMI_LINKS * XGMI_LINK_GBPS))
    mb, ms = stats.get("bucket_mb") or [], stats.get("bucket_ms") or []
    if world > 1 and mb and len(mb) == len(ms):
        f = 2.0 * (world - 1) / world
        bw = [f * m / t if t and t > 0 else None for m, t in zip(mb, ms)]           # MB / ms = GB/s
        out["bucket_busbw_gbps"] = bw
        out["bucket_busbw_frac_of_one_link"] = [None if b is None else b / XGMI_LINK_GBPS for b in bw]
        out["bucket_busbw_frac_of_all_links"] = [None if b is None else b / (XGMI_LINKS * XGMI_LINK_GBPS) for b in bw]
    return out


def measure_train(a, world, rank, dev, wl, steps, warmup, precision=None):
    """K timed training steps (barrier + synchronize on both sides, max over ranks) -> the record (rank 0; None elsewhere).

    One process: the timed steps are hipGraph replays (step_graph.GraphedStep), the eager loop is timed after them.
    Data parallel (world > 1): (1) self-validation -- nccl backend, two eager steps, bit-exact parameter fingerprint of every rank;
    (2) K eager steps timed (the bucketed RCCL exchange launched from the engine's hooks, per-bucket events -> `dp`); (3) the same
    step as ONE hipGraph replay per rank, collectives captured with it, K replays timed, fingerprint compared again.  `value` is the
    graphed loop's when every rank captured (the ranks agree on that through an all-reduce), else the eager loop's; both are
    reported."""
    import torch.distributed as dist
    from . import EfficientTTSCNN, ops as P
    from .dist import DistributedEFTS
    from .optim import EftsAdam, WarmupLR
    from .step_graph import GraphedStep
    precision = precision or a.precision
    B, T1, T2 = wl["B"], wl["T1"], wl["T2"]
    for kv in getattr(a, "train_set", []):
        from . import train as _tr
        k, v = kv.split("=")
        assert hasattr(_tr, k), f"efficient_tts_amd.train has no switch {k}"
        setattr(_tr, k, int(v))
    torch.manual_seed(0)
    model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01,
                            precision=precision).to(dev).train()
    opt = EftsAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
    sch = WarmupLR(opt, warmup_steps=4000)
    ddp = DistributedEFTS(model, algo=getattr(a, "dp_algo", "allreduce"), timing=True) if world > 1 else None
    net = ddp if ddp is not None else model
    g = torch.Generator().manual_seed(1234 + rank)
    text = torch.randint(0, 76, (B, T1), generator=g).to(dev)
    mel = torch.randn(B, T2, 80, generator=g).to(dev)
    tl = torch.full((B,), T1, dtype=torch.int64, device=dev)
    sl = torch.full((B,), T2, dtype=torch.int64, device=dev)

    def step():
        loss, stats, *_ = net(text=text, text_lengths=tl, speech=mel, speech_lengths=sl)
        opt.zero_grad()
        loss.backward()
        if ddp is not None:
            ddp.finish_reduce()
        opt.step(grad_scale=1.0 / world)
        sch.step()
        return loss

    def timed(run, n):
        """exactly n steps between barrier + synchronize on both sides; seconds, max over ranks"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, loss

    def fingerprints():
        fp = _params_fingerprint(model, dev)
        every = [torch.zeros_like(fp) for _ in range(world)]
        dist.all_gather(every, fp)
        return all(bool(torch.equal(e, every[0])) for e in every), every

    rows = P.Rows(B, T2).rows
    want_graph = bool(getattr(a, "train_graph", 1))

    def bracketed(run, n=3):
        """durations of the mel-length k5 launches (HIP events on their launch stream) over n eager steps of their own, NOT inside a timed
        loop: creating the timing events can stall the host for tens of milliseconds when the runtime grows its event pool (seen as one
        77 ms step in ten, i.e. a 10 ms "eager step", when the bracketing ran inside the timed eager loop)"""
        P.PROFILE, P.PROFILE_TAG = [], (5, rows, 512)
        try:
            for _ in range(n):
                run()
            torch.cuda.synchronize()
            return [s.elapsed_time(e) * 1e-3 for (tag, s, e) in (P.PROFILE or []) if tag == (5, rows, 512)]
        finally:
            P.PROFILE, P.PROFILE_TAG = None, None

    def finish(dt, issue, lv, durs, eager_dt, graph_dt, graph_note, poisoned):
        """the record from what has been measured (rank 0; None elsewhere)"""
        assert lv == lv, "NaN loss"
        avg = sum(durs) / max(len(durs), 1)
        conv_flop = 2.0 * B * T2 * 512 * 512 * 5
        from . import train as _tr
        split = model.split
        dg = _tr._RESCONV_DGRAD if _tr._RESCONV_DGRAD >= 0 else (1 if split == 1 else 3)
        on_rc = model._on_resconv(P.Rows(B, T2))
        big = split == 1 and ((rows + 251) // 252) * 4 >= 400           # efts_gemm's own rule for the 256-row kernel
        other = "conv5_kernel<split=1> (256-row tiles)" if big else f"gemm_kernel<taps=5,split={split}> (124-row tiles)"
        if on_rc and _tr._RESCONV_FWD:
            kname = (f"the k5 launches at mel length: resconv5_kernel<split={split}> (forward of the stacks selected by _RESCONV_FWD={_tr._RESCONV_FWD}, "
                     f"dgrad of those selected by {dg}: bit 0 decoder, bit 1 mel encoder) and {other} (the rest)")
        else:
            kname = other
        if rank != 0:
            return None
        frames = world * B * T2 * steps
        res = dict(metric="mel-frames/sec (EFTS-CNN training step, batch 32/GPU, 80-mel LJSpeech shape)", value=frames / dt,
                   unit="mel-frames/s", n_gpus=world, steps=steps, warmup=warmup, ms_per_step=dt / steps * 1e3,
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="bf16" if precision == "bf16" else "bf16x3 (split-bf16 MFMA, fp32-class)", data="synthetic",
                   config=dict(workload=wl["desc"], batch_per_gpu=B, phoneme_len=T1, mel_len=T2, precision=precision,
                               parallelism=f"dp{world}", optimizer="Adam-amsgrad fused, clip 1.0, WarmupLR 4000",
                               allreduce="RCCL, 3 buckets overlapped with backward" if world > 1 else "none", step_issue=issue),
                   eager_ms_per_step=None if eager_dt is None else eager_dt / steps * 1e3,
                   graph_ms_per_step=None if graph_dt is None else graph_dt / steps * 1e3,
                   per_gpu=frames / dt / world, tflops=TRAIN_FLOP_PER_ITEM * B * world * steps / dt / 1e12, loss=lv,
                   roofline=dict(bound="mfma", kernel=f"{kname}; fwd + dgrad launches, timed with events in the eager loop while the text-length stream runs beside them",
                                 achieved=conv_flop / avg / 1e12 if avg else None, peak=2500.0, unit="TFLOP/s",
                                 frac=conv_flop / avg / 1e12 / 2500.0 if avg else None, traffic=None,
                                 avg_launch_us=avg * 1e6, launches_measured=len(durs)))
        if graph_note:
            res["config"]["graph_note"] = graph_note
        if world > 1 and poisoned:
            res["config"]["device_state"] = "a failed capture left streams in capture mode: no further device work in this process"
        if ddp is not None:
            # the last eager step's communication: one record that explains the scaling number (backend, ranks, algorithm,
            # bytes and time per bucket, how much of it the backward did NOT hide)
            res["dp"] = dict(backend=dist.get_backend(), ranks=dist.get_world_size(), grad_mb=ddp.engine.numel * 4 / 1e6,
                             nccl_env={k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                             selfcheck=selfcheck, rccl_log=_rccl_log_lines(), **(dp_stats or {}))
        return res

    selfcheck, dp_stats, eager_dt, graph_dt, graph_note, poisoned = None, None, None, None, None, False
    if world > 1:
        # ---- (1) self-validation before anything is timed
        backend = dist.get_backend()
        # (tests/test_dist_gpu.py drives this very code with two ranks sharing the test GPU, which only gloo allows)
        assert backend == "nccl" or getattr(a, "allow_gloo", False), f"data-parallel runs use the RCCL backend ('nccl'), got {backend!r}"
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        same, every = fingerprints()
        assert same, "data-parallel replicas diverged after 2 steps: " + str([e.tolist() for e in every])
        # every rank draws its OWN minibatch (seed 1234 + rank): gather the seeds and a checksum of each rank's batch
        mine = torch.tensor([1234 + rank, int(text.to(torch.int64).sum()) * 1000003 + int((mel.double() * 1e3).round().to(torch.int64).sum().item() % 1000003)],
                            dtype=torch.int64, device=dev)
        allb = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        selfcheck = dict(backend=backend, world_seen_by_rccl=dist.get_world_size(), steps=2, replicas_bit_identical=same,
                         fingerprint=[int(v) for v in every[0].tolist()], seeds=[int(t[0]) for t in allb],
                         batches_distinct=len({int(t[1]) for t in allb}) == world)
        # ---- (2) the eager loop (what the reference's trainer issues), conv launches bracketed by events for the roofline
        for _ in range(max(warmup, 2)):
            step()
        eager_dt, loss = timed(step, steps)
        dp_stats = ddp.reducer.stats()
        durs = bracketed(step)
        lv_eager = float(loss)
        # ---- (2b) the same K steps with the exchange switched off (the engine's bucket hooks detached): what the step costs this rank
        # without its collectives, under the same clock -- so that the line states its own scaling efficiency.  Replicas diverge in this
        # loop (every rank applies its own gradient), hence parameters, optimizer moments, step counters and the learning-rate schedule
        # are put back afterwards, and the fingerprint check is repeated
        eng = ddp.engine
        snap = (opt.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.vmax.clone(), opt.t, sch.state_dict(), model.dropout_calls,
                [dict(g_) for g_ in ({k: v for k, v in g.items() if k != "params"} for g in opt.param_groups)])
        hooks = (eng.bucket_hook, eng.join_reduce)
        eng.bucket_hook, eng.join_reduce = None, None
        try:
            noex_dt, _ = timed(step, steps)
        finally:
            eng.bucket_hook, eng.join_reduce = hooks
            opt.flat_p.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.vmax.copy_(snap[3])
            opt.t = snap[4]
            sch.load_state_dict(snap[5])
            model.dropout_calls = snap[6]
            for g_, old in zip(opt.param_groups, snap[7]):
                g_.update(old)
            model._packed_sig = None
        torch.cuda.synchronize()
        same_again, every_again = fingerprints()
        assert same_again, "replicas differ after the no-exchange loop was undone: " + str([e.tolist() for e in every_again])
        dp_stats = dict(dp_stats or {})
        dp_stats.update(exchange_model(dp_stats, eager_dt / steps * 1e3, noex_dt / steps * 1e3, world))
        # ---- (3) the step as one hipGraph replay per rank, collectives inside.  LAST device work of the measurement, and fenced: a
        # capture that fails half-way can leave its forked streams in capture mode (ROCm 7.2), after which any use of the default stream
        # raises -- the eager numbers above must survive that, so everything from here on is inside one try, and the record is put
        # together from host values only.  gloo cannot be captured at all (its collectives synchronise the host): not attempted.
        poisoned = False
        # what a watchdog prints if the attempt below never returns (bench.py, N > 1): the eager data-parallel record as it stands
        a.partial_train = finish(eager_dt, "eager launches", lv_eager, durs, eager_dt, None, "graph attempt did not return: eager loop timed", False)
        if want_graph and backend != "nccl":
            graph_note = f"{backend} collectives cannot be captured; eager loop timed"
        elif want_graph:
            try:
                graphed = GraphedStep(ddp, opt, sch)
                ok = 1
                try:
                    for _ in range(max(warmup, 3)):           # eager, capture, replays
                        loss = graphed(text, tl, mel, sl)[0]
                    torch.cuda.synchronize()
                    ok = int(graphed.replays >= 1)
                except Exception as exc:                       # noqa: BLE001 -- reported, the eager measurement stands
                    ok, graph_note = 0, f"capture failed on rank {rank}: {exc}"[:300]
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # every rank replays, or none does
                if int(flag.item()) == 1:
                    r0 = graphed.replays
                    bufs = graphed.inputs(text, tl, mel, sl)          # the batch resident in the buffers the captured launches read
                    if bufs is not None:                              # (as in the one-process measurement below: no copy per step)
                        for s_, t in zip(bufs, (text, tl, mel, sl)):
                            s_.copy_(t)
                    gargs = bufs if bufs is not None else (text, tl, mel, sl)
                    gdt, loss = timed(lambda: graphed(*gargs)[0], steps)
                    assert graphed.replays - r0 == steps, "the timed steps were not graph replays"
                    same2, every2 = fingerprints()
                    assert same2, "data-parallel replicas diverged under graph replays: " + str([e.tolist() for e in every2])
                    selfcheck["replicas_bit_identical_after_graph_replays"] = same2
                    lv_eager = float(loss)
                    graph_dt = gdt
                elif graph_note is None:
                    graph_note = "capture failed on another rank; eager loop timed"
            except Exception as exc:                           # noqa: BLE001
                poisoned = True
                graph_dt = None
                graph_note = ((graph_note + "; ") if graph_note else "") + f"graph attempt abandoned on rank {rank}: {exc}"[:300]
        dt = graph_dt if graph_dt is not None else eager_dt
        issue = "one hipGraph replay per step and rank, bucket collectives captured (step_graph.GraphedStep)" if graph_dt is not None else "eager launches"
    else:
        graphed = GraphedStep(model, opt, sch) if want_graph else None

        def gstep():
            # inputs resident in HBM (the bench contract): once the step is captured, the batch lives in the buffers its launches read
            # (GraphedStep.inputs), as a loader that copies straight into them would leave it -- no device-to-device copy per step
            bufs = graphed.inputs(text, tl, mel, sl)
            if bufs is not None and not getattr(gstep, "filled", False):
                for s_, t in zip(bufs, (text, tl, mel, sl)):
                    s_.copy_(t)
                gstep.filled = True
            return graphed(*(bufs if bufs is not None else (text, tl, mel, sl)))[0]
        run = gstep if graphed is not None else step
        for _ in range(max(warmup, 2)):
            loss = run()
        dt, loss = timed(run, steps)
        if graphed is not None and os.environ.get("EFTS_BENCH_TRAIN_NO_EAGER") == "1":
            graph_dt = dt                          # (profiling a replay's timeline: nothing issued behind the timed replays)
        elif graphed is not None:
            # the same step issued eagerly (what the reference's loop does), with the conv launches bracketed by events for the roofline
            assert graphed.replays >= steps, "the timed steps were not graph replays"
            graph_dt = dt
            eager_dt, _ = timed(step, max(3, min(steps, 10)))
            eager_dt = eager_dt / max(3, min(steps, 10)) * steps
        if os.environ.get("EFTS_BENCH_TRAIN_NO_EAGER") == "1" and graphed is not None:
            durs = []
        else:
            durs = bracketed(step)
        issue = ("one hipGraph replay per step (step_graph.GraphedStep), the batch resident in the buffers the captured launches read "
                 "(GraphedStep.inputs: where the trainer's loader copies it, trainer._stage)") if graphed is not None else "eager launches"
    lv = lv_eager if world > 1 else float(loss)
    return finish(dt, issue, lv, durs, eager_dt, graph_dt, graph_note, poisoned)


def run_train(a, world, rank, dev, wl, cpu_baseline_fn=None):
    import threading
    import torch.distributed as dist
    dog = None
    if world > 1:
        # a captured-step attempt that never returns must not take the eager data-parallel record with it (bench.py's watchdog, same idea)
        def fire():
            part = getattr(a, "partial_train", None)
            if rank == 0 and isinstance(part, dict):
                part.setdefault("config", {})["watchdog"] = "the run did not finish in time (a rank hung): the eager record as it stood"
                print(json.dumps(part), flush=True)
            os._exit(3)                       # (a fired watchdog is a failed run, whatever part of the record could be printed)
        dog = threading.Timer(int(os.environ.get("EFTS_BENCH_DP_TIMEOUT", "240")), fire)
        dog.daemon = True
        dog.start()
    res = measure_train(a, world, rank, dev, wl, a.steps, a.warmup)
    if dog is not None:
        dog.cancel()
    if rank == 0:
        if cpu_baseline_fn is not None and world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_fn(wl["T1"], wl["T2"])
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
