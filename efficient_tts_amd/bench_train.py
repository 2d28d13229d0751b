"""bench.py --workload train32: the data-parallel training step (BASELINE configs 3 and 4):
fwd + bwd + clip(1.0) + Adam-amsgrad + WarmupLR at batch 32 per GPU, (T1, T2) = (128, 800), with the
gradient all-reduce over RCCL overlapped with backward when world > 1
(reference: nntts/trainers/efficient_tts_trainer.py:139-160 under nntts/bin/train.py:210-216)."""
import json
import os
import time

import torch

TRAIN_FLOP_PER_ITEM = 3 * 21.43e9


def cpu_train_baseline(T1, T2):
    """the oracle (CPU restatement of the reference path, torch autograd for the backward)
    timed on this box's host cores for one training step -- fwd + bwd + clip 1.0 + Adam-amsgrad -- on a bounded sample
    (B=4 full-length items instead of 32)."""
    import time as _t
    from oracle import efts_oracle as O          # the cpu_baseline leg: oracle as the thing timed
    cores = os.cpu_count() or 1
    nt = min(cores, 16)
    torch.set_num_threads(nt)
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in O.fill_params().items()}
    params = [v for v in P.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True)
    Bc = 4
    g = torch.Generator().manual_seed(1234)
    text = torch.randint(0, 76, (Bc, T1), generator=g)
    mel = torch.randn(Bc, T2, 80, generator=g)
    tl = torch.full((Bc,), T1, dtype=torch.int64)
    sl = torch.full((Bc,), T2, dtype=torch.int64)
    times = []
    for it in range(4):
        t0 = _t.perf_counter()
        out = O.forward(P, text, tl, mel, sl)
        opt.zero_grad()
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        if it:
            times.append(_t.perf_counter() - t0)
    med = sorted(times)[1]
    return dict(value=Bc * T2 / med, unit="mel-frames/s", cores=nt, host_cpus=cores, kind="port",
                sample=f"oracle training step fp32 (forward, autograd backward, clip 1.0, torch Adam-amsgrad), B={Bc} x (T1={T1}, T2={T2}), "
                       f"median of 3 after 1 warm-up ({med:.3f} s/step at {nt} threads)")


def run_train(a, world, rank, dev, wl):
    import torch.distributed as dist
    from . import EfficientTTSCNN, ops as P
    from .dist import DistributedEFTS
    from .optim import EftsAdam, WarmupLR
    B, T1, T2 = wl["B"], wl["T1"], wl["T2"]
    for kv in getattr(a, "train_set", []):
        from . import train as _tr
        k, v = kv.split("=")
        assert hasattr(_tr, k), f"efficient_tts_amd.train has no switch {k}"
        setattr(_tr, k, int(v))
    torch.manual_seed(0)
    model = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, use_weighted_masking=False, sigma=0.01,
                            precision=a.precision).to(dev).train()
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

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    rows = P.Rows(B, T2).rows
    P.PROFILE, P.PROFILE_TAG = [], (5, rows, 512)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lv = float(loss)
    assert lv == lv, "NaN loss"
    durs = [s.elapsed_time(e) * 1e-3 for (tag, s, e) in P.PROFILE if tag == (5, rows, 512)]
    P.PROFILE, P.PROFILE_TAG = None, None
    avg = sum(durs) / max(len(durs), 1)
    conv_flop = 2.0 * B * T2 * 512 * 512 * 5
    split = model.split
    big = split == 1 and ((rows + 251) // 252) * 4 >= 400           # efts_gemm's own rule for the 256-row kernel
    kname = "conv5_kernel<split=1> (256-row tiles)" if big else f"gemm_kernel<taps=5,split={split}> (124-row tiles)"
    cpu = cpu_train_baseline(T1, T2) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
    if rank == 0:
        frames = world * B * T2 * a.steps
        res = dict(metric="mel-frames/sec (EFTS-CNN training step, batch 32/GPU, 80-mel LJSpeech shape)", value=frames / dt,
                   unit="mel-frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="bf16" if a.precision == "bf16" else "bf16x3 (split-bf16 MFMA, fp32-class)", data="synthetic",
                   config=dict(workload=wl["desc"], batch_per_gpu=B, phoneme_len=T1, mel_len=T2, precision=a.precision,
                               parallelism=f"dp{world}", optimizer="Adam-amsgrad fused, clip 1.0, WarmupLR 4000",
                               allreduce="RCCL, 3 buckets overlapped with backward" if world > 1 else "none"),
                   per_gpu=frames / dt / world, tflops=TRAIN_FLOP_PER_ITEM * B * world * a.steps / dt / 1e12, loss=lv,
                   roofline=dict(bound="mfma", kernel=f"{kname}: fwd + dgrad launches at mel length (timed while the text-length stream runs beside them)",
                                 achieved=conv_flop / avg / 1e12 if avg else None, peak=2500.0, unit="TFLOP/s",
                                 frac=conv_flop / avg / 1e12 / 2500.0 if avg else None, traffic=None,
                                 avg_launch_us=avg * 1e6, launches_measured=len(durs)))
        if ddp is not None:
            # the last timed step's communication: one record that explains the scaling number (backend, ranks, algorithm,
            # bytes and time per bucket, how much of it the backward did NOT hide)
            res["dp"] = dict(backend=dist.get_backend(), ranks=dist.get_world_size(), grad_mb=ddp.engine.numel * 4 / 1e6,
                             nccl_env={k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                             **ddp.reducer.stats())
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
