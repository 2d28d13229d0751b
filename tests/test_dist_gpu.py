"""GPU (-m gpu): the data-parallel path on the RCCL backend itself.

No multi-GPU box is in reach of the test run, so the `nccl` process group is opened with ONE rank on the one GPU: every
`dist.all_reduce` / `reduce_scatter_tensor` / `all_gather_into_tensor` of efficient_tts_amd/dist.py then goes through RCCL (a
one-rank sum is the identity), on the reducer's communication stream, from the training engine's bucket hooks -- the code that
runs under `torch.distributed.run` on an 8-GPU node (reference: nntts/bin/train.py:53-68, :210-216), including the step captured
as one hipGraph with the collectives inside (step_graph.GraphedStep).  The 2-rank arithmetic (mean of means, lock-step replicas)
is covered over gloo in tests/test_gpu_train.py and tests/test_dist_cpu.py."""
import json
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.dist import BucketReducer, DistributedEFTS
    from efficient_tts_amd.optim import EftsAdam
    from efficient_tts_amd.step_graph import GraphedStep
    rec = dict(backend=dist.get_backend(), world=dist.get_world_size())

    # ---- the reducer alone: a one-rank exchange is the identity, bit for bit, for both algorithms; its events make a record
    torch.manual_seed(0)
    for algo in ("allreduce", "rs_ag"):
        flat = torch.randn(1000003, device=dev)
        want = flat.clone()
        red = BucketReducer(flat, [300001, 700000, 1000003], algo=algo, timing=True)
        red.mark("backward_start")
        for i in range(3):
            flat[red.starts[i]:red.ends[i]].mul_(1.0)            # (compute-stream work in front of the hand-over)
            red.reduce(i)
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(flat, want), algo
        st = red.stats()
        assert st["algo"] == algo and st["world"] == 1 and len(st["bucket_ms"]) == 3 and all(v >= 0 for v in st["bucket_ms"]), st
        assert st["exposed_ms"] >= 0 and st["backward_ms"] >= 0 and st["first_bucket_after_ms"] >= 0, st
        rec[f"reducer_{algo}"] = st

    # ---- the training step through DistributedEFTS with the reducer forced on, against the wrapper-free step from the same state
    B, T1, T2 = 3, 40, 130
    gen = torch.Generator().manual_seed(7)
    batch = (torch.randint(0, 76, (B, T1), generator=gen).to(dev), torch.tensor([40, 33, 21]).to(dev),
             torch.randn(B, T2, 80, generator=gen).to(dev), torch.tensor([130, 90, 64]).to(dev))
    for algo in ("allreduce", "rs_ag"):
        torch.manual_seed(1)
        m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision="bf16").to(dev).train()
        opt = EftsAdam(m, lr=1e-4, betas=(0.9, 0.99), eps=1e-9, weight_decay=1e-5, amsgrad=True, grad_norm=1.0)
        ddp = DistributedEFTS(m, algo=algo, timing=True, force_reducer=True)
        assert ddp.reducer is not None and ddp.grad_scale == 1.0
        eng = ddp.engine
        hooks = (eng.bucket_hook, eng.join_reduce, eng.mark)
        assert all(h is not None for h in hooks)

        def one_step(net):
            loss, stats, *_ = net(text=batch[0], text_lengths=batch[1], speech=batch[2], speech_lengths=batch[3])
            opt.zero_grad()
            loss.backward()
            if net is ddp:
                ddp.finish_reduce()
            opt.step(grad_scale=1.0)
            return float(loss)

        one_step(ddp)                                             # warm-up (workspaces, plans)
        state = (opt.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.vmax.clone(), opt.t, m.dropout_calls)

        def restore():
            opt.flat_p.copy_(state[0]); opt.m.copy_(state[1]); opt.v.copy_(state[2]); opt.vmax.copy_(state[3])
            opt.t, m.dropout_calls, m._packed_sig = state[4], state[5], None

        l_dp = one_step(ddp)
        torch.cuda.synchronize()
        st = ddp.reducer.stats()
        assert len(st["bucket_ms"]) == 3 and abs(sum(st["bucket_mb"]) - 82.35) < 0.01 and st["world"] == 1, st
        g_dp, p_dp = eng.flat.clone(), opt.flat_p.clone()
        restore()
        eng.bucket_hook, eng.join_reduce, eng.mark = None, None, None       # the wrapper-free step
        l_pl = one_step(m)
        g_pl, p_pl = eng.flat.clone(), opt.flat_p.clone()
        eng.bucket_hook, eng.join_reduce, eng.mark = hooks
        # (two runs of the SAME eager step differ by ~2e-8 in the gradients: float atomics in the backward; see
        #  test_graphed_training_step_equals_the_eager_loop)
        assert l_dp == l_pl, (l_dp, l_pl)
        assert float((g_dp - g_pl).double().norm()) <= 1e-6 * float(g_pl.double().norm())
        d = (p_dp - p_pl).abs()
        assert float(d.max()) <= 2.05e-4 and float((d <= 1e-7).float().mean()) >= 0.999
        rec[f"step_{algo}"] = dict(loss=l_dp, stats=st)

        # ---- the same step as ONE hipGraph replay, collectives captured with it
        step = GraphedStep(ddp, opt, None)
        assert step.grad_scale == 1.0 and step.model is m
        restore()
        step(*batch)                                              # first call of the shape: eager
        step(*batch)                                              # capture + first replay
        ent = next(iter(step.entries.values()))
        assert ent["graph"] is not None and not ent.get("eager_only"), "the data-parallel step was not captured"
        assert step.replays == 1
        for _ in range(2):
            st0 = (opt.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.vmax.clone(), opt.t, m.dropout_calls)

            def back():
                opt.flat_p.copy_(st0[0]); opt.m.copy_(st0[1]); opt.v.copy_(st0[2]); opt.vmax.copy_(st0[3])
                opt.t, m.dropout_calls, m._packed_sig = st0[4], st0[5], None
            le, _ = step._eager(*batch)
            ge, pe = eng.flat.clone(), opt.flat_p.clone()
            back()
            n0 = step.replays
            lg, _ = step(*batch)
            assert step.replays == n0 + 1
            gg, pg = eng.flat.clone(), opt.flat_p.clone()
            assert float(le) == float(lg)
            assert float((ge - gg).double().norm()) <= 1e-6 * float(ge.double().norm())
            d = (pe - pg).abs()
            assert float(d.max()) <= 2.05e-4 and float((d <= 1e-7).float().mean()) >= 0.999
        rec[f"graph_{algo}"] = dict(replays=step.replays)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    with open(out, "w") as f:
        json.dump(rec, f)


def test_rccl_one_rank_group_runs_the_bucketed_exchange_eager_and_captured(tmp_path):
    """BucketReducer on the `nccl` (= RCCL) backend, both algorithms: identity exchange bit for bit, stats() record, the training
    step through the forced reducer == the wrapper-free step, and GraphedStep over the wrapper == the eager loop."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl1.json")
    mp.spawn(_worker, args=(port, out), nprocs=1, join=True)
    rec = json.load(open(out))
    assert rec["backend"] == "nccl" and rec["world"] == 1
    for algo in ("allreduce", "rs_ag"):
        assert rec[f"graph_{algo}"]["replays"] >= 3 and len(rec[f"step_{algo}"]["stats"]["bucket_ms"]) == 3


def _bench_worker(rank, port, out):
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from efficient_tts_amd.bench_train import measure_train
    a = types.SimpleNamespace(precision="bf16", train_set=[], dp_algo="allreduce", train_graph=1, allow_gloo=True)
    wl = dict(B=3, T1=40, T2=130, desc="test shape")
    rec = measure_train(a, 2, rank, dev, wl, steps=3, warmup=2)
    if rank == 0:
        with open(out, "w") as f:
            json.dump(rec, f)
    else:
        assert rec is None
    dist.destroy_process_group()


def test_bench_dp_record_two_ranks():
    """bench.py's data-parallel measurement (efficient_tts_amd/bench_train.measure_train at world > 1: what the driver's N > 1 runs
    execute on every rank) driven end to end by two ranks that share the test GPU -- gloo, the only backend that allows that: the
    pre-timing self-check (bit-exact replica fingerprints), the eager loop with per-bucket events, the attempt to capture the step with
    its collectives (gloo cannot be captured: both ranks must agree on that through the all-reduce and fall back to the eager number,
    without hanging), and the record with its `dp` block."""
    import tempfile
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rec.json")
        mp.spawn(_bench_worker, args=(port, out), nprocs=2, join=True)
        rec = json.load(open(out))
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["steps"] == 3
    dp = rec["dp"]
    assert dp["ranks"] == 2 and dp["selfcheck"]["replicas_bit_identical"] and len(dp["bucket_ms"]) == 3 and abs(sum(dp["bucket_mb"]) - 82.35) < 0.01
    assert rec["eager_ms_per_step"] > 0 and rec["value"] > 0
    # round 5: the record states its own scaling efficiency (the same steps with the exchange switched off, state restored afterwards)
    assert dp["step_ms_no_exchange"] > 0 and abs(dp["step_ms_with_exchange"] - rec["eager_ms_per_step"]) < 1e-6
    assert abs(dp["efficiency"] - dp["step_ms_no_exchange"] / dp["step_ms_with_exchange"]) < 1e-9 and 0.0 < dp["efficiency"] < 1.5      # (gloo through the host on one shared GPU: a few per cent)
    assert len(dp["bucket_busbw_gbps"]) == 3 and all(b > 0 for b in dp["bucket_busbw_gbps"]) and dp["xgmi"]["links_per_gpu"] == 7
    if rec["graph_ms_per_step"] is None:                      # (gloo: the capture fails on every rank alike)
        assert "eager" in rec["config"]["step_issue"]
    else:
        assert dp["selfcheck"]["replicas_bit_identical_after_graph_replays"]


def _bench_worker8(rank, port, out):
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=8)
    from efficient_tts_amd.bench_train import measure_train
    for algo in ("allreduce", "rs_ag"):
        a = types.SimpleNamespace(precision="bf16", train_set=[], dp_algo=algo, train_graph=0, allow_gloo=True)
        rec = measure_train(a, 8, rank, dev, dict(B=2, T1=24, T2=70, desc="test shape"), steps=2, warmup=2)
        if rank == 0:
            with open(out + "." + algo, "w") as f:
                json.dump(rec, f)
        else:
            assert rec is None
    dist.destroy_process_group()


def test_bench_dp_record_eight_ranks():
    """world 8 -- the node BASELINE config 4 names -- through the data-parallel measurement, eight ranks sharing the test GPU over gloo at
    a reduced batch, both exchange algorithms (20 587 601 gradients: `rs_ag` pads its odd middle bucket to a multiple of 8): eight distinct
    seeds and batches, replicas bit-identical after the optimisation steps, the record's own efficiency and exposed-exchange figures."""
    import tempfile
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rec.json")
        mp.spawn(_bench_worker8, args=(port, out), nprocs=8, join=True)
        recs = {algo: json.load(open(out + "." + algo)) for algo in ("allreduce", "rs_ag")}
    for algo, rec in recs.items():
        dp = rec["dp"]
        assert rec["n_gpus"] == 8 and rec["config"]["parallelism"] == "dp8" and dp["ranks"] == 8 and dp["algo"] == algo
        sc = dp["selfcheck"]
        assert sc["replicas_bit_identical"] and sc["seeds"] == [1234 + r for r in range(8)] and sc["batches_distinct"]
        assert len(dp["bucket_ms"]) == 3 and abs(sum(dp["bucket_mb"]) - 82.35) < 0.01
        assert dp["step_ms_no_exchange"] > 0 and 0.0 < dp["efficiency"] < 1.5 and dp["exposed_ms"] >= 0
        assert abs(dp["efficiency"] - dp["step_ms_no_exchange"] / dp["step_ms_with_exchange"]) < 1e-9
        # 2 (N - 1) / N = 1.75 at N = 8 in the bus-bandwidth figures
        assert all(abs(b - 1.75 * m / t) < 1e-6 for b, m, t in zip(dp["bucket_busbw_gbps"], dp["bucket_mb"], dp["bucket_ms"]))
    # the two algorithms average the same gradients: same parameters after the same steps (sums of 8 fp32 values in different orders: equal
    # up to rounding, so the fingerprints may differ; the losses of the last step agree)
    assert abs(recs["allreduce"]["loss"] - recs["rs_ag"]["loss"]) <= 1e-4 * abs(recs["allreduce"]["loss"])


def test_bench_two_ranks_end_to_end():
    """`python bench.py --gpus 2` as the driver's multi-GPU runs execute it -- self-launched ranks, replica forward in both precisions,
    then the data-parallel training record on every rank -- on the ONE GPU of the test box: EFTS_BENCH_BACKEND=gloo lets the two ranks share
    it (RCCL needs a GPU per rank).  One JSON line, n_gpus 2, the `train32` record with its `dp` block and both precisions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EFTS_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["config"]["parallelism"] == "replicas x2" and rec["scaling"] == "weak"
    assert rec["parity_mode"]["ms_per_step"] > rec["ms_per_step"] > 0
    tr = rec["train32"]
    assert tr["n_gpus"] == 2 and tr["config"]["parallelism"] == "dp2" and tr["dp"]["ranks"] == 2 and tr["dp"]["selfcheck"]["replicas_bit_identical"]
    assert len(tr["dp"]["bucket_ms"]) == 3 and tr["parity_mode"]["ms_per_step"] > 0
    # the collective path lifted to the top level of the N > 1 line
    assert rec["dp_value"] == tr["value"] and rec["dp_ms_per_step"] == tr["ms_per_step"] and rec["dp_efficiency"] == tr["dp"]["efficiency"] > 0
    assert rec["dp_exposed_ms"] == tr["dp"]["exposed_ms"] and "exchange" in rec["dp_note"]


def test_bench_eight_ranks_end_to_end():
    """`python bench.py --gpus 8` -- the driver's widest multi-GPU run -- with the eight self-launched ranks sharing the ONE test GPU over gloo
    (EFTS_BENCH_BACKEND=gloo; RCCL needs a GPU per rank) at a reduced shape (--test-shape): one JSON line, n_gpus 8, eight replicas of the
    forward, and the data-parallel record lifted to the top level: dp_efficiency, dp_exposed_ms, eight distinct seeds / batches, replicas
    bit-identical after the optimisation steps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EFTS_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity-mode", "0",
                          "--test-shape", "2,24,70"], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["parallelism"] == "replicas x8" and "TEST SHAPE" in rec["config"]["workload"]
    tr = rec["train32"]
    dp = tr["dp"]
    assert tr["n_gpus"] == 8 and tr["config"]["parallelism"] == "dp8" and dp["ranks"] == 8 and tr["steps"] == 2
    assert dp["selfcheck"]["replicas_bit_identical"] and dp["selfcheck"]["seeds"] == [1234 + r for r in range(8)] and dp["selfcheck"]["batches_distinct"]
    assert rec["dp_efficiency"] == dp["efficiency"] > 0 and rec["dp_exposed_ms"] == dp["exposed_ms"] >= 0 and rec["dp_value"] == tr["value"] > 0
