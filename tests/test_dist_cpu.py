"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce plumbing (efficient_tts_amd/dist.py)
and the gradient layout it relies on.  The kernels need an MI355X; here the gradient source is the
oracle's autograd, which is enough to check that N ranks averaging bucket-by-bucket reproduce the
mean of the per-rank gradients (the reference's DDP semantics, nntts/bin/train.py:210-216)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, golden, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.dist import BucketReducer
    from efficient_tts_amd.train import grad_layout
    from oracle import efts_oracle as O
    g = np.load(os.path.join(golden, "fwd_tiny.npz"))
    # each rank takes one item of the tiny batch (DistributedSampler-style shard)
    sl = slice(rank, rank + 1)
    T1, T2 = int(g["text_lengths"][rank]), int(g["speech_lengths"][rank])
    args = [torch.from_numpy(g["text"][sl, :T1]), torch.from_numpy(g["text_lengths"][sl]),
            torch.from_numpy(g["speech"][sl, :T2]), torch.from_numpy(g["speech_lengths"][sl])]
    P = {k: v.clone().requires_grad_(True) for k, v in O.fill_params().items()}
    O.forward(P, *args)["loss"].backward()
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01)
    layout = grad_layout(m)
    flat = torch.cat([P[n].grad.reshape(-1) for n, _ in layout])
    local = flat.clone()
    numel = flat.numel()
    ends, acc = [], 0
    for n, p in layout:                      # same three stages as TrainEngine.bucket_ends
        acc += p.numel()
        if n.startswith("decoder.layers.0.") or n.startswith("mel_prenet."):
            last = acc
        if n == [k for k, _ in layout if k.startswith("decoder.layers.0.")][-1] or n == [k for k, _ in layout if k.startswith("mel_prenet.")][-1]:
            ends.append(acc)
    ends.append(numel)
    flat2 = flat.clone()
    red = BucketReducer(flat, ends)
    for i in range(len(ends)):
        red.reduce(i)
    red.finish()
    red.finish()                                  # idempotent: the autograd hook and the trainer both join
    red2 = BucketReducer(flat2, ends, algo="rs_ag")   # reduce_scatter + all_gather per bucket (bucket sizes are not multiples of 2)
    for i in range(len(ends)):
        red2.reduce(i)
    red2.finish()
    same = bool(torch.allclose(flat, flat2, rtol=0, atol=1e-7))
    flat /= world
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ref = sum(gathered) / world
    ok = bool(torch.allclose(flat, ref, rtol=0, atol=1e-7)) and ends[-1] == 20587601 and len(ends) == 3 and same
    if rank == 0:
        open(out, "w").write("ok" if ok else f"mismatch {float((flat - ref).abs().max())} {ends}")
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo(golden_dir, tmp_path):
    out = str(tmp_path / "res.txt")
    mp.spawn(_worker, args=(2, _free_port(), golden_dir, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker8(rank, world, port, ends, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from efficient_tts_amd.dist import BucketReducer
    numel = ends[-1]
    # integer-valued "gradients" (exact in fp32 under any summation order): element i of rank r = (i mod 251) - 125 + 3 r
    base = (torch.arange(numel, dtype=torch.int64) % 251 - 125).float()
    want = base * world + 3.0 * sum(range(world))
    ok, notes = True, []
    for algo in ("allreduce", "rs_ag"):
        flat = base + 3.0 * rank
        red = BucketReducer(flat, ends, algo=algo)
        if algo == "rs_ag":      # every bucket is padded to a multiple of the world size; none of the real buckets is one
            notes.append([int(p.numel()) for p in red.pad])
            ok &= all(p.numel() % world == 0 and p.numel() - (e - s) < world for p, s, e in zip(red.pad, red.starts, red.ends))
        for i in reversed(range(len(ends))):          # any hand-over order: buckets are independent slices
            red.reduce(i)
        red.finish()
        ok &= bool(torch.equal(flat, want))
    if rank == 0:
        open(out, "w").write("ok" if ok else f"mismatch {notes}")
    dist.destroy_process_group()


def test_bucketed_exchange_eight_ranks_gloo(tmp_path):
    """world 8 (the node BASELINE config 4 names) over gloo: both exchange algorithms on the model's REAL bucket boundaries -- 20 587 601
    gradients, an odd number, in three buckets of which the middle one is odd, and then a layout whose three buckets leave remainders 1, 3
    and 7 modulo 8, so `rs_ag` runs on its padded staging buffers with ragged tails -- must produce the exact sum on every element (integer-valued inputs: exact under any order)."""
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.train import grad_layout
    m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01)
    layout = grad_layout(m)
    last_dec = [k for k, _ in layout if k.startswith("decoder.layers.0.")][-1]
    last_pre = [k for k, _ in layout if k.startswith("mel_prenet.")][-1]
    ends, acc = [], 0
    for n, p in layout:                      # the three stages of TrainEngine.bucket_ends
        acc += p.numel()
        if n in (last_dec, last_pre):
            ends.append(acc)
    ends.append(acc)
    assert ends[-1] == 20587601 and len(ends) == 3
    sizes = [e - s for s, e in zip([0] + ends[:-1], ends)]
    assert any(sz % 8 for sz in sizes), sizes             # the premise: the real layout has a bucket (the middle one, 5 553 153) with a ragged tail
    out = str(tmp_path / "res8.txt")
    mp.spawn(_worker8, args=(8, _free_port(), ends, out), nprocs=8, join=True)
    assert open(out).read() == "ok"
    # and a layout in which EVERY bucket leaves a different remainder modulo 8 (1, 3, 7 elements short of / over a multiple)
    ragged = [1000001, 1000001 + 2000003, 1000001 + 2000003 + 1500007]
    assert [(e - s) % 8 for s, e in zip([0] + ragged[:-1], ragged)] == [1, 3, 7]
    mp.spawn(_worker8, args=(8, _free_port(), ragged, out), nprocs=8, join=True)
    assert open(out).read() == "ok"


def test_grad_layout_is_backward_completion_order():
    from efficient_tts_amd import EfficientTTSCNN
    from efficient_tts_amd.train import grad_layout
    m = EfficientTTSCNN(num_symbols=76, use_masking=True)
    names = [n for n, _ in grad_layout(m)]
    assert names[0].startswith("mel_output_layer") and names[-1] == "text_embedding_table.weight"
    assert names.index("decoder.layers.5.conv.0.bias") < names.index("decoder.layers.0.conv.0.bias")
    assert names.index("decoder.layers.0.conv.0.bias") < names.index("duration_predictor.linear.weight")
    assert names.index("mel_prenet.0.weight") < names.index("text_encoder_value.weight")
    assert sorted(names) == sorted(n for n, _ in m.named_parameters())


def test_warmup_lr_matches_closed_form():
    from efficient_tts_amd.optim import WarmupLR
    from oracle import efts_oracle as O
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    sch = WarmupLR(opt, warmup_steps=4000)
    for step in range(1, 6):
        assert abs(opt.param_groups[0]["lr"] - O.warmup_lr(1e-3, step, 4000)) < 1e-15
        opt.step()
        sch.step()


def test_dp_selfcheck_fingerprint_and_rccl_log_parsing(tmp_path, monkeypatch):
    """bench.py --workload train32 --gpus N validates itself before timing (efficient_tts_amd/bench_train.py): the exact
    parameter fingerprint that is all-gathered across ranks must notice a 1-ulp difference and a permutation, and the RCCL log
    reader must pick the topology / algorithm lines out of an NCCL_DEBUG_FILE."""
    from efficient_tts_amd.bench_train import _params_fingerprint, _rccl_log_lines
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    dev = torch.device("cpu")
    f0 = _params_fingerprint(m, dev)
    assert torch.equal(f0, _params_fingerprint(m, dev))
    with torch.no_grad():
        w = m[0].weight
        w.view(-1)[3] = torch.nextafter(w.view(-1)[3], torch.tensor(10.0))           # one ulp
    f1 = _params_fingerprint(m, dev)
    assert not torch.equal(f0, f1)
    with torch.no_grad():
        a, b = w.view(-1)[0].clone(), w.view(-1)[1].clone()
        w.view(-1)[0], w.view(-1)[1] = b, a                                          # same multiset, different order
    f2 = _params_fingerprint(m, dev)
    assert int(f2[0]) == int(f1[0]) and int(f2[1]) != int(f1[1])
    log = tmp_path / "rccl_host_123.log"
    log.write_text("host:123:1 [0] NCCL INFO NCCL version 2.22.3+hip6.3\n"
                   "host:123:1 [0] NCCL INFO Channel 00/08 : 0 1 2 3 4 5 6 7\n"
                   "host:123:1 [0] NCCL INFO something unrelated\n"
                   "host:123:1 [0] NCCL INFO AllReduce: 36441248 Bytes -> Algo 1 proto 2 time 310.2\n")
    monkeypatch.setenv("NCCL_DEBUG_FILE", str(tmp_path / "rccl_%h_%p.log"))
    lines = _rccl_log_lines()
    assert lines and any("Algo 1 proto 2" in ln for ln in lines) and not any("unrelated" in ln for ln in lines)
    monkeypatch.delenv("NCCL_DEBUG_FILE")
    assert _rccl_log_lines() is None


def test_dp_record_states_its_own_scaling_efficiency():
    """round 5: the arithmetic of the data-parallel record (bench_train.exchange_model): efficiency = step without the exchange / step with
    it, and per bucket the all-reduce bus bandwidth 2 (N - 1) / N * bytes / time against one xGMI link and against all seven"""
    from efficient_tts_amd.bench_train import exchange_model, XGMI_LINKS, XGMI_LINK_GBPS
    stats = dict(bucket_mb=[36.4, 25.2, 20.8], bucket_ms=[0.40, 0.30, 0.25], exposed_ms=0.12)
    rec = exchange_model(stats, with_ms=3.60, without_ms=3.42, world=8)
    assert abs(rec["efficiency"] - 0.95) < 1e-9 and rec["step_ms_with_exchange"] == 3.60 and rec["step_ms_no_exchange"] == 3.42
    want = [1.75 * 36.4 / 0.40, 1.75 * 25.2 / 0.30, 1.75 * 20.8 / 0.25]
    assert all(abs(a - b) < 1e-9 for a, b in zip(rec["bucket_busbw_gbps"], want))
    assert abs(rec["bucket_busbw_frac_of_one_link"][0] - want[0] / XGMI_LINK_GBPS) < 1e-12
    assert abs(rec["bucket_busbw_frac_of_all_links"][2] - want[2] / (XGMI_LINKS * XGMI_LINK_GBPS)) < 1e-12
    assert rec["xgmi"]["all_links_gbps"] == XGMI_LINKS * XGMI_LINK_GBPS
    one = exchange_model({}, 3.0, 3.0, 1)                                   # a one-rank group: no bus figures, efficiency 1
    assert one["efficiency"] == 1.0 and "bucket_busbw_gbps" not in one
    assert exchange_model(dict(bucket_mb=[1.0], bucket_ms=[0.0]), 1.0, 1.0, 2)["bucket_busbw_gbps"] == [None]


def test_bench_launches_its_own_ranks_and_refuses_fewer():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (it used to run ONE rank silently and
    print n_gpus 1).  The `rendezvous` workload is the launch path only: every rank joins the group (gloo here, nccl on a GPU
    box), one all-reduce counts them.  A launcher that started a different number of ranks than --gpus asks for is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "rendezvous"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["self_launched"] is True
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "rendezvous"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_bench_watchdog_prints_the_line_and_ends_the_process():
    """N > 1: a data-parallel record that hangs (a rank that failed alone leaves the others inside a collective) must not take the
    forward result with it: bench.py's watchdog prints rank 0's line with the record marked as abandoned and ends the process."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; res = dict(metric='m', value=1.0)\n"
            "bench._watchdog(1, 0, res, 'train32'); time.sleep(60); print('not reached')" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 3 and "not reached" not in out.stdout, out.stderr[-1000:]      # (a fired watchdog is a failed run: non-zero)
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["value"] == 1.0 and "abandoned after 1 s" in rec["train32"]["error"]
    quiet = subprocess.run([sys.executable, "-c", code.replace("(1, 0, res", "(1, 1, res")], capture_output=True, text=True, timeout=300)
    assert quiet.returncode == 3 and quiet.stdout.strip() == ""           # other ranks: no line, same exit
    code2 = code.replace("time.sleep(60)", "t = bench._watchdog(1, 0, res, 'x'); t.cancel(); [w.cancel() for w in __import__('threading').enumerate() if hasattr(w, 'cancel')]; time.sleep(2)")
    kept = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300)
    assert "not reached" in kept.stdout                                   # a cancelled watchdog does nothing
