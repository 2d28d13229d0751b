"""efts_resconv5: one B=64 chain on the whole chip against two B=32 chains on two streams, each planned for half the CUs
(63 groups x 2 column tiles = 126 workgroups of two tiles), started half a layer apart: do the epilogue bursts of one chain hide
under the main loops of the other?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from efficient_tts_amd import lib as L, ops as P
dev = torch.device("cuda:0")
L.load(); L.require_device()
C, T, N = 512, 800, int(os.environ.get("PN", 48))
cus = torch.cuda.get_device_properties(dev).multi_processor_count

def chain(B, split):
    rs = P.Rows(B, T)
    def plane(sp):
        pl = P.Plane.for_rows(rs, C, sp, dev)
        pl.buf.copy_(torch.randn(pl.buf.numel() // 2, device=dev).mul_(0.5).to(torch.bfloat16).view(torch.uint8).view(pl.buf.shape))
        return pl
    xa, xb = plane(split), plane(split)
    la, lb = (plane(1), plane(1)) if split == 1 else (None, None)
    pw = P.PackedWeight(C, C, 5, split, dev); pw.pack((torch.randn(C, C, 5, device=dev) * 0.02).contiguous())
    bias = torch.randn(C, device=dev); gap = torch.ones(rs.rows, device=dev)
    def run(n, plan=None):
        for i in range(n):
            src, dst, sl, dl = (xa, xb, la, lb) if i % 2 == 0 else (xb, xa, lb, la)
            P.resconv5(x=src, x_lo=sl, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=dst, y_lo=dl, plan=plan)
    return rs, run

for split in (1, 2):
    rs64, run64 = chain(64, split)
    with P.stream_scope():
        run64(4); torch.cuda.synchronize()
        t0 = time.perf_counter(); run64(N); torch.cuda.synchronize()
        base = (time.perf_counter() - t0) / N * 1e6
    print(f"split {split}: one chain B=64, automatic plan {P.resconv5_plan(rs64.rows, C, cus)}: {base:.1f} us per layer", flush=True)
    halves = [chain(32, split) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    # the same two chains as parallel branches of ONE hipGraph
    plan = P.resconv5_plan_buf(halves[0][0].rows, C, cus)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        with P.stream_scope():
            main = torch.cuda.current_stream()
            streams[1].wait_stream(main)
            halves[0][1](N, plan)
            with P.on_stream(streams[1]):
                halves[1][1](N, plan)
            main.wait_stream(streams[1])
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e6
    print(f"  two chains B=32 as two branches of one hipGraph: {dt:.1f} us per layer pair ({base / dt:.3f}x)", flush=True)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
        with P.stream_scope():
            run64(N)
    g1.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g1.replay(); torch.cuda.synchronize()
    print(f"  one chain B=64 as a hipGraph: {(time.perf_counter() - t0) / N * 1e6:.1f} us per layer", flush=True)
    for hc in (cus,):
        plan = P.resconv5_plan_buf(halves[0][0].rows, C, hc)
        for offset in (0, 1):
            def go(n):
                for i, (rs, run) in enumerate(halves):
                    with torch.cuda.stream(streams[i]), P.stream_scope():
                        if i == 1 and offset:
                            run64(0)
                            torch.cuda._sleep(int(60e-6 * 2.4e9 / 1.0)) if hasattr(torch.cuda, "_sleep") else None
                        run(n, plan)
            go(2); torch.cuda.synchronize()
            t0 = time.perf_counter(); go(N); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N * 1e6
            print(f"  two chains B=32, plan for {hc} CUs {P.resconv5_plan(halves[0][0].rows, C, hc)}, second chain {'delayed' if offset else 'in phase'}: {dt:.1f} us per layer pair ({base / dt:.3f}x)", flush=True)
