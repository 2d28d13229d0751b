"""which cross-stream event patterns survive hipStreamEndCapture? (origin O, forked B, forked C)"""
import subprocess, sys
if len(sys.argv) > 1:
    import torch
    mode = sys.argv[1]
    dev = torch.device("cuda:0")
    x = [torch.zeros(1 << 20, device=dev) for _ in range(4)]
    B, C = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        O = torch.cuda.current_stream()
        x[0].add_(1)
        B.wait_stream(O)
        if "flat" in mode: C.wait_stream(O)
        with torch.cuda.stream(B):
            if "first" in mode: x[1].add_(1)
            C.wait_stream(B)
            with torch.cuda.stream(C):
                x[2].add_(1)
                evC = torch.cuda.Event(); evC.record(C)
            x[1].add_(1)
            if "xev" in mode:
                B.wait_event(evC)
                x[1].add_(1)
                evB = torch.cuda.Event(); evB.record(B)
                C.wait_event(evB)
                with torch.cuda.stream(C):
                    x[2].add_(1)
            if "many" in mode:
                for i in range(40):
                    x[1].add_(1)
                    e1 = torch.cuda.Event(); e1.record(B); C.wait_event(e1)
                    with torch.cuda.stream(C):
                        x[2].add_(1)
                        e2 = torch.cuda.Event(); e2.record(C)
                    B.wait_event(e2)
            if "nojoinbc" not in mode: B.wait_stream(C)
        O.wait_stream(B)
        if "joinc" in mode: O.wait_stream(C)
        x[0].add_(1)
    g.replay(); torch.cuda.synchronize()
    print(mode, "ok", [float(t[0]) for t in x])
    sys.exit(0)
for mode in ("nojoinbc-joinc", "nojoinbc-joinc-flat", "nojoinbc-joinc-xev", "nojoinbc-joinc-flat-xev", "nojoinbc-joinc-flat-first-many"):
    r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
    print(f"{mode}: rc={r.returncode} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ''}", flush=True)
