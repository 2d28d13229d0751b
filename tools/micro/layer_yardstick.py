"""External yardstick for the dominant layer: the SAME contraction -- k5 Conv1d 512 -> 512 over 64 x 800 frames =
a GEMM of 51 200 x 2 560 x 512, 134.2 GFLOP -- on code this repo did not write, beside `efts_resconv5`, on one box, in one process.

    python tools/micro/layer_yardstick.py > gpurun_out/layer_yardstick.txt

Arms (random bf16 / fp32 data, N(0, 1) activations, 0.02 * N(0, 1) weights):
  matmul_nn      torch.matmul bf16 [51200, 2560] x [2560, 512]            (hipBLASLt / rocBLAS, whatever torch dispatches to)
  matmul_nt      torch.matmul bf16 [51200, 2560] x [512, 2560]^T          (the weight as nn.Linear stores it)
  matmul_big     torch.matmul bf16 [8192, 8192] x [8192, 8192]            (the vendor GEMM on a shape it likes: what the library sustains at all)
  conv1d_bf16    F.conv1d bf16 [64, 512, 800] * [512, 512, 5], pad 2      (MIOpen: what nntts/layers/efts_modules.py:48-51 executes under autocast)
  conv1d_fp32    F.conv1d fp32, same shape                                 (MIOpen: what the reference executes as written)
  resconv5_s1    efts_resconv5, bf16 operands, hi + lo planes in and out (bias + LeakyReLU + residual + row mask fused -- MORE work than the arms above)
  resconv5_s2    efts_resconv5, bf16x3 (three MFMAs per product)
Two passes:
  (1) timing: ROUNDS rounds, arms interleaved inside a round, LAUNCHES (200) launches per arm and round between two events after 20 warm-up launches;
      median / min over the rounds.
  (2) sustained + power: each arm alone in a loop of ~SUSTAIN seconds; socket power sampled beside it (`rocm-smi --showpower` as
      tools/gpu_power_trace.sh does, and the card's hwmon file as a second, faster signal), samples of the first second dropped; us per launch of
      the whole loop, W, mJ per launch.
Kernel names of the library arms: `YROUNDS=1 YLAUNCHES=20 YSUSTAIN=0.05 rocprofv3 --kernel-trace --stats ... -- python tools/micro/layer_yardstick.py`.
FLOP figure of every conv / layer arm: 2 * 51200 * 2560 * 512 = 134.2 GFLOP (matmul_big: 2 * 8192^3).
"""
import glob
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

ROUNDS = int(os.environ.get("YROUNDS", "5"))
LAUNCHES = int(os.environ.get("YLAUNCHES", "200"))
SUSTAIN = float(os.environ.get("YSUSTAIN", "4.0"))
dev = torch.device("cuda:0")
B, T, C, K = 64, 800, 512, 5
FLOP = 2.0 * B * T * C * C * K


class Power:
    """socket power of GPU 0 in W, from TWO sources side by side: `rocm-smi --showpower` (one fork per sample, ~3 per second: the figure
    tools/gpu_power_trace.sh and DESIGN.md 4a' quote) and, when the node exposes it, the hwmon file of the card (20 samples per second; on
    this pool's boxes it reads about HALF of rocm-smi's figure under load and does not drop at idle -- kept as a relative signal only)"""

    def __init__(self):
        self.path = None
        for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
            got = sorted(glob.glob(pat))
            if got:
                self.path = got[0]
                break
        self.source = f"rocm-smi --showpower; hwmon = {self.path}"

    def read_hwmon(self):
        try:
            return float(open(self.path).read()) * 1e-6 if self.path else None
        except Exception:                                  # noqa: BLE001
            return None

    @staticmethod
    def read_smi():
        try:
            out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"Power \(W\): ([0-9.]+)", out)
            return float(m.group(1)) if m else None
        except Exception:                                  # noqa: BLE001
            return None

    def sample_while(self, fn):
        """run fn() (blocking) while sampling; -> (fn's result, [(t, W)] of rocm-smi, [(t, W)] of hwmon)"""
        stop, smi, hw, t0 = threading.Event(), [], [], time.perf_counter()

        def loop(read, got, pause):
            while not stop.is_set():
                w = read()
                if w is not None:
                    got.append((time.perf_counter() - t0, w))
                time.sleep(pause)
        ths = [threading.Thread(target=loop, args=(self.read_smi, smi, 0.0), daemon=True)]
        if self.path:
            ths.append(threading.Thread(target=loop, args=(self.read_hwmon, hw, 0.05), daemon=True))
        for th in ths:
            th.start()
        res = fn()
        stop.set()
        for th in ths:
            th.join()
        return res, smi, hw


def arms():
    torch.manual_seed(0)
    out = {}
    a = torch.randn(B * T, C * K, device=dev).bfloat16()
    w_nn = (torch.randn(C * K, C, device=dev) * 0.02).bfloat16()
    w_nt = (torch.randn(C, C * K, device=dev) * 0.02).bfloat16()
    o = torch.empty(B * T, C, device=dev, dtype=torch.bfloat16)
    out["matmul_nn"] = (lambda: torch.matmul(a, w_nn, out=o), FLOP)
    out["matmul_nt"] = (lambda: torch.matmul(a, w_nt.t(), out=o), FLOP)
    big_a = torch.randn(8192, 8192, device=dev).bfloat16()
    big_b = torch.randn(8192, 8192, device=dev).bfloat16()
    big_o = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    out["matmul_big"] = (lambda: torch.matmul(big_a, big_b, out=big_o), 2.0 * 8192 ** 3)
    torch.backends.cudnn.benchmark = True            # nntts/bin/train.py:60 sets it: MIOpen picks its fastest solver for the shape
    x32 = torch.randn(B, C, T, device=dev)
    w32 = torch.randn(C, C, K, device=dev) * 0.02
    b32 = torch.randn(C, device=dev) * 0.01
    x16, w16, b16 = x32.bfloat16(), w32.bfloat16(), b32.bfloat16()
    out["conv1d_bf16"] = (lambda: F.conv1d(x16, w16, b16, padding=2), FLOP)
    out["conv1d_fp32"] = (lambda: F.conv1d(x32, w32, b32, padding=2), FLOP)
    # efts_resconv5 through the C ABI, as the model's decoder launches it (hi / lo planes in, hi / lo planes out)
    from efficient_tts_amd import lib as L, ops as P
    L.load()
    L.require_device()
    rs = P.Rows(B, T)
    xr = torch.randn(B, T, C, device=dev)
    gap = torch.zeros(rs.rows, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    P.row_masks(lens, rs, gap, None)
    bias = torch.randn(C, device=dev) * 0.01
    keep = []
    for split in (1, 2):
        hi = xr.to(torch.bfloat16)
        x16f = hi.float() + (xr - hi.float()).to(torch.bfloat16).float()
        pa = P.Plane.for_rows(rs, C, split, dev)
        P.pack_rows(x16f, None, pa, rs)
        pa_lo = None
        if split == 1:
            pa_lo = P.Plane.for_rows(rs, C, 1, dev)
            P.pack_rows((x16f - x16f.to(torch.bfloat16).float()).contiguous(), None, pa_lo, rs)
        pw = P.PackedWeight(C, C, K, split, dev)
        pw.pack((torch.randn(C, C, K, device=dev) * 0.02).contiguous())
        y = P.Plane.for_rows(rs, C, split, dev)
        yl = P.Plane.for_rows(rs, C, 1, dev) if split == 1 else None
        keep.append((pa, pa_lo, pw, y, yl))

        def fn(pa=pa, pa_lo=pa_lo, pw=pw, y=y, yl=yl):
            with P.stream_scope():
                P.resconv5(x=pa, x_lo=pa_lo, w=pw, m=rs.rows, n=C, bias=bias, rowmask_ptr=gap.data_ptr(), y=y, y_lo=yl)
        out[f"resconv5_s{split}"] = (fn, FLOP)
    out["_keep"] = (keep, 0)
    return out


def events_us(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    print(f"# layer yardstick: {torch.cuda.get_device_name(0)}, torch {torch.__version__}, hip {torch.version.hip}; "
          f"TORCH_BLAS_PREFER_HIPBLASLT={os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT', '(unset)')}, "
          f"preferred blas backend {torch.backends.cuda.preferred_blas_library()}", flush=True)
    A = arms()
    names = [k for k in A if not k.startswith("_")]
    for n in names:                                     # first calls: library solver search, code-object load
        t0 = time.perf_counter()
        for _ in range(3):
            A[n][0]()
        torch.cuda.synchronize()
        print(f"# first 3 calls of {n}: {time.perf_counter() - t0:.2f} s", flush=True)
    # ---- pass 1: interleaved timing
    per = {n: [] for n in names}
    for r in range(ROUNDS):
        for n in names:
            fn = A[n][0]
            for _ in range(20):
                fn()
            per[n].append(events_us(fn, LAUNCHES))
    print(f"\n## pass 1: {ROUNDS} rounds, arms interleaved, {LAUNCHES} launches per arm and round (us per launch: median, min; TFLOP/s at the median; of the 2.5 PF bf16 peak)")
    for n in names:
        ts = sorted(per[n])
        med = ts[len(ts) // 2]
        tf = A[n][1] / med * 1e-6
        print(f"{n:13s} median {med:8.1f} us  min {ts[0]:8.1f} us  {tf:7.0f} TFLOP/s  {tf / 2500.0:5.3f} of peak   rounds: " + " ".join(f"{t:.1f}" for t in per[n]), flush=True)
    # ---- pass 2: sustained loops with power
    pw = Power()
    torch.cuda.synchronize()
    time.sleep(5.0)
    _, idle_smi, idle_hw = pw.sample_while(lambda: time.sleep(2.0))
    mean = lambda xs: sum(w for _, w in xs) / len(xs) if xs else float("nan")   # noqa: E731
    print(f"\n## pass 2: each arm alone for ~{SUSTAIN:.0f} s, power from {pw.source}; idle (5 s after pass 1): rocm-smi {mean(idle_smi):.0f} W, hwmon {mean(idle_hw):.0f} W; "
          "samples of the first second of a loop dropped; mJ / pJ from the rocm-smi figure")
    try:
        cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
        for ln in cap.splitlines():
            if re.search(r"max|cap", ln, re.I) and "GPU" in ln:
                print("# " + ln.strip())
    except Exception:                                      # noqa: BLE001
        pass
    for n in names:
        fn = A[n][0]
        med = sorted(per[n])[len(per[n]) // 2]
        count = max(200, int(SUSTAIN * 1e6 / med))

        def loop(fn=fn, count=count):
            return events_us(fn, count)
        us, smi, hw = pw.sample_while(loop)
        ws = [w for (t, w) in smi if t > 1.0]
        hs = [w for (t, w) in hw if t > 1.0]
        wavg = sum(ws) / len(ws) if ws else float("nan")
        havg = sum(hs) / len(hs) if hs else float("nan")
        tf = A[n][1] / us * 1e-6
        print(f"{n:13s} {count:6d} launches  {us:8.1f} us per launch  {tf:7.0f} TFLOP/s  {tf / 2500.0:5.3f} of peak   rocm-smi {wavg:6.0f} W ({len(ws)} samples, max {max(ws) if ws else float('nan'):.0f})  "
              f"hwmon {havg:5.0f} W ({len(hs)})  {wavg * us * 1e-3:7.1f} mJ per launch  {wavg * us * 1e-6 / A[n][1] * 1e12:5.2f} pJ per FLOP", flush=True)
        time.sleep(1.0)


if __name__ == "__main__":
    main()
