"""efts_logmel_fft alone in a loop (64 x 800 frames): us per launch -- the launch itself (ctypes call, events around 50 launches) and the
LogMelFrontend call around it (host-side length checks + two small H2D copies + the output allocation).
EFTS_LIB selects a lab build (FF_ABL ablations: tools/r06_fft_abl.sh)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_tts_amd import lib as L, ops as O
from efficient_tts_amd.frontend import LogMelFrontend
dev = torch.device("cuda:0")
B, T2 = 64, 800
audio = (torch.rand(B, T2 * 256) * 2 - 1).mul_(0.3).to(dev)
lengths = torch.full((B,), T2 * 256)
fe = LogMelFrontend(dev)
li = lengths.to(torch.int32).to(dev)
out = torch.empty(B, T2, 80, device=dev)
lib = L.load()


def launch():
    L.check(lib.efts_logmel_fft(audio.data_ptr(), audio.shape[1], li.data_ptr(), fe.window.data_ptr(), fe.basis.data_ptr(), fe.ranges.data_ptr(),
                                out.data_ptr(), B, T2, 1024, 256, 80, torch.cuda.current_stream().cuda_stream), "efts_logmel_fft")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n * 1e3, 1)


print(os.environ.get("EFTS_LIB", "product").split("/")[-1], "launch", timeit(launch), "us;  LogMelFrontend call", timeit(lambda: fe(audio, lengths)), "us")
