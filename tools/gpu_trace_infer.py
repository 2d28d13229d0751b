"""B=1 free-running inference of one LJSpeech utterance, for a rocprofv3 kernel trace: python tools/gpu_trace_infer.py [n_calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from efficient_tts_amd import EfficientTTSCNN
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inference_lj.npz"))
torch.manual_seed(0)
m = EfficientTTSCNN(num_symbols=76, dropout_rate=0.0, use_masking=True, sigma=0.01, precision=os.environ.get("PREC", "bf16"))
with torch.no_grad():
    m.duration_predictor.linear.bias.fill_(1.9)
m = m.to(dev).eval()
m.remove_weight_norm()
x = torch.from_numpy(g["ids3"])[None].to(dev)
for _ in range(5):
    mel, _ = m.inference(x)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(n):
    m.inference(x)
torch.cuda.synchronize()
print(f"T1={x.shape[1]} T2={mel.shape[1]}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call")
