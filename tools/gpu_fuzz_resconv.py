"""One-off robustness sweep: efts_resconv5 with the automatic half-unit schedule against efts_gemm, bit for bit, on random
row-space shapes (both operand formats, both stream formats); prints the shapes that fail, if any."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_resconv_gpu import Case
random.seed(int(os.environ.get("SEED", 1)))
bad = []
n = int(os.environ.get("N", 60))
for it in range(n):
    B = random.choice([1, 2, 3, 5, 8, 13, 16, 32, 64])
    T = random.choice([1, 2, 5, 29, 30, 31, 59, 60, 61, 63, 64, 65, 100, 124, 125, 187, 188, 189, 250, 251, 252, 253, 400, 799, 800, 801, 1200])
    if B * T > 64 * 1200:
        continue
    split = random.choice([1, 2])
    mode = random.choice(["f32", "planes"])
    try:
        Case(B, T, split, seed=it).check(mode)
    except AssertionError as e:
        bad.append((B, T, split, mode, str(e)[:80]))
        print("FAIL", bad[-1], flush=True)
print(f"{n} shapes, {len(bad)} failures", bad)
