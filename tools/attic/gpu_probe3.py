"""one profiled launch of the k5 conv (EFTS_GEMM_PROF=1 prints per-phase cycles)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PITERS", "2")
import runpy
src = open(os.path.join(os.path.dirname(__file__), "gpu_probe2.py")).read().replace("iters=20", "iters=int(os.environ['PITERS'])")
exec(compile(src, "gpu_probe2", "exec"))
