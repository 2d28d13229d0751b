"""Build libefts_hip.so (gfx950) in-tree with hipcc.  No torch / pybind dependency: the
library is a plain C-ABI shared object loaded with ctypes (efficient_tts_amd/lib.py)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["efts_common.hip", "efts_gemm.hip", "efts_gemm_narrow.hip", "efts_conv5.hip", "efts_smallm.hip", "efts_resconv.hip", "efts_resconv_bwd.hip", "efts_prenet.hip", "efts_ops.hip", "efts_align.hip", "efts_train.hip", "efts_frontend.hip", "efts_wgrad.hip", "efts_vocoder.hip", "efts_act.hip"]
LIB = os.path.join(HERE, "libefts_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "efts_abi.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    cc = hipcc()
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [cc, *FLAGS, *os.environ.get("EFTS_CFLAGS", "").split(), "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    out = os.environ.get("EFTS_LIB_OUT", LIB)   # experiments: build a variant next to the product library
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
