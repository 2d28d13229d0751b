#!/bin/bash
# Lab libraries of the dominant kernel, built from HEAD: the product's objects with efts_resconv.hip recompiled under one lab switch each.
#   lab/rc_stamp.so  -DRC_STAMP=1  per-workgroup start / end stamps (dispatch skew, launch span, inter-launch gap, XCD spread)
#   lab/rc_phase.so  -DRC_STAMP=2  per-wave cycle counters of the ping-pong phases (read / its barrier / MFMA / its barrier)
#   lab/rc_marks.so  -DRC_STAMP=3  tile-phase marks (tile 0 main loop / epilogue / rest)
#   lab/rc_exp<N>.so -DRC_EXP=N    ablations (1 no LDS-DMA, 4 no MFMA + fragment reads, 8 no epilogue traffic, 16 no epilogue), with EXPS="1 4 ..."
# Lab builds instantiate only the tile variant the probes launch (5 taps, planes in and out: -DRC_LAB_MIN=1, see efts_resconv.hip).
# The product library is (re)built first, so every lab library exports exactly the product's symbols: a stale lab build is what
# left four traceback-only files under profiles/ in rounds 2 and 3.
set -e
cd "$(dirname "$0")/.."
python -m efficient_tts_amd.build > /dev/null
mkdir -p lab
rm -f lab/rc_*.so            # (left-overs of earlier experiments would fail the symbol check below: every library here is rebuilt)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJS=$(ls efficient_tts_amd/build/*.o | grep -v efts_resconv.o)
build_one() {   # name, define
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DRC_LAB_MIN=1 $2 -c efficient_tts_amd/csrc/efts_resconv.hip -o /tmp/rc_$1.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o lab/rc_$1.so $OBJS /tmp/rc_$1.o
}
pids=""
build_one stamp -DRC_STAMP=1 & pids="$pids $!"
build_one phase -DRC_STAMP=2 & pids="$pids $!"
build_one marks -DRC_STAMP=3 & pids="$pids $!"
for e in ${EXPS:-}; do build_one exp$e -DRC_EXP=$e & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "LAB BUILD FAILED (pid $p)"; exit 1; }; done
# every lab library must export what the ctypes binding asks for
python - <<'PY'
import ctypes, glob, sys
sys.path.insert(0, ".")
from efficient_tts_amd import lib as L
for so in sorted(glob.glob("lab/rc_*.so")):
    h = ctypes.CDLL(so)
    missing = [n for n in L.exported_symbols() if not hasattr(h, n)]
    assert not missing, (so, missing)
    print(so, "ok")
PY
