#!/bin/bash
# round 6 A/Bs of the graphed training step (tools/gpu_probe_train_graph.py: fwd + bwd + Adam as one hipGraph, 3 x 50 replays each), one process per arm,
# arms interleaved twice.   bash tools/r06_train_ab.sh > gpurun_out/train_ab2_r06.txt
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $R
PREC=${PREC:-bf16}
arm() { echo "== $*"; timeout 150 python tools/gpu_probe_train_graph.py $PREC "$@" 2>&1 | grep -E "ms/step|Error|error" | tail -4; }
for rep in 1 2; do
arm _PACK_LATE=0
arm _PACK_LATE=1
arm _PACK_LATE=1 _RESCONV_DGRAD=3
arm _PACK_LATE=1 _WGRAD_GROUP_WGS_ME=192
arm _PACK_LATE=1 _WGRAD_GROUP_WGS_ME=288
done
PREC=bf16x3
arm _PACK_LATE=0
arm _PACK_LATE=1
