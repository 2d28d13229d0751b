#!/bin/bash
# What ONE tile of efts_resconv5 costs as a function of its height h (half units of 32 window rows; a tile yields 32 h - 4 rows): every
# workgroup of the chip (128 groups x 2 column halves) runs exactly one tile of height h -- the row space is sized for that -- so the
# launch time IS the tile's time (prologue + 40 (chunk, tap) steps + epilogue).  The intercept of the line through these points is the fixed
# cost of any additional tile (the 1.3 MB weight stream of a column half + first-operand latency + epilogue ramp): the granularity a
# dynamic tail -- tiles handed out by an atomic counter at the end of a launch -- would have to work with (DESIGN.md 4a''').
#    bash tools/r06_tile_cost.sh > gpurun_out/rc_tile_cost_r06.txt
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $R
for sp in 1 2; do
for h in 2 3 4 5 6 7 8; do
  T=$(( (32 * h - 4) * 128 / 64 - 2 ))
  echo "h=$h rows_per_tile=$((32 * h - 4)) shape=64x$T split=$sp"
  PPLAN=$h PSHAPES=64x$T PSPLIT=$sp PMODES=planes PCHECK=0 PREF=0 PLOOP=300 timeout 120 python tools/gpu_probe_rc.py 2>&1 | grep "us$"
done
done
# the product schedule for comparison: 64 x 800 rows, automatic plan (two tiles per workgroup: 7 + 6 / 6 + 7 half units)
PSHAPES=64x800 PSPLIT=1 PMODES=planes PCHECK=0 PREF=0 PLOOP=300 timeout 120 python tools/gpu_probe_rc.py 2>&1 | grep "us$"
