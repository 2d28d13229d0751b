#!/bin/bash
# efts_resconv5: explicit tile schedules against the automatic one (us per launch, planes mode).  usage: rc_plans.sh BxT "plan" "plan" ...
export PCHECK=0 PREF=0 PMODES=planes PSHAPES=$1
shift
echo "== $PSHAPES automatic"; timeout 200 python tools/gpu_probe_rc.py 2>&1 | grep "us$"
for pl in "$@"; do echo "== $PSHAPES plan $pl"; PPLAN="$pl" timeout 200 python tools/gpu_probe_rc.py 2>&1 | grep "us$"; done
