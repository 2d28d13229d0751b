#!/bin/bash
# A/B of lab builds of efts_resconv5 (us per launch, planes mode): rc_ab.sh BxT[,BxT] name name ...  (name = lab/rc_<name>.so; "cur" = the product library)
export PCHECK=0 PREF=0 PMODES=planes PSHAPES=$1
shift
for rep in 1 2; do
for n in "$@"; do
  if [ $n = cur ]; then unset EFTS_LIB; else export EFTS_LIB=$GRAFT_REPO_ROOT/lab/rc_$n.so; fi
  echo "== $n"; timeout 200 python tools/gpu_probe_rc.py 2>&1 | grep "us$"
done; done
