export PCHECK=0 PREF=0 PMODES=planes PSPLIT=2
for v in C e1 e4 e5 e16; do
  if [ $v = C ]; then unset EFTS_LIB; else export EFTS_LIB=/root/repo/lab/librc_$v.so; fi
  echo "== variant $v : split 2, B=32 sched 4"; EFTS_RC_SCHED=4 PSHAPES=32x800 python tools/gpu_probe_rc.py 2>&1 | grep " us"
  echo "== variant $v : split 2, B=64 default"; PSHAPES=64x800 python tools/gpu_probe_rc.py 2>&1 | grep " us"
done
