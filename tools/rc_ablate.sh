# A/B of efts_resconv5 build variants (lab/librc_<name>.so): time per launch at B=64 / B=32
export PCHECK=${PCHECK:-0} PREF=0 PMODES=planes
for sp in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset EFTS_LIB; else export EFTS_LIB=/root/repo/lab/librc_$v.so; fi
  echo "== variant $v split $sp"; PSPLIT=$sp PSHAPES=${PSHAPES:-32x800,64x800} python tools/gpu_probe_rc.py 2>&1 | grep -E " us|EQUAL|MISMATCH|equal"
done; done
