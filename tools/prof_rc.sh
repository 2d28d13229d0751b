#!/bin/bash
# rocprofv3 PMC passes on the efts_resconv5 launch alone (tools/gpu_probe_rc.py, B=64 x 800 frames, planes mode): where do
# the wave cycles go?  One counter group per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_rc
mkdir -p $OUT
CMD="python $R/tools/gpu_probe_rc.py"
export PCHECK=0 PREF=0 PMODES=planes PSPLIT=${PSPLIT:-1} PSHAPES=64x800
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o rc -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import glob, sqlite3
for d in sorted(glob.glob("$OUT/p*/")):
    f = glob.glob(d + "**/*.db", recursive=True)
    if not f:
        print(d, "no db"); continue
    con = sqlite3.connect(f[0])
    for r in con.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%resconv5%' group by counter_name"):
        print(f"{r[0]:32s} {r[1]:16.1f}  n={r[2]}")
PY
find $OUT -name "*.db" -delete
