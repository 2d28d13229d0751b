mkdir -p gpurun_out
WL=vocoder STEPS=3 TSTEPS=10 TWARM=3 TIMELINE=20 bash tools/prof_conv.sh r06_vocoder_pmc > /dev/null 2>&1
python - <<'P'
import re,collections
txt=open('gpurun_out/prof_r06_vocoder_pmc/summary_r06_vocoder_pmc.txt').read().splitlines()
d=collections.defaultdict(dict)
for l in txt:
    m=re.match(r"(.{50,75}?)\s+(SQ_\w+|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|TCC_\w+)\s+([0-9.]+)\s+n=(\d+) grid=(\d+)",l)
    if m: d[(m.group(1).strip()[:44],m.group(5))][m.group(2)]=float(m.group(3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get('SQ_BUSY_CYCLES',0)):
    if 'SQ_LDS_IDX_ACTIVE' in v:
        print(f"{k[0]:46s} grid={k[1]:>8s} busy={v.get('SQ_BUSY_CYCLES',0)/1e6:6.2f}M lds={v['SQ_LDS_IDX_ACTIVE']/1e6:7.2f}M confl={v['SQ_LDS_BANK_CONFLICT']/1e6:7.2f}M mfma={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6:7.2f}M valu={v.get('SQ_INSTS_VALU',0)/1e6:6.2f}M wait={v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f}")
P
