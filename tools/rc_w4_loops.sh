cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRC_W4=2 $EXTRA -S --cuda-device-only -o rc4.s /root/repo/efficient_tts_amd/csrc/efts_resconv.hip 2>&1 | grep -i "error" | head -3
python3 - <<'PY'
from collections import Counter
txt=open('/tmp/rc4.s').read()
i=txt.index('_ZN4efts17resconv5w4_kernelILi1EEEvNS_6RcArgsE:')
j=txt.index('.Lfunc_end', i)
lines=txt[i:j].split('\n')
bars=[k for k,l in enumerate(lines) if 's_barrier' in l]
rows=[]
for a,b in zip(bars,bars[1:]):
    seg=lines[a:b]
    n=sum('v_mfma' in l for l in seg)
    if n>=40:
        rows.append((n, sum(('scratch_' in l) for l in seg), sum(('v_accvgpr' in l) for l in seg), sum(('ds_read' in l) for l in seg), sum('vmcnt(0)' in l for l in seg), sum(('v_readlane' in l or 'v_writelane' in l) for l in seg), b-a))
print("(mfma, scratch, accvgpr, ds_read, vmcnt0, lane-spill-ops, instrs)")
for r,c in sorted(Counter(rows).items()): print(r,c)
import re
m=re.search(r'resconv5w4_kernelILi1.*?\.vgpr_spill_count:\s*(\d+)', txt, re.S)
PY
