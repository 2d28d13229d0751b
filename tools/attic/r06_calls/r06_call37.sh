mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_frontend.py tests/test_cli_data.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4 ) > gpurun_out/t_fe.txt; cat gpurun_out/t_fe.txt
python - <<'P'
import torch, time
from efficient_tts_amd.frontend import LogMelFrontend
dev=torch.device('cuda:0'); B,T=64,800
a16=(torch.rand(B,T*256)*2-1).mul(0.3*32768).to(torch.int16).to(dev); lengths=torch.full((B,),T*256)
fe=LogMelFrontend(dev)
af=a16.float()/32768.0
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return round(s.elapsed_time(e)/n*1e3,1)
print('int16 batch on the device -> log-mel: pcm16 launch', t(lambda: fe(a16,lengths)), 'us per call; convert first (int16 -> fp32 / 32768, then the fp32 launch)', t(lambda: fe(a16.float()/32768.0,lengths)), 'us; fp32 input', t(lambda: fe(af,lengths)), 'us')
print('equal', torch.equal(fe(a16,lengths)[0], fe(af,lengths)[0]))
P
