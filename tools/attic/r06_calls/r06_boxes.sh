# the driver's own command on a fresh box; one summary line into gpurun_out/bench_boxes_r06.txt
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/_bl.json
python - <<'P' >> gpurun_out/bench_boxes_r06.txt
import json
d=json.loads(open('gpurun_out/_bl.json').read())
t=d['train32']
print('fwd64 %.4f ms (%.2f M frames/s) frac %.4f launch %.1f us | parity %.4f | long16 %.4f | infer64 %.4f | infer_lj %.3f | train32 graph %.4f eager %.4f parity %.4f | stock fwd bf16 %.2f train bf16 %.2f' % (
 d['ms_per_step'], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_us'], d['parity_mode']['ms_per_step'], d['long16']['ms_per_step'], d['infer64']['ms_per_step'], d['infer_lj']['ms_per_step'],
 t['ms_per_step'], t['eager_ms_per_step'], t['parity_mode']['ms_per_step'], d['stock_gpu_baseline']['fwd64_bf16_autocast']['ms_per_step'], d['stock_gpu_baseline']['train32_bf16_autocast']['ms_per_step']))
P
tail -1 gpurun_out/bench_boxes_r06.txt
