"""mJ per launch from tools/gpu_power_trace.sh's output: mean socket power of the samples taken beside a sustained loop of the layer
x the loop's microseconds per launch (the budget the layer is actually bound by, DESIGN.md 4a')"""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
idle = [float(m.group(1)) for ln in txt if ln.startswith("[idle]") for m in [re.search(r"Power \(W\): ([0-9.]+)", ln)] if m]
pidle = sum(idle) / len(idle) if idle else float("nan")
print(f"\n## energy per launch (tools/power_summary.py over {sys.argv[1].split('/')[-1]}; idle socket power {pidle:.0f} W)")
label, watts = None, []
for ln in txt:
    m = re.match(r"\[(.*?)\].*Power \(W\): ([0-9.]+)", ln)
    if m and not m.group(1).startswith("idle"):
        if m.group(1) != label:
            label, watts = m.group(1), []
        watts.append(float(m.group(2)))
        continue
    m = re.match(r"\s*B=(\d+) T=(\d+) split=(\d) planes: ([0-9. ]+) us", ln)
    if m and label and watts:
        us = sorted(float(x) for x in m.group(4).split())[len(m.group(4).split()) // 2]
        w = sum(watts) / len(watts)
        flop = 2.0 * int(m.group(1)) * int(m.group(2)) * 512 * 512 * 5
        print(f"{label}: {w:.0f} W x {us:.1f} us = {w * us * 1e-3:.0f} mJ per launch ({pidle * us * 1e-3:.0f} of them the idle socket), "
              f"{w * us * 1e-6 / flop * 1e12:.2f} pJ per FLOP, {flop / us * 1e-6:.0f} TFLOP/s")
        label, watts = None, []
