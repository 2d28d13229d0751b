#!/bin/bash
# Clock and power of the chip while the dominant kernel runs back to back (evidence for DESIGN.md 4a': the launch is bound by the power
# budget).  rocm-smi is sampled beside (a) an idle chip, (b) the B = 64 layer in a loop, 8-wave kernel, (c) the one-wave-per-SIMD kernel,
# (d) the micro benchmark of the bare main loop on one CU.    bash tools/gpu_power_trace.sh > gpurun_out/power_trace.txt
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $R
sample() {  # label, seconds
  for i in $(seq 1 $2); do
    echo "[$1] $(rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -i 'sclk\|power\|mclk\|fclk' | tr -s ' ' | tr '\n' '|')"
    sleep 0.5
  done
}
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|cap" | head -4
sample idle 3
for k in 0 2; do
  RCK=$k PCHECK=0 PREF=0 PMODES=planes PSHAPES=64x800 PSPLIT=1 PLOOP=30000 timeout 120 python tools/gpu_probe_rc.py > /tmp/pw_$k.log 2>&1 &
  pid=$!
  sleep 6
  sample "resconv5 kernel $k (B=64 layer in a loop)" 8
  wait $pid
  grep "us$" /tmp/pw_$k.log
done
RCK=0 PCHECK=0 PREF=0 PMODES=planes PSHAPES=64x800 PSPLIT=2 PLOOP=15000 timeout 120 python tools/gpu_probe_rc.py > /tmp/pw_x3.log 2>&1 &
pid=$!
sleep 6
sample "resconv5 bf16x3 (B=64 layer in a loop)" 6
wait $pid
grep "us$" /tmp/pw_x3.log
