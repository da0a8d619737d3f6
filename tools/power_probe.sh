#!/bin/bash
# Samples GPU clock/power with rocm-smi while a long bench runs (diagnostic only).
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --variant ${1:-0} > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|busy" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/power_bench.json')); print(d['kernel_ms'], d['roofline']['frac'])"
