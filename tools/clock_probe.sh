#!/bin/bash
# sample the shader clock and the socket power while bench.py runs (is the step power-limited?)  -> gpurun_out/clock_probe.txt
out=gpurun_out/clock_probe.txt; : > $out
python bench.py --no-cpu-baseline --no-inference --no-alt-mode --steps 2500 --warmup 5 --no-kernel-events > gpurun_out/clock_probe_bench.json 2>/dev/null &
pid=$!
sleep 24
for i in $(seq 1 20); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' ' >> $out; echo >> $out
  sleep 0.4
done
wait $pid
python -c "
import json;d=json.load(open('gpurun_out/clock_probe_bench.json'));print('bench', d['value'], d['ms_per_step'])" >> $out
cat $out
