#!/bin/bash
# Polls rocm-smi (clocks, power) while a sustained spectrum / main bench loop runs: is the kernel power-throttled?
LEG=${1:-spectrum}
timeout 120 python bench.py --legs $LEG --no-cpu --steps 6000 --warmup 3 > gpurun_out/watch_$LEG.json 2>/dev/null &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|socclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1.5
done
wait $PID
python -c "
import json; d=json.load(open('gpurun_out/watch_$LEG.json')); d=d.get('$LEG', d); print('$LEG', d.get('ms_per_step'))"
