#!/bin/bash
# Shader clock and power while the headline kernel runs: bench.py in the background, rocm-smi sampled once a second.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python $R/bench.py --steps 150 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $BP
tail -c 300 $R/gpurun_out/clk_bench.json; echo
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
