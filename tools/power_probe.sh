#!/bin/bash
# power / clock readings of the GPU while the headline step runs back to back (is the chip power-managed under the commit's kernels?)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-power}; mkdir -p $O; cd $R
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > $O/idle.txt 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 600 --warmup 5 > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks -t > $O/busy_$i.txt 2>&1
  sleep 0.3
done
wait $BP
tail -c 600 $O/bench.json
grep -h -i "power\|sclk\|mclk\|fclk\|Temperature" $O/idle.txt | head -20
echo ---- busy
grep -h -i "power\|sclk\|Temperature (Sensor junction\|hotspot" $O/busy_*.txt | sort | uniq -c | sort -rn | head -40
