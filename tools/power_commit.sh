#!/bin/bash
# package power (rocm-smi, ~5 Hz) while streamed commits run back to back under three chunk schedules (variants/fracenv.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-powc}; mkdir -p $O; cd $R
run() {  # name frac
  SRS_AMD_LIB=$R/variants/fracenv.so SRS_COMMIT_FRAC=$2 python tools/commit_loop.py 8 > $O/$1.log 2>&1 &
  P=$!
  while ! grep -q "loop start" $O/$1.log 2>/dev/null; do sleep 0.2; kill -0 $P 2>/dev/null || break; done
  sleep 1
  : > $O/$1.pow
  for i in $(seq 1 25); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" >> $O/$1.pow; sleep 0.1; done
  wait $P
  echo "== $1: $(tail -1 $O/$1.log)"
  grep "Package Power" $O/$1.pow | awk '{s+=$NF; n++; if($NF>m)m=$NF} END{printf "   power W: mean %.0f max %.0f (%d samples)\n", s/n, m, n}'
  grep "sclk" $O/$1.pow | sed 's/.*(\([0-9]*\)Mhz).*/\1/' | awk '{s+=$1; n++} END{printf "   sclk MHz (rocm-smi samples): mean %.0f\n", s/n}'
}
run main 0.02,0.058,0.115,0.19,0.285,0.40,0.535,0.69,0.86
run n9 0.0907,0.1877,0.2904,0.3981,0.5106,0.6273,0.748,0.8724
run q4 0.25,0.5,0.75
