#!/bin/bash
# what limits the shader clock under sustained k_accum0s?  amd-smi / rocm-smi metrics while four-chunk commits run back to back
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-thr}; mkdir -p $O; cd $R
SRS_AMD_LIB=$R/variants/fracenv.so SRS_COMMIT_FRAC=${2:-0.25,0.5,0.75} python tools/commit_loop.py 10 > $O/loop.log 2>&1 &
P=$!
while ! grep -q "loop start" $O/loop.log 2>/dev/null; do sleep 0.2; kill -0 $P 2>/dev/null || break; done
sleep 2
(amd-smi metric -g 0 2>&1 | head -150) > $O/amdsmi_metric.txt
(rocm-smi --showtemp --showpower --showvoltage --showclocks --showperflevel 2>&1 | grep -v "^=\|^$") > $O/rocmsmi.txt
(amd-smi metric -g 0 --throttle 2>&1 | head -60) > $O/amdsmi_throttle.txt
wait $P
tail -1 $O/loop.log
grep -i -E "throttl|violation|limit|hotspot|junction|temp|power|clk|volt|activity" $O/amdsmi_metric.txt | head -60
echo ----; cat $O/rocmsmi.txt | head -30; echo ----; head -40 $O/amdsmi_throttle.txt
