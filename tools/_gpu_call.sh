exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', 'ms/step', d['ms_per_step'])"; }
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_a.json 2>$O/err.txt; pick $O/c20_a.json "k20 default"
SRS_MSM_SORT=2 timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_b.json 2>$O/err.txt; pick $O/c20_b.json "k20 two-pass sort"
PROBE_TAG=sort1 SRS_MSM_WIDE=0 SRS_MSM_SORT=1 timeout 200 python tools/msm_probe.py 24 1048576 3145728 5242880 12582912 > $O/p1.txt 2>&1; grep "n=" $O/p1.txt
PROBE_TAG=sort2 SRS_MSM_WIDE=0 SRS_MSM_SORT=2 timeout 200 python tools/msm_probe.py 24 1048576 3145728 5242880 12582912 > $O/p2.txt 2>&1; grep "n=" $O/p2.txt
