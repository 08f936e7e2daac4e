exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_commit_gpu.py -m gpu -x -q -k "wide" > $O/pytest_wide.txt 2>&1; tail -3 $O/pytest_wide.txt
PROBE_TAG=wide timeout 200 python tools/msm_probe.py 24 1048576 3145728 12582912 16777216 > $O/probe_wide.txt 2>&1; grep "n=" $O/probe_wide.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_wide -o kt -- python $R/tools/msm_probe.py 24 12582912 > $O/kt_wide.log 2>&1
f=$(find $O/kt_wide -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then head -16 "$f" | cut -c1-160; fi
cd $R
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$2', 'ms/step', d['ms_per_step'], 'dev-res', d.get('device_resident_ms_per_step'), 'accum0 avg', r.get('avg_launch_ms'), 'madd G/s', (r.get('alu') or {}).get('achieved'))"; }
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_wide.json 2>$O/c20_wide.err; pick $O/c20_wide.json "k20 wide"
timeout 300 python bench.py --config sangria --no-cpu-baseline --steps 20 --warmup 3 > $O/s17_wide.json 2>$O/s17_wide.err; pick $O/s17_wide.json "k17 wide"
