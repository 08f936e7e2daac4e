R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_commit_gpu.py -m gpu -x -q > $O/pytest_commit.txt 2>&1; tail -3 $O/pytest_commit.txt
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$2', 'ms/step', d['ms_per_step'], 'dev-res', d.get('device_resident_ms_per_step'), 'host', d.get('host_path_ms_per_step'), 'accum0 avg', r.get('avg_launch_ms'), 'madd G/s', (r.get('alu') or {}).get('achieved'))"; }
for rep in 1 2; do
for v in pipe nopipe; do
  if [ $v = nopipe ]; then export SRS_MSM_NO_PIPE=1; else unset SRS_MSM_NO_PIPE; fi
  timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_$v.json 2>$O/c20_$v.err; pick $O/c20_$v.json "k20 $v"
  timeout 600 python bench.py --config sangria --no-cpu-baseline --steps 20 --warmup 3 > $O/s17_$v.json 2>$O/s17_$v.err; pick $O/s17_$v.json "k17 $v"
done
done
unset SRS_MSM_NO_PIPE
