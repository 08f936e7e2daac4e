exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_protogalaxy_gpu.py tests/test_sangria_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
pick() { python -c "
import json,sys
d=json.loads([l for l in open('$1').read().strip().splitlines() if l.startswith('{')][-1])
print('$2', 'ms/step', d['ms_per_step'], d.get('kernel_ms'), d.get('state_digest','')[:12])"; }
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20.json 2>$O/err.txt; pick $O/c20.json "k20 affine clusters"
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20b.json 2>$O/err.txt; pick $O/c20b.json "k20 affine clusters (again)"
