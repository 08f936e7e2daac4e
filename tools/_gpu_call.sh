exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_protogalaxy_gpu.py -m gpu -x -q > $O/pytest_pg.txt 2>&1; tail -3 $O/pytest_pg.txt
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', 'ms/step', d['ms_per_step'], d.get('kernel_ms'))"; }
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_new.json 2>$O/err.txt; pick $O/c20_new.json "k20 skip-one"
SRS_PG_G_ALL_POINTS=1 timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_all.json 2>$O/err.txt; pick $O/c20_all.json "k20 all-points"
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --ro-challenge > $O/c20_ro.json 2>$O/err.txt; pick $O/c20_ro.json "k20 skip-one ro"
