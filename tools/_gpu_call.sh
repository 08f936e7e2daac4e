exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_protogalaxy_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
pick() { python -c "
import json,sys
d=json.loads([l for l in open('$1').read().strip().splitlines() if l.startswith('{')][-1])
print('$2', 'ms/step', d['ms_per_step'], d['config'].get('parallelism'), d.get('state_digest','')[:12])"; }
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20.json 2>$O/err.txt; pick $O/c20.json "k20 N=1"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $O/c20_2r.json 2>$O/err2.txt; pick $O/c20_2r.json "k20 2 ranks on one GPU (gloo)"
tail -3 $O/err2.txt
