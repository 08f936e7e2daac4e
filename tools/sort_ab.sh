SRS_MSM_SORT=2 timeout -k 5 300 python -m pytest tests/test_commit_gpu.py -m gpu -x -q 2>&1 | tail -2
for mode in 1 2; do
  echo "== SRS_MSM_SORT=$mode"
  SRS_MSM_SORT=$mode timeout -k 5 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k17', d['value'], d['ms_per_step'])"
  SRS_MSM_SORT=$mode timeout -k 5 300 python tools/cyclefold_probe.py --k 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k20 commit ms', d['witness_commit_ms'])"
  SRS_MSM_SORT=$mode timeout -k 5 300 python tools/microbench.py 2>/dev/null | tail -2 | cut -c1-95
done
