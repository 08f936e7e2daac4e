#!/bin/bash
# A/B of library builds (variants/*.so, built from patched source trees) and of the MSM level-0 part length on the two bench configs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd $R
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$2', 'ms/step', d['ms_per_step'], 'dev-res', d.get('device_resident_ms_per_step'), 'cross', d.get('cross_terms_ms_per_launch'), 'kernels', d.get('kernel_ms'), 'accum0 avg', r.get('avg_launch_ms'), 'madd G/s', (r.get('alu') or {}).get('achieved'))"; }
for lib in "" $R/variants/*.so; do
  tag=$(basename "${lib:-main}" .so)
  SRS_AMD_LIB=$lib python bench.py --config sangria --no-cpu-baseline --steps 20 --warmup 3 > $O/s17_$tag.json 2>/dev/null; pick $O/s17_$tag.json "k17 $tag"
  SRS_AMD_LIB=$lib python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/c20_$tag.json 2>/dev/null; pick $O/c20_$tag.json "k20 $tag"
done
for l0 in $AB_L0; do
  SRS_TEST_TUNING=msm_l0=$l0 python bench.py --config sangria --no-cpu-baseline --steps 20 --warmup 3 > $O/s17_l0_$l0.json 2>/dev/null; pick $O/s17_l0_$l0.json "k17 L0=2^$l0"
  SRS_TEST_TUNING=msm_l0=$l0 python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/c20_l0_$l0.json 2>/dev/null; pick $O/c20_l0_$l0.json "k20 L0=2^$l0"
done
