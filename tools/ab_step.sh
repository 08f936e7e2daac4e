#!/bin/bash
# A/B of library builds on the headline step: main (sirius_amd/csrc/libsirius_amd.so) against variants/*.so (tools/build_variant.py),
# interleaved, $2 rounds (default 2).  usage: tools/ab_step.sh <out-tag> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd $R
pick() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$2', 'ms/step', d['ms_per_step'], 'accum0 avg ms', r.get('avg_launch_ms'), 'madd G/s', (r.get('alu') or {}).get('achieved'), 'digest', d.get('state_digest','')[:12])" | tee -a $O/summary.txt; }
for round in $(seq 1 ${2:-2}); do
  for lib in "" $R/variants/*.so; do
    tag=$(basename "${lib:-main}" .so)
    SRS_AMD_LIB=$lib python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/c20_${tag}_$round.json 2>$O/c20_${tag}_$round.err; pick $O/c20_${tag}_$round.json "k20 $tag #$round"
  done
done
