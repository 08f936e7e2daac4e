#!/bin/bash
# A/B of tunables on the headline step (SRS_TEST_TUNING is the Python mirror's test hook), interleaved with variants/*.so as baselines.
# usage: tools/ab_tuning.sh <out-tag> <rounds> "<name> <lib or -> <k=v,...or ->" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-abtun}; mkdir -p $O; cd $R; rounds=$2; shift 2
for round in $(seq 1 $rounds); do
  for spec in "$@"; do
    set -- $spec; name=$1; lib=$2; tun=$3
    [ "$lib" = "-" ] && lib=""; [ "$tun" = "-" ] && tun=""
    SRS_TEST_TUNING=$tun SRS_AMD_LIB=${lib:+$R/$lib} python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.$round.json 2>$O/$name.$round.err
    python -c "
import json
d=json.loads(open('$O/$name.$round.json').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$name #$round ms/step', d['ms_per_step'], 'accum0 avg ms', r.get('avg_launch_ms'), 'madd G/s', (r.get('alu') or {}).get('achieved'), 'digest', d.get('state_digest','')[:12])" | tee -a $O/summary.txt
  done
done
