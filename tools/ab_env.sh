#!/bin/bash
# A/B of runtime environment settings on the headline step, interleaved.  usage: tools/ab_env.sh <out-tag> <rounds> "name VAR=val [VAR=val ...]" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-abenv}; mkdir -p $O; cd $R; rounds=$2; shift 2
for round in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs=""
    env $envs python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.$round.json 2>$O/$name.$round.err
    python -c "
import json
d=json.loads(open('$O/$name.$round.json').read().strip().splitlines()[-1])
print('$name #$round ms/step', d['ms_per_step'], 'digest', d.get('state_digest','')[:12])" | tee -a $O/summary.txt
  done
done
