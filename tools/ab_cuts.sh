#!/bin/bash
# A/B of chunk schedules of the streamed commit on the headline step: variants/fracenv.so (capi.hip patched to read the cumulative cut
# fractions from SRS_COMMIT_FRAC -- an A/B build, not the product) against itself, interleaved.
# usage: tools/ab_cuts.sh <out-tag> <rounds> <file with lines "name f1,f2,...">
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-abcuts}; mkdir -p $O; cd $R
for round in $(seq 1 ${2:-2}); do
  while read -r name frac; do
    [ -z "$name" ] && continue
    SRS_COMMIT_FRAC=$frac SRS_AMD_LIB=$R/variants/fracenv.so python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.$round.json 2>$O/$name.$round.err
    python -c "
import json
d=json.loads(open('$O/$name.$round.json').read().strip().splitlines()[-1])
print('$name #$round ms/step', d['ms_per_step'], 'digest', d.get('state_digest','')[:12])" | tee -a $O/summary.txt
  done < $3
done
