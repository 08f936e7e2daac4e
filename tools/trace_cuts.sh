#!/bin/bash
# step trace (tools/step_trace.py) of the headline step under a given chunk schedule (variants/fracenv.so, SRS_COMMIT_FRAC)
# usage (GPU box): tools/trace_cuts.sh <out-tag> <file with lines "name f1,f2,...">
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-trcuts}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LIST=$R/$2
while read -r name frac; do
  [ -z "$name" ] && continue
  SRS_COMMIT_FRAC=$frac SRS_AMD_LIB=$R/variants/fracenv.so rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/$name.json 2> $O/$name.err
  KT=$(find $O/kt_$name -name "*kernel_trace.csv" | head -1); MC=$(find $O/kt_$name -name "*memory_copy_trace.csv" | head -1)
  python $R/tools/step_trace.py $KT $MC > $O/step_trace_$name.txt 2>&1
  rm -rf $O/kt_$name
done < $LIST
