#!/bin/bash
# round-6 evidence on one box: kernel stats + step trace of the headline run, SQ counters and HBM counters of the streamed commit's kernels.
# usage (GPU box): tools/r06_profile.sh <out-tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r06prof}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --no-extras --no-cpu-baseline --steps 14 --warmup 3 > $O/bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1); MC=$(find $O/kt -name "*memory_copy_trace.csv" | head -1)
python $R/tools/step_trace.py $KT $MC > $O/step_trace.txt 2>&1
rm -rf $O/kt
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -o pmc -- python $R/tools/commit_upload_only.py 20 > $O/pmc_$tag.log 2>&1
  find $O/pmc_$tag -name "*counter_collection.csv" -exec cp {} $O/pmc_$tag.csv \;
  rm -rf $O/pmc_$tag
done
python - <<PY > $O/sq_summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/pmc_*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    n = max(cnt[k].values())
    print(k, "launches", n, " ".join(f"{c}={v / cnt[k][c]:.4g}" for c, v in sorted(agg[k].items())))
PY
ls -la $O
