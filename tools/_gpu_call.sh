exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02u; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 300 $O/bench_full.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt17 -o kt -- python $R/bench.py --config sangria --no-cpu-baseline --steps 20 --warmup 3 > $O/kt17.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pf -- python $R/bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pw -- python $R/bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
cd $R
ff=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); fw=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
if [ -n "$ff" ] && [ -n "$fw" ]; then python tools/pmc_kernels.py "$ff" "$fw" 63668224 > $O/pmc_accum0.json 2>$O/pmc_err.txt; rm -f "$ff" "$fw"; fi
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
ls $O
