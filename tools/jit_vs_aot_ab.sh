# A/B of the ahead-of-time and the run-time compiled (hiprtc) cross-term kernels on the same box: same source text, same
# ISA (SRS_JIT_DUMP keeps the hiprtc code object so that can be checked with llvm-objdump).  Prints fold-steps/s,
# ms/step and the event-timed cross-term launch for each.
mkdir -p gpurun_out/jitdump
for i in 1 2; do
  echo "== AOT"; timeout -k 5 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cross_terms_ms_per_launch'])"
  echo "== JIT"; SRS_JIT_DUMP=gpurun_out/jitdump SRS_NO_SPEC=1 SRS_JIT_ALWAYS=1 timeout -k 5 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cross_terms_ms_per_launch'])"
done
