#!/bin/bash
# A/B of library builds on the 2^24 NTT: main against variants/<name>.so, interleaved.  usage: tools/ab_ntt.sh <out-tag> <rounds> <variant.so> [log_n]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-abntt}; mkdir -p $O; cd $R
for round in $(seq 1 ${2:-3}); do
  for lib in "" $R/variants/$3; do
    tag=$(basename "${lib:-main}" .so)
    SRS_AMD_LIB=$lib python tools/ntt_probe.py ${4:-24} 2>/dev/null | sed "s/^/$tag #$round /" | tee -a $O/summary.txt
  done
done
