#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-clk}; mkdir -p $O; cd $R
export SRS_AMD_LIB=$R/variants/clk.so
python tools/clk_commit.py > $O/main.txt 2>&1
SRS_COMMIT_FRAC=0.0907,0.1877,0.2904,0.3981,0.5106,0.6273,0.748,0.8724 python tools/clk_commit.py > $O/n9.txt 2>&1
SRS_COMMIT_FRAC=0.25,0.5,0.75 python tools/clk_commit.py > $O/q4.txt 2>&1
tail -n 40 $O/main.txt $O/n9.txt $O/q4.txt
