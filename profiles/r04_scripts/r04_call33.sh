#!/bin/bash
# the default bench line again (roofline block with frac_at_general_step), into the round's profile directory
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/round_r04; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "rc $?"
python tools/roofline.py frac $O/bench_default.json > $O/roofline_frac.txt 2>&1
head -c 1500 $O/bench_default.json; echo
