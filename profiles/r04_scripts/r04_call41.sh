#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ap; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
