#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4j; mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -6 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 500 python tools/soak.py 6000 41 > $OUT/soak.log 2>&1; echo "soak rc $?"; tail -4 $OUT/soak.log
