#!/bin/bash
# the GPU suite and a soak against a library built with every experimental variant (make EXPERIMENTS=1)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4o; mkdir -p $OUT
cd $R
make -C caitlynrenderer_amd/csrc -s clean; make -C caitlynrenderer_amd/csrc -s -j8 EXPERIMENTS=1 > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
echo "experiments build done"
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py::test_bench_line_contract > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -6 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/soak.py 3000 77 > $OUT/soak.log 2>&1; echo "soak rc $?"; tail -3 $OUT/soak.log
