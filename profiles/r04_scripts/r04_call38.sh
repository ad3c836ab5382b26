#!/bin/bash
# what the driver runs at round end, on the final tree: build(), smoke(), the GPU suite, the bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4am; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.log ) 2>&1 | grep real
head -c 600 $OUT/bench_driver.json; echo
