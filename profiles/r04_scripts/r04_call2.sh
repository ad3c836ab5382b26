#!/bin/bash
# GPU box, round 4, call 2: the new full-size tests, the build probe with the background warm-up, one default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4b; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_times or incoherent_blocks or larger_than or failed_growth or two_real or several_devices or more_streams or launch_form or wide_first" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
python3 tools/build_probe.py > $OUT/build_probe.txt 2>&1; grep -v "^W2\|^E2\|^I2\|amdgpu.ids" $OUT/build_probe.txt
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc $?"; tail -c 6000 $OUT/bench.json
