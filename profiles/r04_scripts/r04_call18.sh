#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4t; mkdir -p $OUT
cd $R
export AB_BUILDS="wide_loads|;narrow_loads|-DCRT_GROUP_NARROW_LOADS"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d1|--workload mesh1m --depth 1 --spp 4"
bash tools/ab.sh $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "radiance_matches or scheduling_and_loop" > $OUT/pytest_default.log 2>&1; echo "pytest rc $?"
