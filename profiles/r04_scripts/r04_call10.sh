#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4k; mkdir -p $OUT
cd $R
export AB_BUILDS="k8|;mid|-DCRT_GROUP_MID;first_any|-DCRT_LANES_FIRST_ANY"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160"
bash tools/ab.sh $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or radiance_matches" > $OUT/pytest_default.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_default.log
