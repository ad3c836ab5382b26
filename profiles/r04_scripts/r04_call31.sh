#!/bin/bash
# group phase: lanes take bit positions below the highest pending triangle: parity, then A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ag; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;rank|-DCRT_GROUP_TRI_WINDOW=0"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d1|--workload mesh1m --depth 1 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160"
bash tools/ab.sh $OUT
