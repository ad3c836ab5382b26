#!/bin/bash
# one read-modify-write of the sum for the four samples of a pixel: parity, A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ar; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -4 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;four_rmw|-DCRT_SUM_ONCE=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d1_spp8|--workload mesh1m --depth 1 --spp 8;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d1b|--workload mesh1m --depth 1 --spp 4"
bash tools/ab.sh $OUT
