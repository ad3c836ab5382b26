#!/bin/bash
# uniform node steps with the planes as floats from the scalar cache: parity, then A/B against the byte form
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ac; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="planes|;bytes|-DCRT_UNIFORM_PLANES=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d1_spp8|--workload mesh1m --depth 1 --spp 8;d2|--workload mesh1m --depth 2 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d1_tm1|--workload mesh1m --depth 1 --spp 4 --option tri_min=1;d1_tm3|--workload mesh1m --depth 1 --spp 4 --option tri_min=3"
bash tools/ab.sh $OUT
