#!/bin/bash
# loops without a carried flag; float planes with one wait / three waits / bytes: parity of the default build, then A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ad; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;planes3|-DCRT_PLANES_ONE_WAIT=0;bytes|-DCRT_UNIFORM_PLANES=0;busy|-DCRT_P1_NO_BUSY=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d1_spp8|--workload mesh1m --depth 1 --spp 8;d4|--workload mesh1m --depth 4 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
