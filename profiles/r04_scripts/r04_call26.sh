#!/bin/bash
# uniform node steps in the batched and counting kernels only: parity, then A/B against the build without them
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ab; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;uni_none|-DCRT_UNIFORM_CLOSEST=0 -DCRT_UNIFORM_ANY=0"
export AB_RUNS="cornell|--workload cornell --depth 1 --spp 1 --steps 200;cornell4|--workload cornell --depth 1 --spp 4 --steps 100;d1|--workload mesh1m --depth 1 --spp 4;d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60;d2|--workload mesh1m --depth 2 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10"
bash tools/ab.sh $OUT
