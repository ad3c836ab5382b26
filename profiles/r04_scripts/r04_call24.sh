#!/bin/bash
# uniform node steps through the scalar cache: parity of the default build, then A/B against the builds without them
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4z; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="uni_both|;uni_none|-DCRT_UNIFORM_CLOSEST=0 -DCRT_UNIFORM_ANY=0;uni_closest|-DCRT_UNIFORM_ANY=0;uni_any|-DCRT_UNIFORM_CLOSEST=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d4|--workload mesh1m --depth 4 --spp 4;cornell|--workload cornell --depth 1 --spp 1 --steps 200;d1_spp8|--workload mesh1m --depth 1 --spp 8"
bash tools/ab.sh $OUT
