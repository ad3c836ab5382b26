#!/bin/bash
# exact short square root as well: parity, A/B against the build without the short forms
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4al; mkdir -p $OUT
cd $R
timeout -k 5 120 ./tools/ubench/sqrt_exhaustive | tee $OUT/sqrt_exhaustive.txt
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;div|-DCRT_FAST_RCP=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d4|--workload mesh1m --depth 4 --spp 4;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;cornell|--workload cornell --depth 1 --spp 1 --steps 200;cornell_d4|--workload cornell --depth 4 --spp 1 --steps 100"
bash tools/ab.sh $OUT
