#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4q; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -4 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="final|"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d1_tm1|--workload mesh1m --depth 1 --spp 4 --option tri_min=1;d1_tm3|--workload mesh1m --depth 1 --spp 4 --option tri_min=3;d4|--workload mesh1m --depth 4 --spp 4;d4_s3|--workload mesh1m --depth 4 --spp 4 --streams 3;d4_tm3|--workload mesh1m --depth 4 --spp 4 --option tri_min=3;d4_bins5|--workload mesh1m --depth 4 --spp 4 --option ray_bins=5;d1_spp8|--workload mesh1m --depth 1 --spp 8"
bash tools/ab.sh $OUT
