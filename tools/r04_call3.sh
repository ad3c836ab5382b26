#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4c; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="occ5_6|;occ6_6|-DCRT_SEG_OCC=6;occ6_7|-DCRT_SEG_OCC=6 -DCRT_SEG_OCC_FIRST=7;occ7_7|-DCRT_SEG_OCC=7 -DCRT_SEG_OCC_FIRST=7"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_share0|--workload mesh1m --depth 4 --spp 4 --option tri_share=0;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
