#!/bin/bash
# occupancy of the segment kernels on the lean loops: 5 / 6 / 7 waves per SIMD
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ah; mkdir -p $OUT
cd $R
export AB_BUILDS="w6|;w7_first|-DCRT_SEG_OCC_FIRST=7;w7_all|-DCRT_SEG_OCC_FIRST=7 -DCRT_SEG_OCC=7;w5_bounce|-DCRT_SEG_OCC=5"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d4|--workload mesh1m --depth 4 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
