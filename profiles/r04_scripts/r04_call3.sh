#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4c; mkdir -p $OUT
cd $R
export AB_BUILDS="occ5_6|;occ6_6|-DCRT_SEG_OCC=6;occ6_7|-DCRT_SEG_OCC=6 -DCRT_SEG_OCC_FIRST=7;occ7_7|-DCRT_SEG_OCC=7 -DCRT_SEG_OCC_FIRST=7"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_share0|--workload mesh1m --depth 4 --spp 4 --option tri_share=0;d4_bins4|--workload mesh1m --depth 4 --spp 4 --option ray_bins=4;d4_bins5_s0|--workload mesh1m --depth 4 --spp 4 --option ray_bins=5 --option tri_share=0;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
