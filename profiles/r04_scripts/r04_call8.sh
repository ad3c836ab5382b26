#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4i; mkdir -p $OUT
cd $R
export AB_BUILDS="k8|;k8_occ5|-DCRT_SEG_OCC=5;k8_first|-DCRT_LANES_FIRST"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_tm1|--workload mesh1m --depth 4 --spp 4 --option tri_min=1;d4_tm3|--workload mesh1m --depth 4 --spp 4 --option tri_min=3;d4_s3|--workload mesh1m --depth 4 --spp 4 --streams 3;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
