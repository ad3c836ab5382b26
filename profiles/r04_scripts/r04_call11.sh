#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4l; mkdir -p $OUT
cd $R
export AB_BUILDS="both|;closest_only|-DCRT_LANES_ANY=0;any_only|-DCRT_LANES_CLOSEST=0"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d4_l1|--workload mesh1m --depth 4 --spp 4 --option lanes_per_ray=1;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10"
bash tools/ab.sh $OUT
