#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4n; mkdir -p $OUT
cd $R
export AB_BUILDS="first|-DCRT_LANES_FIRST;first_any|-DCRT_LANES_FIRST_ANY;first_closest|-DCRT_LANES_FIRST -DCRT_LANES_FIRST_NOANY"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d1_l1|--workload mesh1m --depth 1 --spp 4 --option lanes_per_ray=1;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d4|--workload mesh1m --depth 4 --spp 4"
bash tools/ab.sh $OUT
