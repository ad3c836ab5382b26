#!/bin/bash
# uniform node steps in the single-sample kernel, on the lean loops
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ai; mkdir -p $OUT
cd $R
export AB_BUILDS="dflt|;single|-DCRT_UNIFORM_SINGLE=1;single_plain|-DCRT_UNIFORM_SINGLE=1 -DCRT_UNIFORM_PLAIN=1"
export AB_RUNS="d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60;cornell|--workload cornell --depth 1 --spp 1 --steps 200;k4_spp1|--workload mesh1m --depth 1 --spp 1 --resolution 3840x2160 --steps 30;d1|--workload mesh1m --depth 1 --spp 4;hbm_spp1|--workload mesh520 --device-built sah --depth 1 --spp 1 --steps 20"
bash tools/ab.sh $OUT
