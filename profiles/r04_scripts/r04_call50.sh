#!/bin/bash
# plane loads of the uniform step in one / two / three waits, on the one-pass kernels (fewer spilled SGPRs than when this was last measured)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ay; mkdir -p $OUT
cd $R
export AB_BUILDS="wait3|;wait2|-DCRT_PLANES_ONE_WAIT=2;wait1|-DCRT_PLANES_ONE_WAIT=1"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d1b|--workload mesh1m --depth 1 --spp 4"
bash tools/ab.sh $OUT
