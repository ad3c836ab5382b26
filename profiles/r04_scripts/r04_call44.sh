#!/bin/bash
# the one-pass build of the batched first-segment kernel: parity, A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4as; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -4 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;general|-DCRT_ONE_PASS_KERNEL=0"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;d2|--workload mesh1m --depth 2 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d1b|--workload mesh1m --depth 1 --spp 4"
bash tools/ab.sh $OUT
