#!/bin/bash
# uniform node steps in the plain per-lane loop (few-node scenes): A/B, then parity of the default build
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4aa; mkdir -p $OUT
cd $R
export AB_BUILDS="uni_all|;uni_noplain|-DCRT_UNIFORM_PLAIN=0;uni_none|-DCRT_UNIFORM_CLOSEST=0 -DCRT_UNIFORM_ANY=0 -DCRT_UNIFORM_PLAIN=0"
export AB_RUNS="cornell|--workload cornell --depth 1 --spp 1 --steps 200;cornell4|--workload cornell --depth 1 --spp 4 --steps 100;d1|--workload mesh1m --depth 1 --spp 4;d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;cornell_d4|--workload cornell --depth 4 --spp 1 --steps 100"
bash tools/ab.sh $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
