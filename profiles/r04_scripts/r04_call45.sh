#!/bin/bash
# one-pass builds of the material / texture kernels (6 waves): parity, A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4at; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -4 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;mat5|-DCRT_ONE_MAT_OCC6=0;general|-DCRT_ONE_PASS_KERNEL=0"
export AB_RUNS="d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;d1_disney|--workload mesh1m --depth 1 --spp 4 --materials disney;d2_disney|--workload mesh1m --depth 2 --spp 4 --materials disney;d1|--workload mesh1m --depth 1 --spp 4;cornell4|--workload cornell --depth 1 --spp 4 --steps 100"
bash tools/ab.sh $OUT
