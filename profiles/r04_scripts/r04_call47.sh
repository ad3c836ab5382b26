#!/bin/bash
# bounce kernels without the alternative walks: A/B (same registers: is there anything to gain?)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4av; mkdir -p $OUT
cd $R
export AB_BUILDS="lean|;general|-DCRT_LEAN_BOUNCE=0"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;d4b|--workload mesh1m --depth 4 --spp 4"
bash tools/ab.sh $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "config4 or incoherent or radiance_matches or scheduling_and_loop or million" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
