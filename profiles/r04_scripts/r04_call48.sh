#!/bin/bash
# lean single-sample first-segment kernels (crt_render_frame on default options): parity, A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4aw; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -4 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;general|-DCRT_LEAN_SINGLE=0"
export AB_RUNS="d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60;k4_spp1|--workload mesh1m --depth 1 --spp 1 --resolution 3840x2160 --steps 30;d4_spp1|--workload mesh1m --depth 4 --spp 1 --steps 30;hbm_spp1|--workload mesh520 --device-built sah --depth 1 --spp 1 --steps 20;d1_disney_spp1|--workload mesh1m --depth 1 --spp 1 --materials disney --steps 60;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
