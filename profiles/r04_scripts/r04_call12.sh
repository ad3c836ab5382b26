#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4m; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or mirror_and_disney_materials_match or config4 or streams_option" > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="two_loops|"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_l1|--workload mesh1m --depth 4 --spp 4 --option lanes_per_ray=1;d2|--workload mesh1m --depth 2 --spp 4;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm_d4_l1|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option lanes_per_ray=1"
bash tools/ab.sh $OUT
