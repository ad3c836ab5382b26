#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4e; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or full_resolution or mirror_and_disney_materials_match or any_hit" > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d1_al0|--workload mesh1m --depth 1 --spp 4 --option any_lanes=0;d4|--workload mesh1m --depth 4 --spp 4;d4_al0|--workload mesh1m --depth 4 --spp 4 --option any_lanes=0;d4_al1_ts0|--workload mesh1m --depth 4 --spp 4 --option tri_share=0;d2|--workload mesh1m --depth 2 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d1_al0|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10 --option any_lanes=0;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm_d4_al0|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option any_lanes=0;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;k4_al0|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160 --option any_lanes=0"
bash tools/ab.sh $OUT
