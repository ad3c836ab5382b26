#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4d; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or config4_million or mirror_and_disney_materials_match" > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="occ6|"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d1_ts8|--workload mesh1m --depth 1 --spp 4 --option tri_share=11;d1_ts8_tm1|--workload mesh1m --depth 1 --spp 4 --option tri_share=11 --option tri_min=1;d4|--workload mesh1m --depth 4 --spp 4;d4_ts0|--workload mesh1m --depth 4 --spp 4 --option tri_share=0;d4_ts4|--workload mesh1m --depth 4 --spp 4 --option tri_share=4;d4_ts5|--workload mesh1m --depth 4 --spp 4 --option tri_share=5;d4_ts0_s3|--workload mesh1m --depth 4 --spp 4 --option tri_share=0 --streams 3;d4_ts4_s3|--workload mesh1m --depth 4 --spp 4 --option tri_share=4 --streams 3;d2_ts4|--workload mesh1m --depth 2 --spp 4 --option tri_share=4;d2_ts0|--workload mesh1m --depth 2 --spp 4 --option tri_share=0;hbm_d4_ts0|--workload mesh520 --device-built sah --depth 4 --spp 4 --option tri_share=0 --steps 10;hbm_d4_ts4|--workload mesh520 --device-built sah --depth 4 --spp 4 --option tri_share=4 --steps 10;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d1_ts8|--workload mesh520 --device-built sah --depth 1 --spp 4 --option tri_share=11 --steps 10"
bash tools/ab.sh $OUT
