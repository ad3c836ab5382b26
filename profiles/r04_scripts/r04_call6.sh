#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4g; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or full_resolution or mirror_and_disney_materials_match or any_hit or closest_hit or exact_ties or edge_cases or zero_components" > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="occ6|;occ5|-DCRT_SEG_OCC=5 -DCRT_SEG_OCC_FIRST=5"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d1_l1|--workload mesh1m --depth 1 --spp 4 --option lanes_per_ray=1;d1_l4|--workload mesh1m --depth 1 --spp 4 --option lanes_per_ray=4;d4|--workload mesh1m --depth 4 --spp 4;d4_l1|--workload mesh1m --depth 4 --spp 4 --option lanes_per_ray=1;d4_l2|--workload mesh1m --depth 4 --spp 4 --option lanes_per_ray=2;d4_l4|--workload mesh1m --depth 4 --spp 4 --option lanes_per_ray=4;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm_d4_l1|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option lanes_per_ray=1"
bash tools/ab.sh $OUT
