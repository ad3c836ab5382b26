#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4u; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or mirror_and_disney_materials_match or config4 or full_resolution" > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC (default build: narrow loads, group_phase)"; tail -3 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;first_any_groups|-DCRT_FIRST_ANY_GROUPS=1"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d4|--workload mesh1m --depth 4 --spp 4;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
# the variant's own parity
rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s EXTRA="-DCRT_FIRST_ANY_GROUPS=1" > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_triangle or scheduling_and_loop or render_frames_equals or radiance_matches or full_resolution" > $OUT/pytest_variant.log 2>&1; echo "variant pytest rc $?"; tail -3 $OUT/pytest_variant.log
