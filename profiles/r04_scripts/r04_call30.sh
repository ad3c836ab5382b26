#!/bin/bash
# A/B on the lean loops: two waits for the planes, the first segment's shadow rays through the voting loop, tri_min
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4af; mkdir -p $OUT
cd $R
export AB_BUILDS="dflt|;wait2|-DCRT_PLANES_ONE_WAIT=2;any_voting|-DCRT_FIRST_ANY_GROUPS=0 -DCRT_FIRST_ANY_VOTING=1"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d4|--workload mesh1m --depth 4 --spp 4;d1_tm1|--workload mesh1m --depth 1 --spp 4 --option tri_min=1;d1_tm3|--workload mesh1m --depth 1 --spp 4 --option tri_min=3;d4_tm1|--workload mesh1m --depth 4 --spp 4 --option tri_min=1;d4_tm3|--workload mesh1m --depth 4 --spp 4 --option tri_min=3"
bash tools/ab.sh $OUT
rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s EXTRA="-DCRT_FIRST_ANY_GROUPS=0 -DCRT_FIRST_ANY_VOTING=1" > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduling_and_loop or render_frames_equals or radiance_matches or full_resolution or million" > $OUT/pytest_any_voting.log 2>&1; echo "any_voting pytest rc $?"; tail -3 $OUT/pytest_any_voting.log
rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s > /dev/null 2>&1
