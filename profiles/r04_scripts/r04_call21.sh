#!/bin/bash
# A/B: node prefetch through global_load_lds (group phase / one-lane phase), with the control build that only has the third slot
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4w; mkdir -p $OUT
cd $R
export AB_BUILDS="dflt|;slots3|-DCRT_HIT_SLOTS=3;group_pf|-DCRT_HIT_SLOTS=3 -DCRT_GROUP_PREFETCH;both_pf|-DCRT_HIT_SLOTS=3 -DCRT_GROUP_PREFETCH -DCRT_P1_PREFETCH"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d1|--workload mesh1m --depth 1 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10"
bash tools/ab.sh $OUT
# parity of the last variant
rm -f caitlynrenderer_amd/csrc/rt_kernels.o caitlynrenderer_amd/csrc/crt_device.o; make -C caitlynrenderer_amd/csrc -s EXTRA="-DCRT_HIT_SLOTS=3 -DCRT_GROUP_PREFETCH -DCRT_P1_PREFETCH" > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduling_and_loop or render_frames_equals or radiance_matches or full_resolution or config4" > $OUT/pytest_variant.log 2>&1; echo "variant pytest rc $?"; tail -3 $OUT/pytest_variant.log
# parity of the first-segment shadow walk = plain loop + group phase
rm -f caitlynrenderer_amd/csrc/rt_kernels.o caitlynrenderer_amd/csrc/crt_device.o; make -C caitlynrenderer_amd/csrc -s EXTRA="-DCRT_FIRST_ANY_GROUPS=1" > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_first_any_groups.log 2>&1; echo "first_any_groups pytest rc $?"; tail -3 $OUT/pytest_first_any_groups.log
rm -f caitlynrenderer_amd/csrc/rt_kernels.o caitlynrenderer_amd/csrc/crt_device.o; make -C caitlynrenderer_amd/csrc -s > /dev/null 2>&1
