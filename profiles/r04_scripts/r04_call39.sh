#!/bin/bash
# group phase: the group's node staged through LDS by global_load_lds_dwordx4: A/B, then the variant's parity
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4an; mkdir -p $OUT
cd $R
export AB_BUILDS="dflt|;slots4|-DCRT_HIT_SLOTS=4;stage|-DCRT_HIT_SLOTS=4 -DCRT_GROUP_LDS_STAGE=1"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d2|--workload mesh1m --depth 2 --spp 4;d4_disney|--workload mesh1m --depth 4 --spp 4 --materials disney;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d1|--workload mesh1m --depth 1 --spp 4;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10"
bash tools/ab.sh $OUT
rm -f caitlynrenderer_amd/csrc/rt_kernels.o caitlynrenderer_amd/csrc/crt_device.o; make -C caitlynrenderer_amd/csrc -s EXTRA="-DCRT_HIT_SLOTS=4 -DCRT_GROUP_LDS_STAGE=1" > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest_stage.log 2>&1; echo "stage pytest rc $?"; tail -3 $OUT/pytest_stage.log
rm -f caitlynrenderer_amd/csrc/rt_kernels.o caitlynrenderer_amd/csrc/crt_device.o; make -C caitlynrenderer_amd/csrc -s > /dev/null 2>&1
