# k_shadow_deferred compiled for 6 (default) / 7 / 8 waves per SIMD; its persistent grid sized for 6 / 7 / 8
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aa; mkdir -p $O
export AB_CHECK=" "
M="--workload mesh1m --spp 4"
export AB_LIBS="default,shocc7,shocc8"
export AB_RUNS="d4|$M --depth 4;d4_w6|$M --depth 4 --option shadow_waves=6;d4_w7|$M --depth 4 --option shadow_waves=7;d2|$M --depth 2;disney|$M --depth 4 --materials disney;hbm_d4|--workload mesh520 --depth 4 --spp 4 --device-built sah"
bash tools/ab_run.sh $O
