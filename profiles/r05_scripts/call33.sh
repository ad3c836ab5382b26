# which segments defer their shadow rays: 3 = bounce segments (default), 0 = all of them, 2 = as 3 on every tree
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ah; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
export AB_CHECK=" "
export AB_RUNS="d4|$M --depth 4;d4_all|$M --depth 4 --option inplace_shadow=0;d2|$M --depth 2;d2_all|$M --depth 2 --option inplace_shadow=0;hbm|--workload mesh520 --depth 4 --spp 4 --device-built sah;hbm_all|--workload mesh520 --depth 4 --spp 4 --device-built sah --option inplace_shadow=0;d4_p128|$M --depth 4 --option shadow_pool=128;d4_p512|$M --depth 4 --option shadow_pool=512;d4_r8|$M --depth 4 --option shadow_refill_min=8;d4_r24|$M --depth 4 --option shadow_refill_min=24"
bash tools/ab_run.sh $O
