set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default" AB_CHECK=" "
M="--workload mesh1m --depth 4 --spp 4"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|$M;d4_def|$M --option inplace_shadow=2;d4_def_p256|$M --option inplace_shadow=2 --option shadow_pool=256 --option shadow_refill_min=8;d4_def_p128|$M --option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16;d4_refill|$M --option bounce_refill=1;d4_refill128|$M --option bounce_refill=1 --option refill_pool=128;d4_refill64ls|$M --option bounce_refill=1 --option refill_pool=64 --option refill_min=65;d4_wavefront|$M --option bounce_refill=1 --option inplace_shadow=2;d4_wavefront_p|$M --option bounce_refill=1 --option inplace_shadow=2 --option shadow_pool=256 --option shadow_refill_min=8;d2|--workload mesh1m --depth 2 --spp 4;d2_def|--workload mesh1m --depth 2 --spp 4 --option inplace_shadow=2"
bash tools/ab_run.sh gpurun_out/r5c
