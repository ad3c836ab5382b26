# records / nodes fetched with the non-temporal hint (global_load ... nt): does keeping the records out of the caches' way help the nodes?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5at; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,trint,nodent"
export AB_CHECK=" "
export AB_RUNS="d4|$M --depth 4;d2|$M --depth 2;d1|$M --depth 1;hbm_d4|--workload mesh520 --depth 4 --spp 4 --device-built sah;hbm_d1|--workload mesh520 --depth 1 --spp 4 --device-built sah"
bash tools/ab_run.sh $O
