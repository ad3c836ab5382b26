# queue entries and path state (written once, read once) with the non-temporal hint
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5au; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,streamnt"
export AB_CHECK=" "
export AB_RUNS="d4|$M --depth 4;d2|$M --depth 2;disney|$M --depth 4 --materials disney;hbm_d4|--workload mesh520 --depth 4 --spp 4 --device-built sah;d4_again|$M --depth 4"
bash tools/ab_run.sh $O
