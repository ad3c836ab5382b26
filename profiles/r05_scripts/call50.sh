# s_setprio around the node / record loads: waves that are about to feed the address unit issue first
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ay; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,prio3,prio1"
export AB_CHECK=" "
export AB_RUNS="d4|$M --depth 4;d2|$M --depth 2;d1|$M --depth 1;hbm_d4|--workload mesh520 --depth 4 --spp 4 --device-built sah;cornell|--workload cornell --depth 1 --spp 1"
bash tools/ab_run.sh $O
