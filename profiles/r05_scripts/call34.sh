# line-aligned nodes / records again, on this round's memory-path-bound bounce kernels (round 3: -2 %)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ai; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,rows8,rows8t4,t4"
export AB_CHECK=" "
export AB_RUNS="d4|$M --depth 4;d2|$M --depth 2;d1|$M --depth 1;hbm_d4|--workload mesh520 --depth 4 --spp 4 --device-built sah;hbm_d1|--workload mesh520 --depth 1 --spp 4 --device-built sah"
bash tools/ab_run.sh $O
