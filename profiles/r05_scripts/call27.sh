# is today's library slower on the multi-segment frames?  default (working tree) vs prev (the commit that measured 7.7 / 10.2)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ab; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,prev,shocc8"
export AB_RUNS="d4|$M --depth 4;d2|$M --depth 2;d1|$M --depth 1;d4_again|$M --depth 4"
bash tools/ab_run.sh $O
