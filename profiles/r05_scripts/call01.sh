set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default,exp"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;d4_q|--workload mesh1m --depth 4 --spp 4 --option inplace_shadow=0;d4_refill|--workload mesh1m --depth 4 --spp 4 --option bounce_refill=1;d4_refill_q|--workload mesh1m --depth 4 --spp 4 --option bounce_refill=1 --option inplace_shadow=0"
bash tools/ab_run.sh gpurun_out/r5a
