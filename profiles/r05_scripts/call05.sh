set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default,noguard,r4,default,noguard,r4"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;cornell|--workload cornell --depth 1 --spp 1"
bash tools/ab_run.sh gpurun_out/r5e
