set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default,occ7,occ8,default,occ8"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;hbm4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d2|--workload mesh1m --depth 2 --spp 4;d4dis|--workload mesh1m --depth 4 --spp 4 --materials disney"
bash tools/ab_run.sh gpurun_out/r5u
