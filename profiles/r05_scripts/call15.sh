set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default"
M="--workload mesh1m --depth 4 --spp 4"
export AB_RUNS="d4|$M;d4_bins1|$M --option ray_bins=1;d4_bins4|$M --option ray_bins=4;d4_bins5|$M --option ray_bins=5;d4_s3|$M --streams 3;d4_s1|$M --streams 1;d4_tm1|$M --option tri_min=1;d4_tm3|$M --option tri_min=3"
bash tools/ab_run.sh gpurun_out/r5p
