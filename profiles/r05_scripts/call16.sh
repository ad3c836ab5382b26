set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="gridpad"
M="--workload mesh1m --depth 4 --spp 4 --streams 1"
export AB_RUNS="pad1|$M;pad1b|$M"
bash tools/ab_run.sh gpurun_out/r5q
export CRT_GRID_PAD=2
export AB_RUNS="pad2|$M"
bash tools/ab_run.sh gpurun_out/r5q
export CRT_GRID_PAD=3
export AB_RUNS="pad3|$M"
bash tools/ab_run.sh gpurun_out/r5q
