# multi-segment frames on 1 / 2 / 3 streams with the deferred shadow rays
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ae; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
R=""
for S in 2 1 3 2; do R="$R;d4_s$S|$M --depth 4 --streams $S;d2_s$S|$M --depth 2 --streams $S"; done
for S in 2 1; do R="$R;hbm_s$S|--workload mesh520 --depth 4 --spp 4 --device-built sah --streams $S;disney_s$S|$M --depth 4 --materials disney --streams $S"; done
export AB_RUNS="${R#;}"
bash tools/ab_run.sh $O
