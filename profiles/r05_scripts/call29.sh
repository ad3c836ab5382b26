# the persistent any-hit grid of the deferred shadow rays sized for 4 .. 8 waves per SIMD (kernel compiled for 8)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ad; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
R=""
for W in 8 4 5 6 7 8 6; do R="$R;d4_w$W|$M --depth 4 --option shadow_waves=$W"; done
for W in 8 5 6; do R="$R;d2_w$W|$M --depth 2 --option shadow_waves=$W;hbm_w$W|--workload mesh520 --depth 4 --spp 4 --device-built sah --option shadow_waves=$W"; done
export AB_RUNS="${R#;}"
bash tools/ab_run.sh $O
