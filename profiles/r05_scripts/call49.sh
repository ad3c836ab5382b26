# vote ratio of the voting loops once more, now that the bounce kernels are memory-path-bound (a triangle step is the TA-heavier one)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ax; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
R=""
for T in 2 1 3 4 6 2; do R="$R;d4_t$T|$M --depth 4 --option tri_min=$T"; done
for T in 2 4; do R="$R;hbm_t$T|--workload mesh520 --depth 4 --spp 4 --device-built sah --option tri_min=$T"; done
export AB_RUNS="${R#;}"
bash tools/ab_run.sh $O
