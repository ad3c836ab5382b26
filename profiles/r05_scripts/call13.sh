set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5n/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5n/pytest.log
export AB_LIBS="default"
export AB_RUNS="d1_lanes|--workload mesh1m --depth 1 --spp 4;d1_seq4|--workload mesh1m --depth 1 --spp 4 --option wave_samples=0;d1_single|--workload mesh1m --depth 1 --spp 1;d1_single_w5|--workload mesh1m --depth 1 --spp 1 --option wide_first=0"
bash tools/ab_run.sh gpurun_out/r5n
