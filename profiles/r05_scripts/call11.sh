set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5k/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5k/pytest.log
export AB_LIBS="default"
export AB_RUNS="d2|--workload mesh1m --depth 2 --spp 4;d2_in|--workload mesh1m --depth 2 --spp 4 --option inplace_shadow=1;d4dis|--workload mesh1m --depth 4 --spp 4 --materials disney;d4dis_in|--workload mesh1m --depth 4 --spp 4 --materials disney --option inplace_shadow=1;hbm4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm4_in|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option inplace_shadow=1;cornell3|--workload cornell --depth 3 --spp 1;cornell3_def|--workload cornell --depth 3 --spp 1 --option inplace_shadow=2"
bash tools/ab_run.sh gpurun_out/r5k
