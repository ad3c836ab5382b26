set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
CRT_LIB=$GRAFT_REPO_ROOT/variants/mid7/libcrt.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scheduling_and_loop or deferred_shadow" > gpurun_out/r5y/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5y/pytest.log
export AB_LIBS="head,mid8,mid7"
M="--workload mesh1m --depth 4 --spp 4"
export AB_RUNS="d4|$M;d4_mid|$M --option mid_pairs=1;d2_mid|--workload mesh1m --depth 2 --spp 4 --option mid_pairs=1;d2|--workload mesh1m --depth 2 --spp 4"
bash tools/ab_run.sh gpurun_out/r5y
python3 tools/lane_util.py mesh1m 1 lanes 2>/dev/null | tee gpurun_out/r5y/lane.txt; python3 tools/lane_util.py mesh1m 4 lanes 2>/dev/null | tee -a gpurun_out/r5y/lane.txt
